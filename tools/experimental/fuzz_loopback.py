#!/usr/bin/env python3
"""One-off fuzz of the single-strip transports: a strip that exchanges its edge rows with ITSELF (device copies: loopback 1; one-rank
RCCL communicator: loopback 2) under random plans, exchange schedules, graph replay and call patterns, against the same strip on the
plainest schedule (six-iteration LDS shape, serialised, one exchange per launch, device copies). Bit for bit.
    python3 tools/experimental/fuzz_loopback.py SEED [CASES]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _progress import Progress      # noqa: E402
PROGRESS = Progress("fuzz_loopback", sys.argv[1] if len(sys.argv) > 1 else "1")      # (sets LBM_TRACE before the library loads)
lbm = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")


def run(nx, rows, kw, opts, calls):
    with lbm.Context(nx, rows, options=opts, **kw) as c:
        if opts.get("loopback") == 2:
            c.comm_init(0, 1, c.comm_unique_id())
        c.initialise()
        for n, of in calls:
            c.step(n, of)
        c.step(1, 0)
        c.sync()
        return c.first_unstable_step(), c.populations("f_next")[1:-1], c.drain_force_log(), c.graph_replays()


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(ncases):
        nx = int(rng.choice([64, 130, 256, 300, 512, 1000]))
        rows = int(rng.choice([24, 40, 64, 96, 128, 160, 200, 256]))
        precision = "f64" if rng.integers(0, 4) else "f32"
        arith = int(rng.integers(0, 2))
        kw = dict(inlet_velocity=float(rng.uniform(0.01, 0.08)), tau=float(rng.uniform(0.56, 1.0)), cylinder_radius=float(rng.choice([0.0, 0.08, 0.15])), precision=precision)
        calls = [(int(rng.integers(1, 200)), int(rng.choice([0, 7, 31, 64, 160]))) for _ in range(int(rng.integers(1, 4)))]
        base = dict(tune=0, layout=1, nt=1, alternate=0, pair_ty=12, xcd=1, deep=1, arith=arith, loopback=1, overlap=0, deep_halo=1, graph=0)
        deep = int(rng.choice([1, 2, 3, 6, 7, 9] + ([8] if precision == "f32" else [])))
        opts = dict(base, deep=deep, nt=int(rng.integers(0, 2)) if deep != 8 else 0, ntl=int(rng.integers(0, 2)), loopback=int(rng.integers(1, 3)),
                    overlap=int(rng.integers(0, 3)), deep_halo=int(rng.integers(0, 3)), graph=int(rng.integers(0, 2)), trailing_pair=int(rng.integers(0, 2)),
                    halo_trim=int(rng.integers(0, 2)))
        PROGRESS.start(f"case {k}: {nx}x{rows} {precision} calls={calls} opts={opts}")
        try:
            ref = run(nx, rows, kw, base, calls)
            got = run(nx, rows, kw, opts, calls)
            ok = ref[0] == got[0] and (ref[0] != -1 or (np.array_equal(ref[1], got[1]) and ref[2] == got[2]))
        except Exception as e:      # noqa: BLE001
            ok = False
            got = (None, None, None, str(e))
        PROGRESS.done("ok" if ok else "MISMATCH")
        if not ok:
            bad += 1
            print(f"MISMATCH case {k}: {nx}x{rows} {kw} calls={calls} opts={opts} -> {got[3] if got[0] is None else 'differs'}", flush=True)
        elif k % 10 == 9:
            print(f"seed {seed}: {k + 1} cases, {bad} failing (last: deep {deep}, replays {got[3]})", flush=True)
    print(f"seed {seed}: {ncases} cases, {bad} failing")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
