#!/usr/bin/env python3
"""One-off fuzz of in-process groups of strips on one GPU (peer copies between the strips' streams, persistent host threads or not) at
sizes where launches overlap for real: random uneven strips, plans, exchange schedules and output cadences against the one-domain run
with one launch per iteration, bit for bit (populations) — the configuration class in which round 4's edge-band race showed.
    python3 tools/experimental/fuzz_groups.py SEED [CASES]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _progress import Progress      # noqa: E402
PROGRESS = Progress("fuzz_groups", sys.argv[1] if len(sys.argv) > 1 else "1")      # (sets LBM_TRACE before the library loads)
lbm = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(ncases):
        nx = int(rng.choice([256, 512, 777, 1024, 2048]))
        ny = int(rng.integers(150, 700))
        precision = "f64" if rng.integers(0, 4) else "f32"
        arith = int(rng.integers(0, 2))
        kw = dict(inlet_velocity=float(rng.uniform(0.01, 0.08)), tau=float(rng.uniform(0.56, 1.0)), cylinder_radius=float(rng.choice([0.0, 0.08, 0.15])), precision=precision)
        n = int(rng.integers(2, 6))
        cuts = sorted(rng.choice(np.arange(14, ny - 14), size=n - 1, replace=False).tolist())
        edges = [0] + cuts + [ny]
        if min(b - a for a, b in zip(edges, edges[1:])) < 14:
            continue
        bounds = [(a, b - a) for a, b in zip(edges, edges[1:])]
        calls = [(int(rng.integers(20, 300)), int(rng.choice([0, 31, 50, 70, 140]))) for _ in range(int(rng.integers(1, 3)))]
        deep = int(rng.choice([0, 1, 2, 3, 6, 7, 9] + ([8] if precision == "f32" else [])))
        opts = dict(tune=0, layout=1, nt=int(rng.integers(0, 2)) if deep != 8 else 0, ntl=int(rng.integers(0, 2)), alternate=0, pair_ty=12, xcd=1,
                    arith=arith, overlap=int(rng.integers(0, 3)), deep_halo=int(rng.integers(0, 3)), group_threads=int(rng.integers(0, 2)),
                    halo_trim=int(rng.integers(0, 2)))
        if deep:
            opts["deep"] = deep
        else:
            opts["fuse"] = int(rng.integers(1, 4))
        PROGRESS.start(f"case {k}: {nx}x{ny} bounds={bounds} {precision} calls={calls} opts={opts}")
        try:
            with lbm.Context(nx, ny, options=dict(tune=0, layout=1, nt=1, alternate=0, fuse=1, arith=arith), **kw) as w:
                w.initialise()
                for steps, of in calls:
                    w.step(steps, of)
                r_bad, r_fn = w.first_unstable_step(), w.populations("f_next")
            with lbm.Group(nx, ny, bounds, options=opts, **kw) as g:
                g.initialise()
                for steps, of in calls:
                    g.step(steps, of)
                ok = g.first_unstable_step() == r_bad and (r_bad != -1 or np.array_equal(g.populations("f_next"), r_fn))
            msg = "differs"
        except Exception as e:      # noqa: BLE001
            ok, msg = False, str(e)
        PROGRESS.done("ok" if ok else f"MISMATCH {msg}")
        if not ok:
            bad += 1
            print(f"MISMATCH case {k}: {nx}x{ny} bounds={bounds} {kw} calls={calls} opts={opts} -> {msg}", flush=True)
        elif k % 10 == 9:
            print(f"seed {seed}: {k + 1} cases, {bad} failing", flush=True)
    print(f"seed {seed}: {ncases} cases, {bad} failing")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
