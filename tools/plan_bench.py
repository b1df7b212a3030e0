#!/usr/bin/env python3
"""Times named plans (lbm_set_option sets) of the step path on one GPU, interleaved in one process.

    python tools/plan_bench.py [--nx 4096 --ny 1024 --precision f64 --steps 3000 --rounds 3] [--plans a,b,...]

Prints us per iteration and GLUPS per plan (median / best over rounds). No torch, no oracle: ctypes on liblbm_hip.so.
"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lbm = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")

PLANS = {
    "auto": None,
    "site": dict(tune=0, layout=1, nt=1, alternate=0, fuse=1),
    "vec": dict(tune=0, layout=0, nt=0, alternate=1, fuse=1),
    "tile2": dict(tune=0, layout=0, nt=1, alternate=0, fuse=2, pair_ty=12, xcd=0),
    "tile3": dict(tune=0, layout=0, nt=1, alternate=0, fuse=3, pair_ty=12, xcd=0),
    "tile3-rowil": dict(tune=0, layout=1, nt=1, alternate=0, fuse=3, pair_ty=12, xcd=1),
    "fast-tile3": dict(tune=0, layout=0, nt=1, alternate=0, fuse=3, pair_ty=12, xcd=0, arith=1),
    "fast-tile3-rowil-xcd": dict(tune=0, layout=1, nt=1, alternate=0, fuse=3, pair_ty=12, xcd=1, arith=1),
    "fast-tile3-rowil-xcd-8": dict(tune=0, layout=1, nt=1, alternate=0, fuse=3, pair_ty=8, xcd=1, arith=1),
    "tile3-rowil-xcd": dict(tune=0, layout=1, nt=1, alternate=0, fuse=3, pair_ty=12, xcd=1),
    "ft3-nt0-alt0": dict(tune=0, layout=1, nt=0, alternate=0, fuse=3, pair_ty=12, xcd=1, arith=1),
    "ft3-nt0-alt1": dict(tune=0, layout=1, nt=0, alternate=1, fuse=3, pair_ty=12, xcd=1, arith=1),
    "ft3-nt1-alt1": dict(tune=0, layout=1, nt=1, alternate=1, fuse=3, pair_ty=12, xcd=1, arith=1),
    "ft3-planar-xcd": dict(tune=0, layout=0, nt=1, alternate=0, fuse=3, pair_ty=12, xcd=1, arith=1),
    "ft3-planar-xcd-alt": dict(tune=0, layout=0, nt=1, alternate=1, fuse=3, pair_ty=12, xcd=1, arith=1),
    "tile4": dict(tune=0, layout=1, nt=1, alternate=0, fuse=4, pair_ty=8, xcd=1),
    "fast-tile4": dict(tune=0, layout=1, nt=1, alternate=0, fuse=4, pair_ty=8, xcd=1, arith=1),
    "fast-tile4-planar": dict(tune=0, layout=0, nt=1, alternate=0, fuse=4, pair_ty=8, xcd=1, arith=1),
    "fast-col5": dict(tune=0, layout=1, nt=1, alternate=0, pair_ty=12, xcd=1, deep=6, arith=1),
    "fast-col6": dict(tune=0, layout=1, nt=1, alternate=0, pair_ty=12, xcd=1, deep=7, arith=1),
    "fast-col6-nt0": dict(tune=0, layout=1, nt=0, alternate=0, pair_ty=12, xcd=1, deep=7, arith=1),
    "col6": dict(tune=0, layout=1, nt=1, alternate=0, pair_ty=12, xcd=1, deep=7),
    "fast-deep6": dict(tune=0, layout=1, nt=1, alternate=0, pair_ty=12, xcd=1, deep=1, arith=1),
    "fast-deep7": dict(tune=0, layout=1, nt=1, alternate=0, pair_ty=12, xcd=1, deep=2, arith=1),
    "fast-deep8": dict(tune=0, layout=1, nt=1, alternate=0, pair_ty=12, xcd=1, deep=3, arith=1),
    "deep7": dict(tune=0, layout=1, nt=1, alternate=0, pair_ty=12, xcd=1, deep=2),
    "deep8": dict(tune=0, layout=1, nt=1, alternate=0, pair_ty=12, xcd=1, deep=3),
    "fast-auto": dict(arith=1),
    "fast-site": dict(tune=0, layout=1, nt=1, alternate=0, fuse=1, arith=1),
    "fast-vec": dict(tune=0, layout=0, nt=0, alternate=1, fuse=1, arith=1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=4096)
    ap.add_argument("--ny", type=int, default=1024)
    ap.add_argument("--re", type=float, default=200.0)
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--plans", default="tile3")
    ap.add_argument("--set", action="append", default=[], help="extra option key=value applied to every plan")
    args = ap.parse_args()
    u_in = args.re * ((0.6 - 0.5) / 3.0) / (2.0 * 0.05 * args.ny)
    extra = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.set}
    names = args.plans.split(",")
    ctxs = {}
    for n in names:
        opts = dict(PLANS[n] or {})
        opts.update(extra)
        opts["trailing_pair"] = 1
        c = lbm.Context(args.nx, args.ny, tau=0.6, inlet_velocity=u_in, precision=args.precision, options=opts)
        c.initialise()
        c.set_option("timing", 1)
        c.step(300, 0)
        c.sync()
        ctxs[n] = c
    res = {n: [] for n in names}
    for _ in range(args.rounds):
        for n in names:
            c = ctxs[n]
            c.sync()
            t0 = time.perf_counter()
            c.step(args.steps, 0)
            c.sync()
            wall = time.perf_counter() - t0
            ms, launches, its = c.last_step_stats()
            res[n].append((ms * 1e3 / its, wall * 1e6 / its))
    cells = args.nx * args.ny
    print(f"== {args.nx}x{args.ny} {args.precision}, {args.steps} iterations x {args.rounds} rounds")
    for n in names:
        dev = sorted(r[0] for r in res[n])
        wall = sorted(r[1] for r in res[n])
        med, best = dev[len(dev) // 2], dev[0]
        bad = ctxs[n].first_unstable_step()
        print(f"  {n:18s} device {med:8.2f} us/it ({cells / med / 1e3:7.1f} GLUPS)  best {best:8.2f} ({cells / best / 1e3:7.1f})"
              f"  wall {wall[len(wall) // 2]:8.2f} us/it  unstable={bad}  [{ctxs[n].kernel_name()} | {ctxs[n].plan()}]", flush=True)
    for c in ctxs.values():
        c.close()


if __name__ == "__main__":
    main()
