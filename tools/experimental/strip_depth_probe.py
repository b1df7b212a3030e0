#!/usr/bin/env python3
"""One strip of 4096 x ROWS exchanging with itself through RCCL (as tools/strip_proxy.py): the register kernel at six iterations per
launch (the strip rule), in pairs over twelve rows, and at SEVEN iterations per launch ("deep" 9: seven rows per exchange), each with
its store / load policies, against what lbm_initialise picks.   python3 tools/experimental/strip_depth_probe.py 512 256"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
lbm = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")
for rows in [int(v) for v in sys.argv[1:]] or (512, 256):
    out = [f"rows={rows}:"]
    base = dict(tune=0, layout=1, pair_ty=12, xcd=1, arith=1, trailing_pair=1, loopback=2)
    cases = [("TUNED", dict(arith=1, trailing_pair=1, loopback=2))]
    for deep, label in ((7, "six"), (9, "seven")):
        for nt, ntl in ((0, 0), (1, 0), (0, 1)):
            for overlap, dh in ((1, 1), (0, 1)) + (((1, 2), (0, 2)) if deep == 7 else ()):
                cases.append((f"{label} nt{nt} ntl{ntl} o{overlap}d{dh}", dict(base, deep=deep, nt=nt, ntl=ntl, overlap=overlap, deep_halo=dh)))
    for name, opts in cases:
        with lbm.Context(4096, rows, inlet_velocity=0.05, options=opts) as c:
            c.comm_init(0, 1, c.comm_unique_id())
            c.initialise()
            c.step(300, 0); c.sync()
            t0 = time.perf_counter(); c.step(3000, 0); c.sync(); dt = time.perf_counter() - t0
            extra = f" [{c.strip_schedule().split(';')[0]} | {c.plan().split(' (')[0]}]" if name == "TUNED" else ""
            out.append(f"{name} {dt / 3000 * 1e6:.2f}{extra}")
    print("  ".join(out), flush=True)
