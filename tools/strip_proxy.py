#!/usr/bin/env python3
"""tools/strip_proxy.py — how host-bound is one rank of an N-strip run? Runs a 4096 x (1024/N) strip on one GPU with the
loopback halo transport (same host-side choreography as the RCCL path: edge launches, events, side stream; device
copies in place of ncclSend/ncclRecv, or RCCL send/recv to self on a one-rank communicator) and reports us per
iteration vs the same strip without any exchange."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

lbm = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")
nx, ny = 4096, 1024
for n in [int(v) for v in sys.argv[1:]] or (1, 2, 4, 8):
    rows = ny // n
    line = [f"N={n} rows={rows}:"]
    base = dict(tune=0, layout=1, nt=1, fuse=3, pair_ty=12, xcd=1, arith=1, trailing_pair=1)
    deep = dict(tune=0, layout=1, nt=1, pair_ty=12, xcd=1, arith=1, trailing_pair=1, deep=7)      # k_stepc_col, six iterations per launch
    lds6 = dict(deep, deep=1)                                  # k_stepd_tile<64,16,6>: one 1024-thread block per CU, one cell per thread
    for name, opts in (("no-exchange", dict(base)),
                       ("no-exchange lds6", dict(lds6)),
                       ("rccl-self lds6 ovl", dict(lds6, loopback=2, overlap=1)),
                       ("rccl-self lds6 ser", dict(lds6, loopback=2, overlap=0)),
                       ("no-exchange deep", dict(deep)),
                       ("rccl-self deep ovl", dict(deep, loopback=2, overlap=1)),
                       ("rccl-self deep ser", dict(deep, loopback=2, overlap=0)),
                       ("rccl-self lds7 ovl", dict(lds6, deep=2, loopback=2, overlap=1)),      # round 4: seven / eight iterations per launch on strips
                       ("rccl-self lds7 ser", dict(lds6, deep=2, loopback=2, overlap=0)),
                       ("no-exchange lds8", dict(lds6, deep=3)),
                       ("rccl-self lds8 ovl", dict(lds6, deep=3, loopback=2, overlap=1)),
                       ("rccl-self lds8 ser", dict(lds6, deep=3, loopback=2, overlap=0)),
                       ("rccl-self lds6 PAIRS ovl", dict(lds6, loopback=2, overlap=1, deep_halo=2)),      # round 4: twelve rows per two launches
                       ("rccl-self lds6 PAIRS ser", dict(lds6, loopback=2, overlap=0, deep_halo=2)),
                       ("rccl-self lds6 PAIRS next", dict(lds6, loopback=2, overlap=2, deep_halo=2)),
                       ("rccl-self deep PAIRS ovl", dict(deep, loopback=2, overlap=1, deep_halo=2)),
                       ("rccl-self deep PAIRS ser", dict(deep, loopback=2, overlap=0, deep_halo=2)),
                       ("rccl-self deep PAIRS next", dict(deep, loopback=2, overlap=2, deep_halo=2)),
                       ("rccl-self TUNED", dict(arith=1, trailing_pair=1, loopback=2)),
                       ("rccl-self ovl+deep", dict(base, loopback=2, overlap=1, deep_halo=1)),
                       ("rccl-self ser+deep", dict(base, loopback=2, overlap=0, deep_halo=1)),
                       ("rccl-self next+deep", dict(base, loopback=2, overlap=2, deep_halo=1)),
                       ("rccl-self ser+shallow", dict(base, loopback=2, overlap=0, deep_halo=0)),
                       ):
        with lbm.Context(nx, rows, inlet_velocity=0.05, options=opts) as c:
            if opts.get("loopback") == 2:
                c.comm_init(0, 1, c.comm_unique_id())      # one-rank communicator: ncclSend/ncclRecv to self
            c.initialise()
            c.step(300, 0); c.sync()
            t0 = time.perf_counter(); c.step(3000, 0); t1 = time.perf_counter(); c.sync(); dt = time.perf_counter() - t0
            extra = f" [{c.strip_schedule()} | {c.kernel_name()}]" if "TUNED" in name else ""
            # (host: the time lbm_step needs to ISSUE the launches, events and exchanges — the floor of a host-bound strip)
            line.append(f"{name} {dt / 3000 * 1e6:.2f} us/it, host {(t1 - t0) / 3000 * 1e6:.2f} ({nx * rows * 3000 / dt / 1e6 * n:.0f} MLUPS x{n} ranks){extra}")
    print("  ".join(line), flush=True)
