"""tools/proxy_edge.py — one rank of an N-strip run of 4096x1024 exchanging with itself through RCCL on the eight-iteration LDS shape
("deep" 3): the edge-band launch on the plan's 32x32 tiles against thin bands on 32x16 tiles ("edge_band" 16), every overlap; and the
bit-equality of the two forms (populations after 333 iterations).   python3 tools/proxy_edge.py [ROWS ...]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lbm = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")
for rows in [int(v) for v in sys.argv[1:]] or (128, 96, 64):
    ref = None
    for edge in (0, 16):
        for overlap in (1, 0, 2):
            opts = dict(tune=0, layout=1, nt=1, pair_ty=12, xcd=1, arith=1, trailing_pair=1, deep=3, loopback=2, overlap=overlap, deep_halo=1, edge_band=edge)
            with lbm.Context(4096, rows, inlet_velocity=0.05, options=opts) as c:
                c.comm_init(0, 1, c.comm_unique_id())
                c.initialise()
                c.step(333, 0); c.step(1, 0); c.sync()
                fn = c.populations("f_next")[1:-1]
                if ref is None:
                    ref = fn
                same = np.array_equal(fn, ref)
                best = 1e9
                for rep in range(3):
                    c.step(304, 0); c.sync()
                    t0 = time.perf_counter(); c.step(3040, 0); c.sync(); best = min(best, (time.perf_counter() - t0) / 3040 * 1e6)
                print(f"rows {rows} edge_band {edge:2d} overlap {overlap}: {best:.2f} us/iteration, graph replays {c.graph_replays()}, bit-equal to the first form: {same}", flush=True)
