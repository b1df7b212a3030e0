"""tools/proxy_tuned.py ROWS — one strip of 4096 x ROWS exchanging with itself through RCCL, plan and schedule measured; prints
what the tuner picked, whether launch groups were replayed from a graph, and us per iteration."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lbm = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")
rows = int(sys.argv[1])
with lbm.Context(4096, rows, inlet_velocity=0.05, options=dict(arith=1, trailing_pair=1, loopback=2)) as c:
    c.comm_init(0, 1, c.comm_unique_id())
    c.initialise()
    print("after initialise:", c.strip_schedule(), "|", c.plan(), flush=True)
    c.step(300, 0); c.sync()
    t0 = time.perf_counter(); c.step(3000, 0); t1 = time.perf_counter(); c.sync(); dt = time.perf_counter() - t0
    print(f"{dt / 3000 * 1e6:.2f} us/it, host {(t1 - t0) / 3000 * 1e6:.2f}; {c.strip_schedule()}")
