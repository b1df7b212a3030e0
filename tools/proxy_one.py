import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
lbm = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")
rows = int(sys.argv[1]); lb = int(sys.argv[2]); ov = int(sys.argv[3])
opts = dict(tune=0, layout=1, nt=1, fuse=3, pair_ty=12, xcd=1, loopback=lb, overlap=ov)
with lbm.Context(4096, rows, inlet_velocity=0.05, options=opts) as c:
    if lb == 2: c.comm_init(0, 1, c.comm_unique_id())
    c.initialise(); c.step(60, 0); c.sync()
    t0 = time.perf_counter(); c.step(300, 0); c.sync(); print((time.perf_counter()-t0)/300*1e6, "us/it")
