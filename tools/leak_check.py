import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
lbm = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")
free0 = torch.cuda.mem_get_info()[0]
for k in range(40):
    with lbm.Context(2048, 512, inlet_velocity=0.05, precision="f32" if k % 2 else "f64") as c:
        c.initialise(); c.step(50, 10); c.macros(); c.populations("f_current"); c.drain_force_log()
free1 = torch.cuda.mem_get_info()[0]
print("free before/after MB:", free0 >> 20, free1 >> 20, "delta MB", (free0 - free1) >> 20)
