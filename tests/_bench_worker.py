"""Worker for tests/test_bench_multirank_cpu.py: one rank of `bench.py --gpus N` with the HIP package replaced by a stand-in that
computes nothing (a TEST DOUBLE for the control flow only: gloo rendezvous, MAX-reductions, the strips.parity gather, the JSON
line, the exit code) — the product bench.py is run unmodified through runpy; the stand-in exists only inside this process.
argv: scenario(ok|mismatch) then bench.py's arguments."""
import os
import runpy
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "highperformancecomputing-latticeboltzmannmethod_amd"
SCENARIO = sys.argv[1]


class FakeContext:
    HALO_ROWS = 6

    def __init__(self, nx, ny, tau=0.6, inlet_velocity=0.01, y_start=0, local_ny=0, precision="f64", device=0, options=None, **kw):
        self.nx, self.ny, self.y_start = nx, ny, y_start
        self.local_ny = local_ny if local_ny > 0 else ny - y_start
        self.steps = 0
        self.opts = dict(options or {})
        self.last = (0.0, 0, 0)

    def __enter__(self): return self
    def __exit__(self, *a): self.close()
    def close(self): pass
    def set_option(self, k, v): self.opts[k] = int(v)
    def comm_unique_id(self): return b"\0" * 128
    def comm_init(self, rank, world, ident): self.rank, self.world = rank, world
    def initialise(self): self.steps = 0; return 0

    def step(self, n, of=0):
        launches = (n + 5) // 6
        time.sleep(0.0005 * launches)
        self.steps += n
        self.last = (0.15 * launches, launches, n)

    def sync(self): pass
    def last_step_stats(self): return self.last
    def first_unstable_step(self): return -1
    def graph_replays(self): return 0
    def kernel_name(self): return "k_stepc_col<double,4,8,6,false,1>"
    def plan(self): return "row-interleaved/6-step 64x32 in registers/xcd (test double)"
    def plan_options(self): return dict(layout=1, nt=0, alternate=0, pair_ty=12, xcd=1, deep=7)
    def strip_schedule(self): return "overlap=1 deep_halo=0 (test double); 1783296 B per face and exchange = 297216 B per face and iteration (6 iterations per exchange)"

    def populations(self, which):
        # a function of (global row, column, direction, iterations): strips reproduce the whole grid — unless the scenario says not
        rows = np.arange(self.y_start - 1, self.y_start + self.local_ny + 1, dtype=np.float64)[:, None, None]
        cols = np.arange(self.nx + 2, dtype=np.float64)[None, :, None]
        a = rows * 1e-3 + cols * 1e-6 + np.arange(9, dtype=np.float64)[None, None, :] + self.steps
        if SCENARIO == "mismatch" and self.y_start > 0:
            a[3, 5, 2] += 1e-9
        return a


fake = types.ModuleType(PKG)
fake.Context = FakeContext
fake.lib = lambda: None
fake.device_count = lambda: 1
fake.build_id = lambda: "0123456789abcdef"
fake.runtime_versions = lambda: {"rccl": 22707, "hip_runtime": 70200000, "hip_driver": 70200000}
sys.modules[PKG] = fake
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
