"""Characterises a rare mismatch of deep_halo=2 on 64x16 LDS tiles (3 strips of 66/67 rows): which pinned plan / schedule shows it."""
import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
lbm = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")
nx, ny, steps, of = 512, 200, 333, 70
kw = dict(inlet_velocity=0.05, cylinder_radius=0.1)
with lbm.Context(nx, ny, options=dict(tune=0, layout=1, nt=1, alternate=0, fuse=1), **kw) as whole:
    whole.initialise(); whole.step(steps, of); w_fn = whole.populations("f_next")
base = dict(tune=0, layout=1, alternate=0, pair_ty=12, xcd=1, deep=1)
for name, extra in (("tuned", dict(deep_halo=2)), ("nt1", dict(base, nt=1, deep_halo=2)), ("nt1 nothr", dict(base, nt=1, deep_halo=2, group_threads=0)),
                    ("nt1 ser", dict(base, nt=1, deep_halo=2, overlap=0)), ("nt1 single", dict(base, nt=1, deep_halo=1)),
                    ("nt1 of0", dict(base, nt=1, deep_halo=2))):
    bad = []
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        with lbm.Group(nx, ny, 3, options=extra, **kw) as g:
            g.initialise()
            plan = g.ctxs[0].plan()[:60]
            g.step(steps, 0 if name.endswith('of0') else of)
            fn = g.populations("f_next")
            if name.endswith('of0'):
                if rep == 0: ref0 = fn
                d = np.nonzero((fn != ref0).any(axis=(1, 2)))[0]
            else:
                d = np.nonzero((fn != w_fn).any(axis=(1, 2)))[0]
            if d.size: bad.append((rep, int(d.min()), int(d.max()), int(d.size), plan))
    print(name, "mismatching runs (rep, first row, last row, rows, plan):", bad, flush=True)
