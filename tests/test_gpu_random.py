"""Seeded random sweep of the parameter space: grid sizes (ragged, tiny, one tile column, many), tau, inlet velocity,
cylinder position/radius (inside, on the inlet, on a wall, on a corner, absent, covering the outlet), fusion depth,
kernel family (LDS tile / deep LDS tile / register column), layout, store policy, collision arithmetic and — where the grid is tall enough —
an in-process group of strips with a random exchange schedule. Every case is compared with the CPU oracle: strict
arithmetic populations bit for bit, contracted arithmetic within 1e-10; rho/u and forces to 1e-10 either way.
"""
import importlib

import numpy as np
import pytest

from tests.helpers import macro_errors

pytestmark = pytest.mark.gpu
PKG = "highperformancecomputing-latticeboltzmannmethod_amd"


def cases(n=36, seed=20260104):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        nx = int(rng.choice([rng.integers(2, 40), rng.integers(40, 200), rng.integers(200, 700), 64, 128, 192, 65, 63]))
        ny = int(rng.choice([rng.integers(2, 12), rng.integers(12, 60), rng.integers(60, 140), 8, 12, 13, 24, 25]))
        tau = float(rng.uniform(0.56, 1.2))
        u = float(rng.uniform(0.005, 0.09))
        cyl = rng.integers(0, 6)
        cx, cy, cr = [(0.2, 0.5, 0.05), (-1.0, 0.5, 0.0), (0.0, 0.5, 0.15), (0.5, 0.0, 0.2), (0.0, 0.0, 0.3), (0.98, 0.5, 0.25)][cyl]
        cx += float(rng.uniform(-0.02, 0.02)) if cyl not in (1,) else 0.0
        steps = int(rng.integers(1, 90))
        of = int(rng.integers(1, 25))
        opts = dict(tune=0, layout=int(rng.integers(0, 2)), retired=int(rng.integers(0, 2)), nt=int(rng.integers(0, 2)),
                    alternate=int(rng.integers(0, 2)), fuse=int(rng.integers(1, 4)), pair_ty=int(rng.choice([8, 12])),
                    xcd=int(rng.integers(0, 2)))
        out.append((k, nx, ny, tau, u, cx, cy, cr, steps, of, opts, 1))
    # round 2: the same generator continued with the new degrees of freedom
    for k in range(n, 2 * n):
        nx = int(rng.choice([rng.integers(2, 40), rng.integers(40, 200), rng.integers(200, 700), 64, 128, 192, 65, 63]))
        ny = int(rng.choice([rng.integers(2, 12), rng.integers(12, 60), rng.integers(60, 140), 8, 12, 13, 24, 25, 48, 96]))
        tau = float(rng.uniform(0.56, 1.2))
        u = float(rng.uniform(0.005, 0.09))
        cyl = rng.integers(0, 6)
        cx, cy, cr = [(0.2, 0.5, 0.05), (-1.0, 0.5, 0.0), (0.0, 0.5, 0.15), (0.5, 0.0, 0.2), (0.0, 0.0, 0.3), (0.98, 0.5, 0.25)][cyl]
        cx += float(rng.uniform(-0.02, 0.02)) if cyl not in (1,) else 0.0
        steps = int(rng.integers(1, 90))
        of = int(rng.integers(1, 25))
        fuse = int(rng.integers(1, 5))
        opts = dict(tune=0, layout=int(rng.integers(0, 2)), retired=int(rng.integers(0, 2)), nt=int(rng.integers(0, 2)),
                    alternate=int(rng.integers(0, 2)), fuse=fuse, pair_ty=int(rng.choice([8, 12])),
                    xcd=int(rng.integers(0, 2)), col=int(rng.integers(0, 2)) if fuse > 1 else 0, arith=int(rng.integers(0, 2)))
        # (round 2 drew the sliding-window kernel here; round 3 retired it: the same draw now selects the register-column
        # kernel k_stepc_col, five or six iterations per launch, on these small odd grids and their strips)
        if opts.pop("col"):
            opts.update(deep=6 + fuse % 2, fuse=5 + fuse % 2)
        strips = int(rng.integers(1, 4)) if ny >= 36 else 1
        if strips > 1:
            opts.update(layout=1, overlap=int(rng.integers(0, 3)), deep_halo=int(rng.integers(0, 2)),
                        group_threads=int(rng.integers(0, 2)))
        out.append((k, nx, ny, tau, u, cx, cy, cr, steps, of, opts, strips))
    # deep tile kernel (6/7/8 iterations per launch): grids from one partial tile to several tiles per direction
    for k in range(2 * n, 2 * n + 40):
        nx = int(rng.choice([rng.integers(2, 40), rng.integers(40, 200), rng.integers(200, 500), 64, 128, 32, 33, 65, 96]))
        ny = int(rng.choice([rng.integers(2, 12), rng.integers(12, 60), rng.integers(60, 140), 16, 17, 32, 33, 48, 64]))
        tau = float(rng.uniform(0.56, 1.2))
        u = float(rng.uniform(0.005, 0.09))
        cyl = rng.integers(0, 6)
        cx, cy, cr = [(0.2, 0.5, 0.05), (-1.0, 0.5, 0.0), (0.0, 0.5, 0.15), (0.5, 0.0, 0.2), (0.0, 0.0, 0.3), (0.98, 0.5, 0.25)][cyl]
        cx += float(rng.uniform(-0.02, 0.02)) if cyl not in (1,) else 0.0
        steps = int(rng.integers(1, 120))
        of = int(rng.integers(1, 40))
        deep = [0, 1, 2, 3, 6, 7][int(rng.integers(1, 6))]      # 1..3: LDS tiles of 1024 threads, 6 / 7: the register-column kernel
        opts = dict(tune=0, layout=int(rng.integers(0, 2)), retired=int(rng.integers(0, 2)), nt=int(rng.integers(0, 2)),
                    alternate=int(rng.integers(0, 2)), pair_ty=int(rng.choice([8, 12])), xcd=int(rng.integers(0, 2)),
                    deep=deep, arith=int(rng.integers(0, 2)), fuse=[0, 6, 7, 8, 0, 0, 5, 6][deep])     # ("fuse" is only the label here: set below)
        strips = int(rng.integers(1, 3)) if ny >= 36 and k % 4 == 0 else 1       # with faces the library falls back to three
        if strips > 1:
            opts.update(layout=1, overlap=int(rng.integers(0, 3)), deep_halo=int(rng.integers(0, 2)))
        out.append((k, nx, ny, tau, u, cx, cy, cr, steps, of, opts, strips))
    # round 3: the register-column kernel (five / six, on whole domains seven iterations per launch) on grids from less than
    # one region to several tiles per direction, alone and under the strip choreography (edge bands, exchange, interior rows)
    for k in range(2 * n + 40, 2 * n + 100):
        nx = int(rng.choice([rng.integers(2, 60), rng.integers(60, 200), rng.integers(200, 700), 54, 55, 56, 57, 108, 112, 64, 128]))
        ny = int(rng.choice([rng.integers(2, 24), rng.integers(24, 80), rng.integers(80, 160), 22, 23, 24, 25, 44, 48, 32, 64]))
        tau = float(rng.uniform(0.56, 1.2))
        u = float(rng.uniform(0.005, 0.09))
        cyl = rng.integers(0, 6)
        cx, cy, cr = [(0.2, 0.5, 0.05), (-1.0, 0.5, 0.0), (0.0, 0.5, 0.15), (0.5, 0.0, 0.2), (0.0, 0.0, 0.3), (0.98, 0.5, 0.25)][cyl]
        cx += float(rng.uniform(-0.02, 0.02)) if cyl not in (1,) else 0.0
        steps = int(rng.integers(1, 130))
        of = int(rng.integers(1, 45))
        deep = int(rng.integers(6, 8))
        opts = dict(tune=0, layout=int(rng.integers(0, 2)), retired=int(rng.integers(0, 2)), nt=int(rng.integers(0, 2)),
                    alternate=int(rng.integers(0, 2)), pair_ty=int(rng.choice([8, 12])), xcd=int(rng.integers(0, 2)),
                    deep=deep, arith=int(rng.integers(0, 2)), fuse=deep - 1)             # ("fuse" is only the label here)
        strips = int(rng.integers(1, 4)) if ny >= 36 and k % 2 == 0 else 1
        if strips > 1:
            opts.update(layout=1, alternate=0, overlap=int(rng.integers(0, 3)), deep_halo=int(rng.integers(0, 2)),
                        group_threads=int(rng.integers(0, 2)))
        out.append((k, nx, ny, tau, u, cx, cy, cr, steps, of, opts, strips))
    # round 4: what the twelve-row ghost frame added — deep plans in pairs over a twelve-row exchange ("deep_halo" 2), the seven- /
    # eight-iteration LDS shapes and the register kernel with seven iterations as its plan's depth ("deep" 9) on strips (seven /
    # eight rows per exchange), non-temporal level-1 loads — on ragged grids, alone and in groups of two to four strips
    for k in range(2 * n + 100, 2 * n + 148):
        nx = int(rng.choice([rng.integers(2, 60), rng.integers(60, 200), rng.integers(200, 600), 52, 53, 54, 104, 64, 128, 32, 33]))
        ny = int(rng.choice([rng.integers(24, 80), rng.integers(80, 200), 20, 22, 40, 44, 48, 64, 96, 100]))
        tau = float(rng.uniform(0.56, 1.2))
        u = float(rng.uniform(0.005, 0.09))
        cyl = rng.integers(0, 6)
        cx, cy, cr = [(0.2, 0.5, 0.05), (-1.0, 0.5, 0.0), (0.0, 0.5, 0.15), (0.5, 0.0, 0.2), (0.0, 0.0, 0.3), (0.98, 0.5, 0.25)][cyl]
        cx += float(rng.uniform(-0.02, 0.02)) if cyl not in (1,) else 0.0
        steps = int(rng.integers(1, 150))
        of = int(rng.integers(1, 60))
        deep = [1, 2, 3, 6, 7, 9][int(rng.integers(0, 6))]
        opts = dict(tune=0, layout=int(rng.integers(0, 2)), nt=int(rng.integers(0, 2)), ntl=int(rng.integers(0, 2)),
                    alternate=int(rng.integers(0, 2)), pair_ty=int(rng.choice([8, 12])), xcd=int(rng.integers(0, 2)),
                    deep=deep, arith=int(rng.integers(0, 2)), trailing_pair=int(rng.integers(0, 2)), fuse={1: 6, 2: 7, 3: 8, 6: 5, 7: 6, 9: 7}[deep])
        strips = min(int(rng.integers(2, 5)), ny // 12) if ny >= 40 and k % 3 != 0 else 1      # (a strip with neighbours holds at least twelve rows)
        if strips > 1:
            opts.update(layout=1, alternate=0, overlap=int(rng.integers(0, 3)), deep_halo=int(rng.integers(0, 3)),
                        group_threads=int(rng.integers(0, 2)))
        out.append((k, nx, ny, tau, u, cx, cy, cr, steps, of, opts, strips))
    return out


@pytest.mark.parametrize("case", cases(), ids=lambda c: f"{c[0]}-{c[1]}x{c[2]}-f{c[10]['fuse']}" + ("c" if c[10].get("deep", 0) >= 6 else "d" if c[10].get("deep") else "")
                         + ("-fast" if c[10].get("arith") else "") + (f"-{c[11]}strips" if c[11] > 1 else ""))
def test_random_case_matches_oracle(case):
    from oracle.oracle import Oracle, make_params
    lbm = importlib.import_module(PKG)
    k, nx, ny, tau, u, cx, cy, cr, steps, of, opts, strips = case
    opts = {key: v for key, v in opts.items() if key != "retired"}      # (the draw of an option retired in round 5: kept so that the cases stay the same)
    if opts.get("deep"):
        opts = {key: v for key, v in opts.items() if key != "fuse"}     # "deep" sets the depth itself
    strict = not opts.get("arith")
    kw = dict(tau=tau, inlet_velocity=u, cylinder_x=cx, cylinder_y=cy, cylinder_radius=cr)
    o = Oracle(make_params(nx, ny, **kw))
    ref_forces = []
    bad = o.run(steps, of, ref_forces)
    with (lbm.Group(nx, ny, strips, options=opts, **kw) if strips > 1 else lbm.Context(nx, ny, options=opts, **kw)) as ctx:
        assert ctx.initialise() == o.solid_count()
        if strips == 1:
            assert np.array_equal(ctx.solid(), o.solid)
        if opts.get("trailing_pair") and steps > 1:      # (a call may then end on a fused launch; the snapshots below need a last single step)
            ctx.step(steps - 1, of)
            ctx.step(1, of)
        else:
            ctx.step(steps, of)
        assert ctx.first_unstable_step() == bad
        if bad != -1:
            return                                   # blown up: only the reported iteration is defined
        fn, fc = ctx.populations("f_next"), ctx.populations("f_current")
        if strict and strips == 1:
            assert np.array_equal(fn, o.f_next) and np.array_equal(fc, o.f_current)
        elif strict:
            assert np.array_equal(fn, o.f_next) and np.array_equal(fc[1:-1], o.f_current[1:-1])
        else:
            scale = float(np.max(np.abs(o.f_next)))
            assert np.max(np.abs(fn - o.f_next)) <= 1e-10 * scale and np.max(np.abs(fc[1:-1] - o.f_current[1:-1])) <= 1e-10 * scale
        rho, ux, uy = ctx.macros()
        er, eu = macro_errors(rho, ux, uy, o.rho, o.ux, o.uy)
        assert er < 1e-10 and eu < 1e-10, (er, eu)
        fscale = max([abs(r[1]) for r in ref_forces] + [1e-300])
        log = ctx.drain_force_log()
        assert [r[0] for r in log] == [r[0] for r in ref_forces]
        for (t, fx, fy), r in zip(log, ref_forces):
            assert abs(fx - r[1]) <= 1e-10 * fscale and abs(fy - r[2]) <= 1e-10 * fscale


def fp32_cases(n=24, seed=20261004):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        nx = int(rng.choice([rng.integers(2, 60), rng.integers(60, 200), rng.integers(200, 500), 50, 52, 53, 64, 104, 128]))
        ny = int(rng.choice([rng.integers(2, 30), rng.integers(30, 120), rng.integers(120, 260), 36, 38, 50, 52, 64, 104]))
        tau = float(rng.uniform(0.56, 1.2))
        u = float(rng.uniform(0.005, 0.09))
        cyl = rng.integers(0, 6)
        cx, cy, cr = [(0.2, 0.5, 0.05), (-1.0, 0.5, 0.0), (0.0, 0.5, 0.15), (0.5, 0.0, 0.2), (0.0, 0.0, 0.3), (0.98, 0.5, 0.25)][cyl]
        steps = int(rng.integers(1, 120))
        of = int(rng.integers(1, 50))
        arith = int(rng.integers(0, 2))
        opts = dict(tune=0, layout=int(rng.integers(0, 2)), nt=0, alternate=int(rng.integers(0, 2)), pair_ty=12,
                    xcd=int(rng.integers(0, 2)), deep=8, arith=arith, trailing_pair=int(rng.integers(0, 2)))
        strips = min(int(rng.integers(2, 4)), ny // 12) if ny >= 40 and k % 2 else 1
        if strips > 1:
            opts.update(layout=1, alternate=0, overlap=int(rng.integers(0, 3)), group_threads=int(rng.integers(0, 2)))
        out.append((k, nx, ny, tau, u, cx, cy, cr, steps, of, opts, strips))
    return out


@pytest.mark.parametrize("case", fp32_cases(), ids=lambda c: f"{c[0]}-{c[1]}x{c[2]}" + ("-fast" if c[10]["arith"] else "") + (f"-{c[11]}strips" if c[11] > 1 else ""))
def test_random_fp32_tall_regions_match_one_launch_per_iteration(case):
    """fp32 has no oracle (the reference is fp64): the tall register shape ("deep" 8: 64x48 regions — twelve waves x four rows, strict arithmetic eight x six; seven
    iterations per launch, six / eight for remainders) is held bit for bit to one k_step_site launch per iteration in the same
    arithmetic on random ragged grids — whole domains and groups of strips; the fp32 path itself is held to the fp64 oracle at a
    stated tolerance in tests/test_gpu_parity.py."""
    lbm = importlib.import_module(PKG)
    k, nx, ny, tau, u, cx, cy, cr, steps, of, opts, strips = case
    kw = dict(tau=tau, inlet_velocity=u, cylinder_x=cx, cylinder_y=cy, cylinder_radius=cr, precision="f32")
    with lbm.Context(nx, ny, options=dict(tune=0, layout=1, nt=1, alternate=0, fuse=1, arith=opts["arith"]), **kw) as ref:
        ref.initialise()
        ref.step(steps, of)
        r_bad, r_fn, r_log = ref.first_unstable_step(), ref.populations("f_next"), ref.drain_force_log()
    with (lbm.Group(nx, ny, strips, options=opts, **kw) if strips > 1 else lbm.Context(nx, ny, options=opts, **kw)) as ctx:
        ctx.initialise()
        if opts.get("trailing_pair") and steps > 1:
            ctx.step(steps - 1, of)
            ctx.step(1, of)
        else:
            ctx.step(steps, of)
        assert ctx.first_unstable_step() == r_bad
        if r_bad != -1:
            return
        assert np.array_equal(ctx.populations("f_next"), r_fn)
        log = ctx.drain_force_log()
        assert [r[0] for r in log] == [r[0] for r in r_log]
        for (t, fx, fy), (_, wx, wy) in zip(log, r_log):
            if strips == 1:
                assert fx == wx and fy == wy
            else:       # (the strips' partial sums are added in another order)
                assert abs(fx - wx) <= 1e-5 * max(1.0, abs(wx)) and abs(fy - wy) <= 1e-5 * max(1.0, abs(wy))
