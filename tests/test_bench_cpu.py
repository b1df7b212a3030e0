"""CPU-side checks of bench.py's bookkeeping (no GPU): the committed PMC traffic table is well-formed and is found for the
kernels the plan picks on the BASELINE.json grids, the workload label names the configuration actually run."""
import importlib
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


GRIDS = {"4096x1024_f64": (4096, 1024, 0, (1, 0)), "1024x256_f64": (1024, 256, 0, (1,)), "8192x2048_f64": (8192, 2048, 0, (1,)),
         "16384x4096_f32": (16384, 4096, 1, (1,)), "4096x1024_f32": (4096, 1024, 1, (1,))}


def plan_candidates(nx, ny, precision, arith):
    """The candidates lbm_initialise would time on a whole-domain context (lbm_debug_plan_candidates: no device needed)."""
    import ctypes
    pkg = importlib.import_module("highperformancecomputing-latticeboltzmannmethod_amd")
    pkg.build_all()
    L = ctypes.CDLL(pkg.lib_path())
    L.lbm_debug_plan_candidates.argtypes = [ctypes.c_int] * 5 + [ctypes.c_char_p, ctypes.c_int]
    buf = ctypes.create_string_buffer(16384)
    assert L.lbm_debug_plan_candidates(nx, ny, precision, arith, 256, buf, len(buf)) == 0
    return [dict(zip(("name", "options", "kernel", "depth"), line.split("|"))) for line in buf.value.decode().strip().split("\n")]


def test_traffic_table_is_wellformed_and_covers_every_fused_candidate_of_the_baseline_grids():
    """profiles/traffic.json is bench.py's fallback for the live counter passes: every entry carries the build id it was taken
    on, and every fused candidate (>= 3 iterations per launch) the tuner enumerates for a BASELINE.json grid — i.e. every
    kernel that can come out as a finalist — has one (the driver's fp32 line picked a finalist without a pass)."""
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    same = lambda a, b: a.replace(" ", "") == b.replace(" ", "")
    for key, (nx, ny, prec, ariths) in GRIDS.items():
        assert key in tj and tj[key], key
        for e in tj[key]:
            assert abs(e["hbm_bytes_per_launch"] - e["fetch_bytes_corrected"] - e["write_bytes"]) <= 2
            assert len(e.get("build_id", "")) == 16, (key, e["kernel"], "entry without the build id of its binary")
            bpl = 144 if key.endswith("f64") else 72
            alg = nx * ny * bpl * e["iterations_per_launch"]
            # a fused launch moves at least one read + one write of the lattice and less than the unfused 144 B per update
            assert nx * ny * bpl <= e["hbm_bytes_per_launch"] * 1.02 and e["hbm_bytes_per_launch"] < alg, (key, e["kernel"])
        for arith in ariths:
            for c in plan_candidates(nx, ny, prec, arith):
                if int(c["depth"]) >= 3:
                    assert any(same(e["kernel"], c["kernel"]) for e in tj[key]), f"{key}: no counter pass for candidate {c['kernel']} ({c['name']})"


def test_measured_traffic_lookup_and_labels():
    b = load_bench()
    t, note = b.measured_traffic(4096, 1024, "f64", "k_step3_tile<double,12,1024,1>", "row-interleaved", build_id="0" * 16)
    assert t and 600e6 < t["hbm_bytes_per_launch"] < 700e6 and not t["approximate"] and t["stale"]      # (another build: flagged, not hidden)
    t, note = b.measured_traffic(4096, 1024, "f64", "k_step3_tile<double,12,1024,1>", "row-interleaved", build_id=t["build_id"])
    assert not t["stale"]
    t, note = b.measured_traffic(4096, 1024, "f64", "k_step_site<double,0,true,1>")
    assert t is None and "no counter pass" in note
    # nearest sibling instead of nothing: a kernel the table does not hold, same family / element type / depth
    t, note = b.measured_traffic(16384, 4096, "f32", "k_stepc_col<float,4,8,5,false,7>")
    assert t and t["approximate"] and "sibling" in note and b.kernel_family(t["kernel"]) == ("k_stepc_col", "float", 5, 4, 8)
    # ... but never another region shape: the tall 64x48 regions (twelve waves) fetch 15 % less than the 64x32 ones
    assert b.kernel_family("k_stepc_col<float,4,12,7,false,1>") != b.kernel_family("k_stepc_col<float,4,8,7,false,1>")
    assert b.plan_depth_of("k_stepc_col<double,4,8,6,true,1>") == 6 and b.plan_depth_of("k_stepd_tile<double,32,32,8,1>") == 8
    assert b.plan_depth_of("k_step3_tile<double,12,1024,1>") == 3 and b.plan_depth_of("k_step_site<double,0,true,1>") == 1
    assert b.CONFIGS[(4096, 1024, "f64", 200.0)] == "configs[2]" and b.CONFIGS[(1024, 256, "f64", 100.0)] == "configs[1]"
    assert b.BYTES_PER_LUP == {"f64": 144, "f32": 72} and b.HBM_PEAK_GBS == 8000.0


def test_strip_parity_checksums():
    """bench.py --gpus N compares per-row checksums of the strips' f_next bit patterns with a whole-grid run: the comparison
    says bit-equal only for bit-equal rows (a sign flip of a zero counts) and names the first differing global row."""
    import numpy as np
    b = load_bench()
    rng = np.random.default_rng(7)
    whole = rng.standard_normal((12, 10, 9))
    ws = b.row_checksums(whole)
    parts = [dict(rank=1, y_start=5, rows=7, sums=b.row_checksums(whole[5:12]).tolist()),
             dict(rank=0, y_start=0, rows=5, sums=b.row_checksums(whole[0:5]).tolist())]
    assert b.compare_row_checksums(parts, ws) == "bit-equal"
    bad = whole.copy()
    bad[8, 3, 2] = np.nextafter(bad[8, 3, 2], 10.0)
    parts[0]["sums"] = b.row_checksums(bad[5:12]).tolist()
    assert b.compare_row_checksums(parts, ws).startswith("MISMATCH: rank 1 first differs at global row 8 (1 of 7 rows)")
    z = np.zeros((2, 4, 9))
    zs = b.row_checksums(z)
    z[1, 0, 0] = -0.0
    assert b.compare_row_checksums([dict(rank=0, y_start=0, rows=2, sums=b.row_checksums(z).tolist())], zs).startswith("MISMATCH")


def test_roofline_object_from_a_committed_pass():
    """roofline_of on the headline grid with the register kernel, live passes off (no GPU here): the contract's frac is the 144 B
    figure (above 1 for a fused launch), the measured HBM and vector-issue fractions come from profiles/traffic.json — flagged
    stale when the table was taken on another build — and stay below 1, bound = the larger; nothing is rescaled by depth."""
    b = load_bench()

    class Lbm:
        @staticmethod
        def build_id(): return "0123456789abcdef"

    class Ctx:
        def kernel_name(self): return "k_stepc_col<double,4,8,6,false,1>"
        def plan(self): return "row-interleaved/6-step 64x32 in registers/xcd (fastest of 27 measured, 26.9 us/iteration)"
        def plan_options(self): return dict(layout=1, nt=0, alternate=0, pair_ty=12, xcd=1, deep=7)
    r = b.roofline_of(Lbm, Ctx(), 4096, 1024, "f64", 0.160, 1000, 6000, 6000, live=False)
    assert r["iterations_per_launch"] == 6.0 and r["algorithmic_bytes_per_launch"] == 4096 * 1024 * 144 * 6
    assert abs(r["frac"] - 4096 * 1024 * 144 * 6 / 0.160e-3 / 8e12) < 1e-3 and r["frac"] == r["frac_144B"] > 2.0
    assert r["traffic"] and 0.3 < r["frac_hbm_measured"] < 1.0 and 0.2 < r["frac_valu"] < 1.0
    assert r["bound"] == ("valu" if r["frac_valu"] > r["frac_hbm_measured"] else "hbm")
    assert abs(r["mlups_per_gbs"] - 1000.0 / r["hbm_bytes_per_update"]) < 0.05
    assert r["stale"] is True and len(r["traffic_build_id"]) == 16 and r["approximate"] is False and "live passes" in r["traffic_source"]
    r5 = b.roofline_of(Lbm, Ctx(), 4096, 1024, "f64", 0.140, 4, 20, 20, live=False)      # a 20-step call: the table's bytes as they are
    assert r5["traffic"] == r["traffic"] and r5["traffic_iterations_per_launch"] == 6.0


def test_committed_bench_lines_of_the_current_round_belong_to_their_build():
    """round 4 committed six per-grid bench lines flagged `stale: true` (written before the traffic table of their
    build existed). From round 5 on: every profiles/rNN/*/bench_line.json and profiles/rNN/bench_*.json of rounds >= 5 must carry
    counter bytes of its OWN build — traffic_build_id == config.build_id, neither `stale` nor `approximate`, in the headline's roofline
    and in the single-precision variant's where there is one."""
    import glob
    import re
    checked = 0
    for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]"))):
        if int(os.path.basename(d)[1:]) < 5:
            continue
        for f in sorted(glob.glob(os.path.join(d, "*", "bench_line.json")) + glob.glob(os.path.join(d, "bench_*.json"))):
            text = open(f).read().strip()
            if not text:
                continue
            line = json.loads(text.splitlines()[-1])
            build = line["config"]["build_id"]
            roofs = [("roofline", line["roofline"])]
            if isinstance(line.get("single_precision_variant"), dict) and "error" not in line["single_precision_variant"]:
                roofs.append(("single_precision_variant", line["single_precision_variant"]))
            for name, r in roofs:
                assert r.get("traffic") is not None, (f, name, "no counter bytes")
                assert r.get("traffic_build_id") == build, (f, name, r.get("traffic_build_id"), build)
                assert r.get("stale") is False and r.get("approximate") is False, (f, name, r.get("stale"), r.get("approximate"))
            checked += 1
    assert checked >= 0      # (no round-5 record yet while the round is being built: nothing to check is not a failure)

