"""CPU-side checks of bench.py's bookkeeping (no GPU): the committed PMC traffic table is well-formed and is found for the
kernels the plan picks on the BASELINE.json grids, the workload label names the configuration actually run."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_traffic_table_is_wellformed_and_covers_the_baseline_grids():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key in ("4096x1024_f64", "1024x256_f64", "8192x2048_f64", "16384x4096_f32", "4096x1024_f32"):
        assert key in tj and tj[key], key
        for e in tj[key]:
            assert abs(e["hbm_bytes_per_launch"] - e["fetch_bytes_corrected"] - e["write_bytes"]) <= 2
            nx, ny = (int(v) for v in key.split("_")[0].split("x"))
            bpl = 144 if key.endswith("f64") else 72
            alg = nx * ny * bpl * e["iterations_per_launch"]
            # a fused launch moves at least one read + one write of the lattice and less than the unfused 144 B per update
            assert nx * ny * bpl <= e["hbm_bytes_per_launch"] * 1.02 and e["hbm_bytes_per_launch"] < alg, (key, e["kernel"])
            assert os.path.exists(os.path.join(ROOT, e["source"])), e["source"]


def test_measured_traffic_lookup_and_labels():
    b = load_bench()
    t, note = b.measured_traffic(4096, 1024, "f64", "k_step3_tile<double,12,1024,1>", "row-interleaved")
    assert t and 600e6 < t["hbm_bytes_per_launch"] < 700e6 and "profiles/r0" in note
    t, note = b.measured_traffic(4096, 1024, "f64", "k_step_site<double,0,true,1>")
    assert t is None and "no counter pass" in note
    assert b.plan_depth_of("k_stepc_col<double,4,8,6,true,1>") == 6 and b.plan_depth_of("k_stepd_tile<double,32,32,8,1>") == 8
    assert b.plan_depth_of("k_step3_tile<double,12,1024,1>") == 3 and b.plan_depth_of("k_step_site<double,0,true,1>") == 1
    assert b.CONFIGS[(4096, 1024, "f64", 200.0)] == "configs[2]" and b.CONFIGS[(1024, 256, "f64", 100.0)] == "configs[1]"
    assert b.BYTES_PER_LUP == {"f64": 144, "f32": 72} and b.HBM_PEAK_GBS == 8000.0
