"""bench.py --gpus N end to end on the CPU with two gloo ranks and a compute-free stand-in for the HIP package (tests/_bench_worker.py):
the first real multi-GPU run is the driver's, so the part of bench.py only that run executes — rendezvous, MAX-reductions over the
ranks, the strips.parity gather and comparison, the N>1 fields of the JSON line, the exit code — is exercised here: a wrong halo
exchange must not print a throughput."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(scenario, world=2, extra=(), grid=("64", "48")):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_bench_worker.py"), scenario, "--gpus", str(world),
                                       "--steps", "20", "--warmup", "5", "--nx", grid[0], "--ny", grid[1], *extra],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    return [p.returncode for p in procs], outs


@pytest.mark.timeout(600)
def test_two_rank_bench_line_and_exit_code():
    rcs, outs = run_bench("ok")
    assert rcs == [0, 0], outs
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5 and line["scaling"] == "strong" and line["value"] > 0
    st = line["strips"]
    assert st["parity"] == "bit-equal" and st["nranks"] == 2 and len(st["per_rank"]) == 2
    for key in ("host_issue_us_per_iteration", "gpu_us_per_iteration", "graph", "halo_bytes_per_face_and_exchange",
                "ms_per_step_compute_only", "ms_per_step_exchange_exposed", "schedule"):
        assert key in st, key
    assert st["halo_bytes_per_face_and_exchange"] == 1783296 and st["graph"].startswith("off") and line["roofline"]["frac"] > 0 and "cpu_baseline" not in line


@pytest.mark.timeout(600)
def test_a_strip_mismatch_voids_the_line_on_every_rank():
    rcs, outs = run_bench("mismatch")
    assert rcs == [3, 3], outs
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
    assert line["value"] is None and line["value_unverified"] > 0 and line["strips"]["parity"].startswith("MISMATCH: rank 1")
    assert "INVALID" in outs[0][1]


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("nx,ny,precision,label", [(8192, 2048, "f64", "configs[3]"), (16384, 4096, "f32", "configs[4]")])
def test_the_two_eight_gpu_configs_of_baseline_json_run_end_to_end(nx, ny, precision, label):
    """VERDICT r04 #7b: `bench.py --gpus N --nx 8192 --ny 2048` and `--nx 16384 --ny 4096 --precision f32` are the two 8-GPU
    configurations BASELINE.json names; the first run of either is the driver's. Here on two gloo ranks with the compute-free
    stand-in: the workload label names the config, the strips split the named grid (strong scaling), the parity window compares
    every row of the full-size grid (the gather and the checksum path at 2 x 2.4 GB), dtype and metric follow the precision, exit 0."""
    rcs, outs = run_bench("ok", extra=("--precision", precision), grid=(str(nx), str(ny)))
    assert rcs == [0, 0], [o[1][-2000:] for o in outs]
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
    assert label in line["config"]["workload"] and f"{nx}x{ny} {precision}" in line["config"]["workload"]
    assert line["dtype"] == precision and line["metric"] == f"MLUPS ({'fp64' if precision == 'f64' else 'fp32'})"
    assert line["config"]["rows_per_gpu"] == ny // 2 and line["config"]["decomposition"] == "2 row strip(s)" and line["scaling"] == "strong"
    st = line["strips"]
    assert st["parity"] == "bit-equal" and [r["rows"] for r in st["per_rank"]] == [ny // 2, ny // 2]
    assert line["value"] > 0 and line["roofline"]["algorithmic_bytes_per_launch"] > 0


@pytest.mark.timeout(900)
def test_eight_ranks_the_shape_of_the_drivers_scaling_run():
    """The driver's scaling run ends at N = 8 (`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...`): the same control flow with
    eight gloo ranks and the stand-in — one line from rank 0, eight strips of the named grid, the parity gather over eight ranks, every
    rank leaving with rank 0's verdict."""
    rcs, outs = run_bench("ok", world=8, grid=("64", "192"))
    assert rcs == [0] * 8, [o[1][-500:] for o in outs]
    lines = [l for o in outs for l in o[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["config"]["rows_per_gpu"] == 24 and line["config"]["decomposition"] == "8 row strip(s)"
    st = line["strips"]
    assert st["nranks"] == 8 and st["parity"] == "bit-equal" and [r["rank"] for r in st["per_rank"]] == list(range(8))
    rcs, outs = run_bench("mismatch", world=8, grid=("64", "192"))
    assert rcs == [3] * 8, [o[1][-300:] for o in outs]


@pytest.mark.timeout(600)
def test_weak_scaling_keeps_the_rows_per_gpu():
    """`--scaling weak`: every rank keeps `--ny` rows (the lattice grows with N), the line says "weak" and is not labelled as a
    BASELINE.json configuration (the metric is quoted on the fixed 4096x1024 grid: strong scaling)."""
    rcs, outs = run_bench("ok", world=2, extra=("--scaling", "weak"), grid=("64", "48"))
    assert rcs == [0, 0], [o[1][-500:] for o in outs]
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
    assert line["scaling"] == "weak" and line["config"]["ny"] == 96 and line["config"]["rows_per_gpu"] == 48
    assert "not a BASELINE.json config" in line["config"]["workload"] and line["strips"]["parity"] == "bit-equal"

