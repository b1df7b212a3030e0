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


def run_bench(scenario, world=2, extra=()):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_bench_worker.py"), scenario, "--gpus", str(world),
                                       "--steps", "20", "--warmup", "5", "--nx", "64", "--ny", "48", *extra],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
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
