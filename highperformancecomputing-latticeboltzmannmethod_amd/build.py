"""Builds csrc/liblbm_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "liblbm_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_all(force=False, verbose=False):
    hipcc = shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "lbm_hip.h"))
    if force or _newer(LIB, srcs):
        # -ffp-contract=off: no fused multiply-add is formed behind the source's back, so every formulation of the
        # step kernel (site / vector, any layout) and the strict-IEEE oracle evaluate the same operation sequence
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-fPIC", "-shared",
               "-o", LIB, os.path.join(CSRC, "lbm_hip.hip"),
               "-L" + os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB
