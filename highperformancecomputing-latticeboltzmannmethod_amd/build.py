"""Builds csrc/liblbm_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

The library carries the SHA-256 of its own sources (csrc/* and include/lbm_hip.h) as `lbm_build_id()`; build_all()
rebuilds whenever the id embedded in the .so on disk differs from the hash of the sources in the tree, so a stale
binary (the .so is git-ignored but travels with the working tree) can never be mistaken for HEAD."""
import hashlib
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "liblbm_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
_MARK = b"LBM_BUILD_ID="


def _sources():
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h")))
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "lbm_hip.h"))
    return srcs


def source_id():
    """SHA-256 (first 16 hex digits) over the names and contents of the library's sources."""
    h = hashlib.sha256()
    for s in _sources():
        h.update(os.path.basename(s).encode() + b"\0")
        with open(s, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def embedded_id(path=LIB):
    """The id baked into a built library (read from the file, no dlopen), or None."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        m = re.search(_MARK + rb"([0-9a-f]{16})", f.read())
    return m.group(1).decode() if m else None


def build_all(force=False, verbose=False):
    hipcc = shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")
    want = source_id()
    if force or embedded_id() != want:
        # -ffp-contract=off: no fused multiply-add is formed behind the source's back, so every formulation of the
        # step kernel (site / vector / fused, any layout) and the strict-IEEE oracle evaluate the same operation sequence.
        # Five objects compiled side by side (the column kernel's instantiations of one element type take as long as all
        # the other kernels together), then linked.
        units = [("lbm_hip.hip", ["-DLBM_BUILD_ID_STR=\"" + want + "\""], "lbm_hip.o"),
                 ("lbm_col.hip", ["-DLBM_COL_T=double"], "lbm_col_f64.o"), ("lbm_col.hip", ["-DLBM_COL_T=float"], "lbm_col_f32.o"),
                 ("lbm_col.hip", ["-DLBM_COL_TALL=1"], "lbm_col_tall_c.o"), ("lbm_col.hip", ["-DLBM_COL_TALL=0"], "lbm_col_tall_s.o")]
        common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-fPIC", "-pthread", "-c"]
        procs, objs = [], []
        for src, extra, oname in units:
            obj = os.path.join(CSRC, oname)
            cmd = common + extra + ["-o", obj, os.path.join(CSRC, src)]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(obj)
        for cmd, p in procs:
            if p.wait() != 0:
                raise subprocess.CalledProcessError(p.returncode, cmd)
        tmp = LIB + ".tmp"
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", tmp] + objs + \
               ["-L" + os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
        for o in objs:
            os.remove(o)
        os.replace(tmp, LIB)
        assert embedded_id() == want, "the built library does not carry the expected build id"
    return LIB
