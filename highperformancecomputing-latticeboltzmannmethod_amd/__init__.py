"""MI355X-native D2Q9-BGK lattice-Boltzmann timestep (drop-in for the hot path of
LGMOak/HighPerformanceComputing-LatticeBoltzmannMethod: the loop body of LBM::Solver::run, LBMSolver.h:48-76).

The product is csrc/ (hand-written HIP kernels for gfx950 + the extern "C" layer of include/lbm_hip.h, built
into csrc/liblbm_hip.so) and host/ (the C++20 mirror of the reference's Solver/Grid/IOManager surface).
This Python package is a thin ctypes view of the same C-ABI used by tests/ and bench.py; it contains no
numerics of its own and fails loudly if the HIP library is missing.
"""
from .binding import (LbmError, Params, Context, Group, lib, lib_path, device_count, build_id, runtime_versions, device_memory)  # noqa: F401
from .build import build_all, source_id, embedded_id  # noqa: F401
from .strips import partition_rows, GlooHalo  # noqa: F401
