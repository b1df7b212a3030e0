"""oracle/oracle.py — TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liblbm_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (the HIP library behind include/lbm_hip.h) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblbm_oracle.so")


class Params(C.Structure):
    """Field-for-field the physics part of LBM::SimulationParams (/root/reference/include/LBMConfig.h:36-51)."""
    _fields_ = [("tau", C.c_double), ("inlet_velocity", C.c_double), ("nx", C.c_int), ("ny", C.c_int),
                ("cylinder_x", C.c_double), ("cylinder_y", C.c_double), ("cylinder_radius", C.c_double)]


def build(force=False):
    src = os.path.join(HERE, "lbm_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, LIB_PATH])
    return LIB_PATH


_lib = None


def host_cores():
    """Usable host threads: scheduler affinity clipped by a cgroup CPU quota (the GPU box exposes 256 hardware
    threads but grants a 16-CPU share; an OpenMP team larger than the quota spins instead of computing)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
    except Exception:
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = C.CDLL(LIB_PATH)
        L.lbmo_set_threads.restype = None; L.lbmo_set_threads.argtypes = [C.c_int]
        if "OMP_NUM_THREADS" not in os.environ:
            L.lbmo_set_threads(host_cores())
        dp = C.POINTER(C.c_double)
        L.lbmo_create.restype = C.c_void_p
        L.lbmo_create.argtypes = [C.POINTER(Params), C.c_int, C.c_int]
        for name in ("destroy", "initialise", "collide", "exchange_physical", "stream", "boundaries"):
            f = getattr(L, "lbmo_" + name); f.restype = None; f.argtypes = [C.c_void_p]
        for name in ("solid_count", "check_stability", "step"):
            f = getattr(L, "lbmo_" + name); f.restype = C.c_int; f.argtypes = [C.c_void_p]
        L.lbmo_run.restype = C.c_int; L.lbmo_run.argtypes = [C.c_void_p, C.c_int]
        L.lbmo_forces.restype = None; L.lbmo_forces.argtypes = [C.c_void_p, dp, dp]
        L.lbmo_get_edge_row.restype = None; L.lbmo_get_edge_row.argtypes = [C.c_void_p, C.c_int, dp]
        L.lbmo_set_ghost_row.restype = None; L.lbmo_set_ghost_row.argtypes = [C.c_void_p, C.c_int, dp]
        L.lbmo_max_velocity_sq.restype = C.c_double; L.lbmo_max_velocity_sq.argtypes = [C.c_void_p]
        for name in ("rho", "ux", "uy", "f_current", "f_next"):
            f = getattr(L, "lbmo_" + name); f.restype = dp; f.argtypes = [C.c_void_p]
        L.lbmo_solid.restype = C.POINTER(C.c_ubyte); L.lbmo_solid.argtypes = [C.c_void_p]
        L.lbmo_threads.restype = C.c_int
        for name in ("nu", "reynolds"):
            f = getattr(L, "lbmo_" + name); f.restype = C.c_double; f.argtypes = [C.POINTER(Params)]
        for name in ("cylinder_x_cells", "cylinder_y_cells", "cylinder_radius_cells"):
            f = getattr(L, "lbmo_" + name); f.restype = C.c_int; f.argtypes = [C.POINTER(Params)]
        _lib = L
    return _lib


def make_params(nx, ny, tau=0.6, inlet_velocity=0.01333, cylinder_x=0.2, cylinder_y=0.5, cylinder_radius=0.05):
    return Params(tau, inlet_velocity, nx, ny, cylinder_x, cylinder_y, cylinder_radius)


class Oracle:
    """One domain (or one row strip [y_start, y_start+local_ny) of it) of the CPU restatement."""

    def __init__(self, params, y_start=0, local_ny=None):
        self.p = params
        self.nx, self.ny = params.nx, params.ny
        self.y_start = y_start
        self.local_ny = self.ny if local_ny is None else local_ny
        self.L = lib()
        self.h = self.L.lbmo_create(C.byref(params), y_start, self.local_ny)
        self.L.lbmo_initialise(self.h)
        self.t = 0

    def close(self):
        if self.h:
            self.L.lbmo_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # phases
    def collide(self): self.L.lbmo_collide(self.h)
    def exchange_physical(self): self.L.lbmo_exchange_physical(self.h)
    def stream(self): self.L.lbmo_stream(self.h)
    def boundaries(self): self.L.lbmo_boundaries(self.h)
    def stable(self): return bool(self.L.lbmo_check_stability(self.h))
    def solid_count(self): return self.L.lbmo_solid_count(self.h)

    def forces(self):
        fx, fy = C.c_double(), C.c_double()
        self.L.lbmo_forces(self.h, C.byref(fx), C.byref(fy))
        return fx.value, fy.value

    def coefficients(self, fx, fy):
        """LBMIO.h:171-178."""
        d_ref = 2.0 * self.L.lbmo_cylinder_radius_cells(C.byref(self.p))
        q = 0.5 * 1.0 * self.p.inlet_velocity * self.p.inlet_velocity * d_ref
        return (fx / q, fy / q) if q > 1e-12 else (0.0, 0.0)

    def step(self, output_frequency=None, forces_out=None):
        """Loop body of Solver::run (LBMSolver.h:49-60) for a whole-domain handle. Returns stable?"""
        self.collide()
        if output_frequency and self.t % output_frequency == 0 and forces_out is not None:
            fx, fy = self.forces()
            forces_out.append((self.t, fx, fy) + self.coefficients(fx, fy))
        self.exchange_physical()
        self.stream()
        self.boundaries()
        ok = self.stable()
        self.t += 1
        return ok

    def run(self, nsteps, output_frequency=None, forces_out=None):
        for _ in range(nsteps):
            if not self.step(output_frequency, forces_out):
                return self.t - 1
        return -1

    def max_velocity(self):
        return float(np.sqrt(self.L.lbmo_max_velocity_sq(self.h)))

    def edge_row(self, north):
        buf = np.empty(self.nx * 9, dtype=np.float64)
        self.L.lbmo_get_edge_row(self.h, int(north), buf.ctypes.data_as(C.POINTER(C.c_double)))
        return buf

    def set_ghost_row(self, north, buf):
        buf = np.ascontiguousarray(buf, dtype=np.float64)
        self.L.lbmo_set_ghost_row(self.h, int(north), buf.ctypes.data_as(C.POINTER(C.c_double)))

    # views (no copy)
    def _macro(self, name):
        ptr = getattr(self.L, "lbmo_" + name)(self.h)
        return np.ctypeslib.as_array(ptr, shape=(self.local_ny, self.nx))

    @property
    def rho(self): return self._macro("rho")
    @property
    def ux(self): return self._macro("ux")
    @property
    def uy(self): return self._macro("uy")

    def _f(self, name):
        ptr = getattr(self.L, "lbmo_" + name)(self.h)
        return np.ctypeslib.as_array(ptr, shape=(self.local_ny + 2, self.nx + 2, 9))

    @property
    def f_current(self): return self._f("f_current")
    @property
    def f_next(self): return self._f("f_next")

    @property
    def solid(self):
        return np.ctypeslib.as_array(self.L.lbmo_solid(self.h), shape=(self.local_ny, self.nx))
