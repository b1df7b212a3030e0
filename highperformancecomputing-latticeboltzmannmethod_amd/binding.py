"""ctypes view of include/lbm_hip.h (liblbm_hip.so). No numerics here; every call goes to the HIP library."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("LBM_HIP_LIBRARY") or os.path.join(HERE, "csrc", "liblbm_hip.so")   # (override: diagnostic builds)


class LbmError(RuntimeError):
    pass


class Params(C.Structure):
    """struct lbm_params (include/lbm_hip.h) == the physics fields of LBM::SimulationParams + the strip."""
    _fields_ = [("tau", C.c_double), ("inlet_velocity", C.c_double), ("nx", C.c_int), ("ny", C.c_int),
                ("cylinder_x", C.c_double), ("cylinder_y", C.c_double), ("cylinder_radius", C.c_double),
                ("y_start", C.c_int), ("local_ny", C.c_int), ("precision", C.c_int),
                ("force_log_capacity", C.c_int)]


class ForceRow(C.Structure):
    _fields_ = [("timestep", C.c_int), ("fx", C.c_double), ("fy", C.c_double)]


_lib = None


def lib_path():
    return _LIB_PATH


def lib():
    """Loads liblbm_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise LbmError(f"{_LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        vp = C.c_void_p
        L.lbm_last_error.restype = C.c_char_p
        L.lbm_device_count.restype = C.c_int
        L.lbm_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(vp)]
        L.lbm_destroy.argtypes = [vp]; L.lbm_destroy.restype = None
        L.lbm_initialise.argtypes = [vp, C.POINTER(C.c_int)]
        L.lbm_step.argtypes = [vp, C.c_int, C.c_int]
        L.lbm_sync.argtypes = [vp]
        L.lbm_steps_done.argtypes = [vp]
        L.lbm_first_unstable_step.argtypes = [vp, C.POINTER(C.c_int)]
        L.lbm_get_forces.argtypes = [vp, dp, dp]
        L.lbm_drain_force_log.argtypes = [vp, C.POINTER(ForceRow), C.c_int]
        L.lbm_get_macros.argtypes = [vp, dp, dp, dp]
        L.lbm_max_velocity_sq.argtypes = [vp, dp]
        L.lbm_get_populations.argtypes = [vp, C.c_int, dp]
        L.lbm_set_f_current.argtypes = [vp, dp]
        L.lbm_get_solid.argtypes = [vp, C.POINTER(C.c_ubyte)]
        L.lbm_comm_unique_id.argtypes = [vp]
        L.lbm_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
        L.lbm_comm_allreduce.argtypes = [vp, dp, C.c_int, C.c_int]
        pp = C.POINTER(vp)
        L.lbm_group_link.argtypes = [pp, C.c_int, C.c_int]
        L.lbm_group_initialise.argtypes = [pp, C.c_int, C.POINTER(C.c_int)]
        L.lbm_group_step.argtypes = [pp, C.c_int, C.c_int, C.c_int]
        L.lbm_group_refresh_halos.argtypes = [pp, C.c_int]
        L.lbm_halo_export.argtypes = [vp, dp, dp]
        L.lbm_halo_import.argtypes = [vp, dp, dp]
        L.lbm_set_option.argtypes = [vp, C.c_char_p, C.c_long]
        L.lbm_save_state.argtypes = [vp, C.c_char_p]
        L.lbm_load_state.argtypes = [vp, C.c_char_p]
        L.lbm_last_step_kernel_ms.argtypes = [vp, dp]
        L.lbm_last_step_stats.argtypes = [vp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lbm_graph_replays.restype = C.c_long; L.lbm_graph_replays.argtypes = [vp]
        L.lbm_kernel_name.argtypes = [vp]; L.lbm_kernel_name.restype = C.c_char_p
        L.lbm_plan.argtypes = [vp]; L.lbm_plan.restype = C.c_char_p
        L.lbm_plan_options.argtypes = [vp]; L.lbm_plan_options.restype = C.c_char_p
        L.lbm_build_id.restype = C.c_char_p
        L.lbm_runtime_versions.argtypes = [C.POINTER(C.c_int)] * 3
        L.lbm_strip_schedule.argtypes = [vp]; L.lbm_strip_schedule.restype = C.c_char_p
        L.lbm_device_memory.argtypes = [C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
        _lib = L
    return _lib


def device_count():
    return lib().lbm_device_count()


def runtime_versions():
    """{'rccl': .., 'hip_runtime': .., 'hip_driver': ..} as bound by THIS process (ncclGetVersion etc.)."""
    r, h, d = C.c_int(), C.c_int(), C.c_int()
    if lib().lbm_runtime_versions(C.byref(r), C.byref(h), C.byref(d)) < 0:
        raise LbmError(lib().lbm_last_error().decode())
    return {"rccl": r.value, "hip_runtime": h.value, "hip_driver": d.value}


def device_memory(device=0):
    """(free, total) bytes of a device (hipMemGetInfo)."""
    f, t = C.c_ulonglong(), C.c_ulonglong()
    if lib().lbm_device_memory(device, C.byref(f), C.byref(t)) < 0:
        raise LbmError(lib().lbm_last_error().decode())
    return f.value, t.value


def build_id():
    """Source hash the loaded library was compiled from (lbm_build_id)."""
    return lib().lbm_build_id().decode()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


class Context:
    """One strip of the lattice on one GPU (struct lbm_ctx)."""

    def __init__(self, nx, ny, tau=0.6, inlet_velocity=0.01333, cylinder_x=0.2, cylinder_y=0.5,
                 cylinder_radius=0.05, y_start=0, local_ny=0, precision="f64", device=0, force_log_capacity=0,
                 options=None):
        self.L = lib()
        self.params = Params(tau, inlet_velocity, nx, ny, cylinder_x, cylinder_y, cylinder_radius, y_start,
                             local_ny, {"f64": 0, "f32": 1}[precision], force_log_capacity)
        self.nx, self.ny = nx, ny
        self.y_start = y_start
        self.local_ny = local_ny if local_ny > 0 else ny - y_start
        self.h = C.c_void_p()
        self._chk(self.L.lbm_create(C.byref(self.params), device, C.byref(self.h)))
        self.solid_count = None
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def _chk(self, rc):
        if rc < 0:
            raise LbmError(f"lbm_hip error {rc}: {self.L.lbm_last_error().decode()}")
        return rc

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.lbm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_option(self, key, value):
        self._chk(self.L.lbm_set_option(self.h, key.encode(), int(value)))

    def initialise(self):
        n = C.c_int()
        self._chk(self.L.lbm_initialise(self.h, C.byref(n)))
        self.solid_count = n.value
        return n.value

    def step(self, nsteps=1, output_frequency=0):
        self._chk(self.L.lbm_step(self.h, nsteps, output_frequency))

    def sync(self):
        self._chk(self.L.lbm_sync(self.h))

    @property
    def steps_done(self):
        return self.L.lbm_steps_done(self.h)

    def first_unstable_step(self):
        t = C.c_int()
        self._chk(self.L.lbm_first_unstable_step(self.h, C.byref(t)))
        return t.value

    def forces(self):
        fx, fy = C.c_double(), C.c_double()
        self._chk(self.L.lbm_get_forces(self.h, C.byref(fx), C.byref(fy)))
        return fx.value, fy.value

    def drain_force_log(self, max_rows=4096):
        rows = (ForceRow * max_rows)()
        n = self._chk(self.L.lbm_drain_force_log(self.h, rows, max_rows))
        return [(rows[k].timestep, rows[k].fx, rows[k].fy) for k in range(n)]

    def macros(self):
        shape = (self.local_ny, self.nx)
        rho, ux, uy = (np.empty(shape, dtype=np.float64) for _ in range(3))
        self._chk(self.L.lbm_get_macros(self.h, _dp(rho), _dp(ux), _dp(uy)))
        return rho, ux, uy

    def max_velocity_sq(self):
        v = C.c_double()
        self._chk(self.L.lbm_max_velocity_sq(self.h, C.byref(v)))
        return v.value

    def populations(self, which):
        """which: 'f_current' | 'f_next' -> [(local_ny+2), (nx+2), 9] like Grid::f_current(gx,gy,i)."""
        out = np.empty((self.local_ny + 2, self.nx + 2, 9), dtype=np.float64)
        self._chk(self.L.lbm_get_populations(self.h, {"f_current": 0, "f_next": 1}[which], _dp(out)))
        return out

    def set_f_current(self, aos):
        """Write side of Grid::f_current: [(local_ny+2), (nx+2), 9]; interior cells replace the pre-collision state."""
        a = np.ascontiguousarray(aos, dtype=np.float64)
        assert a.shape == (self.local_ny + 2, self.nx + 2, 9)
        self._chk(self.L.lbm_set_f_current(self.h, _dp(a)))

    def solid(self):
        m = np.empty((self.local_ny, self.nx), dtype=np.uint8)
        self._chk(self.L.lbm_get_solid(self.h, m.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return m

    # ---- strips ----
    def comm_unique_id(self):
        buf = (C.c_ubyte * 128)()
        self._chk(self.L.lbm_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, rank, nranks, id128):
        buf = (C.c_ubyte * 128).from_buffer_copy(id128)
        self._chk(self.L.lbm_comm_init(self.h, rank, nranks, buf))

    def allreduce(self, vals, op="sum"):
        a = np.ascontiguousarray(vals, dtype=np.float64)
        self._chk(self.L.lbm_comm_allreduce(self.h, _dp(a), a.size, {"sum": 0, "max": 1, "min": 2}[op]))
        return a

    HALO_ROWS = 6   # LBM_HALO_ROWS

    def halo_export(self, south=True, north=True):
        """(south_out, north_out): my bottom / top HALO_ROWS interior rows, each [HALO_ROWS, 9, nx]."""
        s = np.empty((self.HALO_ROWS, 9, self.nx), dtype=np.float64) if south else None
        n = np.empty((self.HALO_ROWS, 9, self.nx), dtype=np.float64) if north else None
        self._chk(self.L.lbm_halo_export(self.h, _dp(s), _dp(n)))
        return s, n

    def halo_import(self, south=None, north=None):
        s = np.ascontiguousarray(south, dtype=np.float64) if south is not None else None
        n = np.ascontiguousarray(north, dtype=np.float64) if north is not None else None
        self._chk(self.L.lbm_halo_import(self.h, _dp(s), _dp(n)))

    def save_state(self, path):
        self._chk(self.L.lbm_save_state(self.h, os.fspath(path).encode()))

    def load_state(self, path):
        self._chk(self.L.lbm_load_state(self.h, os.fspath(path).encode()))

    def last_step_stats(self):
        """(device ms of the last step() call, step-kernel launches it issued, iterations it advanced)."""
        ms, nl, ni = C.c_double(), C.c_int(), C.c_int()
        self._chk(self.L.lbm_last_step_stats(self.h, C.byref(ms), C.byref(nl), C.byref(ni)))
        return ms.value, nl.value, ni.value

    def graph_replays(self):
        """Replays of the captured launch-group graph so far (strips with a device transport on a deep plan)."""
        return int(self.L.lbm_graph_replays(self.h))

    def last_step_kernel_ms(self):
        v = C.c_double()
        self._chk(self.L.lbm_last_step_kernel_ms(self.h, C.byref(v)))
        return v.value

    def strip_schedule(self):
        return self.L.lbm_strip_schedule(self.h).decode()

    def kernel_name(self):
        return self.L.lbm_kernel_name(self.h).decode()

    def plan(self):
        return self.L.lbm_plan(self.h).decode()

    def plan_options(self):
        """The plan as {option: value} (lbm_plan_options): with tune=0 it pins the same plan on another context."""
        return {k: int(v) for k, v in (kv.split("=") for kv in self.L.lbm_plan_options(self.h).decode().split())}


class Group:
    """n strips of one lattice driven in lockstep by this process (lbm_group_*): one Context per strip, bottom to top.
    transport: "peer" (device copies / hipMemcpyPeerAsync) or "rccl" (ncclCommInitAll; distinct devices)."""

    def __init__(self, nx, ny, bounds, devices=None, transport="peer", options=None, **kw):
        from .strips import partition_rows
        if isinstance(bounds, int):
            bounds = partition_rows(ny, bounds)
        devices = devices or [0] * len(bounds)
        self.nx, self.ny = nx, ny
        self.ctxs = [Context(nx, ny, y_start=y0, local_ny=n, device=d, options=options, **kw)
                     for (y0, n), d in zip(bounds, devices)]
        self.L = lib()
        self._arr = (C.c_void_p * len(self.ctxs))(*[c.h for c in self.ctxs])
        self._n = len(self.ctxs)
        self._chk(self.L.lbm_group_link(self._arr, self._n, {"peer": 0, "rccl": 1}[transport]))
        self.solid_count = None

    def _chk(self, rc):
        if rc < 0:
            raise LbmError(f"lbm_hip error {rc}: {self.L.lbm_last_error().decode()}")
        return rc

    def initialise(self):
        n = C.c_int()
        self._chk(self.L.lbm_group_initialise(self._arr, self._n, C.byref(n)))
        self.solid_count = n.value
        return n.value

    def step(self, nsteps=1, output_frequency=0):
        self._chk(self.L.lbm_group_step(self._arr, self._n, nsteps, output_frequency))

    def refresh_halos(self):
        self._chk(self.L.lbm_group_refresh_halos(self._arr, self._n))

    def sync(self):
        for c in self.ctxs:
            c.sync()

    @property
    def steps_done(self):
        return self.ctxs[0].steps_done

    def first_unstable_step(self):
        """min over the strips (the reference's MPI_Allreduce(MIN) of the stability flag, LBMGrid.h:315)."""
        bad = [t for t in (c.first_unstable_step() for c in self.ctxs) if t >= 0]
        return min(bad) if bad else -1

    def macros(self):
        """(rho, ux, uy) of the whole lattice: the strips' rows concatenated by y_start (LBMSolver.h:340-357)."""
        parts = [c.macros() for c in self.ctxs]
        return tuple(np.concatenate([p[j] for p in parts], axis=0) for j in range(3))

    def populations(self, which):
        """Ghost-inclusive [(ny+2), (nx+2), 9]: interior rows of every strip + the physical ghost rows of the end strips."""
        parts = [c.populations(which) for c in self.ctxs]
        rows = [parts[0][:1]] + [p[1:-1] for p in parts] + [parts[-1][-1:]]
        return np.concatenate(rows, axis=0)

    def drain_force_log(self):
        """Rows summed over the strips (the reference's MPI_Reduce(SUM), LBMIO.h:167-168)."""
        logs = [c.drain_force_log() for c in self.ctxs]
        return [(logs[0][k][0], sum(l[k][1] for l in logs), sum(l[k][2] for l in logs)) for k in range(len(logs[0]))]

    def forces(self):
        f = [c.forces() for c in self.ctxs]
        return sum(v[0] for v in f), sum(v[1] for v in f)

    def max_velocity_sq(self):
        return max(c.max_velocity_sq() for c in self.ctxs)

    def close(self):
        for c in self.ctxs:
            c.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
