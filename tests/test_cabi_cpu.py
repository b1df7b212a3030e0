"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol that
include/lbm_hip.h declares, and refuses to run without a GPU (no CPU fallback). No compute calls here."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "highperformancecomputing-latticeboltzmannmethod_amd"


@pytest.fixture(scope="module")
def pkg():
    p = importlib.import_module(PKG)
    p.build_all()
    return p


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lbm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for s in ("lbm_create", "lbm_destroy", "lbm_initialise", "lbm_step", "lbm_get_macros", "lbm_get_forces",
              "lbm_get_populations", "lbm_first_unstable_step", "lbm_max_velocity_sq", "lbm_comm_init",
              "lbm_halo_export", "lbm_halo_import"):
        assert s in syms


def test_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg.lib_path())
    for s in declared_symbols():
        assert hasattr(lib, s), f"liblbm_hip.so does not export {s}"


def test_library_carries_the_hash_of_its_sources(pkg):
    """build() ties the binary to the tree: the id embedded in the .so (readable without dlopen) == SHA-256 of csrc/* and
    include/lbm_hip.h == what lbm_build_id() returns; a stale .so is rebuilt (build.py)."""
    assert re.fullmatch(r"[0-9a-f]{16}", pkg.source_id())
    assert pkg.embedded_id() == pkg.source_id() == pkg.build_id()


def test_no_cpu_fallback(pkg):
    """Without a HIP device lbm_create must fail with LBM_ERR_HIP; with one this test is vacuous."""
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.LbmError, match="no HIP device"):
        pkg.Context(64, 32)


def test_product_sources_never_touch_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may reference it."""
    bad = []
    for base in (os.path.join(ROOT, PKG), os.path.join(ROOT, "include")):
        for d, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    txt = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"lbm_oracle|oracle\.oracle|liblbm_oracle|lbmo_", txt):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_argument_errors_are_reported_without_a_device(pkg):
    """Error behaviour of the boundary (no exceptions across the C-ABI: negative codes + lbm_last_error text)."""
    import ctypes as C
    L = pkg.lib()
    h = C.c_void_p()
    assert L.lbm_create(None, 0, C.byref(h)) == -1 and b"null" in L.lbm_last_error()
    p = pkg.Params(0.6, 0.01, 0, 32, 0.2, 0.5, 0.05, 0, 0, 0, 0)           # nx = 0
    assert L.lbm_create(C.byref(p), 0, C.byref(h)) == -1 and b"bad nx/ny/tau" in L.lbm_last_error()
    p = pkg.Params(0.6, 0.01, 64, 32, 0.2, 0.5, 0.05, 0, 0, 7, 0)          # unknown precision
    assert L.lbm_create(C.byref(p), 0, C.byref(h)) == -1 and b"precision" in L.lbm_last_error()
    assert L.lbm_step(None, 1, 0) == -1
    assert L.lbm_steps_done(None) == -1
    assert L.lbm_set_option(None, b"tune", 0) == -1
    L.lbm_destroy(None)                                                     # destroying nothing is a no-op


def test_strip_schedule_pins_are_agreed_over_the_ranks(pkg):
    """the schedule trials of lbm_initialise are collective, so ranks with different overlap / deep_halo pins must get
    an error on every rank, not a different number of trials (a hang in RCCL). lbm_debug_strip_pins runs the very decision
    functions of tune_strip_schedule (strip_pins_pack -> MIN over the ranks -> strip_pins_agree) without a device."""
    lib = ctypes.CDLL(pkg.lib_path())
    lib.lbm_debug_strip_pins.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]

    def agree(ranks):
        flat = (ctypes.c_int * (5 * len(ranks)))(*[v for r in ranks for v in r])
        out = (ctypes.c_int * 5)()
        rc = lib.lbm_debug_strip_pins(flat, len(ranks), out)
        return rc, list(out)

    # nobody pins anything: tune, nothing pinned
    assert agree([(1, 0, 1, 0, 1)] * 4) == (0, [1, 0, -1, 0, -1])
    # everybody pins overlap=2: agreed, the deep-halo half is still measured
    assert agree([(1, 1, 2, 0, 1)] * 3) == (0, [1, 1, 2, 0, -1])
    # one strip too short to tune: nobody tunes
    rc, out = agree([(1, 0, 1, 0, 1), (0, 0, 1, 0, 1)])
    assert rc == 0 and out[0] == 0
    # one rank pins overlap, the other does not / pins another value: error on every rank
    assert agree([(1, 1, 1, 0, 1), (1, 0, 1, 0, 1)])[0] < 0
    assert agree([(1, 1, 1, 0, 1), (1, 1, 0, 0, 1)])[0] < 0
    assert agree([(1, 0, 1, 1, 0), (1, 0, 1, 1, 1)])[0] < 0
    # both halves pinned alike everywhere
    assert agree([(1, 1, 0, 1, 1)] * 8) == (0, [1, 1, 0, 1, 1])
