"""The launch choreography of a strip run, checked exhaustively on the CPU (no GPU).

Round 4's edge-band race (commit 0223fbc: a one-iteration remainder launch kept a six-row edge band while twelve rows travelled, so the
interior launch wrote rows the neighbour's pull was reading) was caught by a 1-in-15 flake of a GPU test. The geometry and the event
pattern are pure host arithmetic, so the class gets a deterministic detector instead: `lbm_debug_choreography` (include/lbm_hip.h) runs
the very functions that issue a launch group — plan_launch, issue_before, the exchanges, issue_after, join_comm — on contexts without
a device, recording every kernel, event record, cross-stream wait, copy, send and receive, and replays the record with vector clocks:
  RACE  = two accesses to the same row of the same buffer, at least one a write, that no event orders
          (every row the exchange reads must be written by a launch ordered before ev_edge; every ghost row a launch reads must have
          been received behind ev_comm; nobody overwrites a row a neighbour's pull may still be reading);
  STALE = a launch reads, inside its dependency cone, a row that does not hold the iteration it starts from.
It replaces what the reference gets for free from MPI_Waitall before unpack_received_data (/root/reference/include/LBMGrid.h:278-283).
The enumeration: every plan depth (1-8 iterations per launch: the tile kernels, the LDS shapes deep 1/2/3, the register shapes deep
6/7/9 and, fp32, 8), both arithmetic modes (their regions differ: 64x32 / 64x24), deep_halo 0/1/2, overlap 0/1/2, calls that end on
every remainder depth and cross force outputs, strips of 12 to 600 rows in every position (bottom, middle, top: face present / absent),
all four transports.
"""
import ctypes as C
import importlib
import itertools

import pytest

PKG = "highperformancecomputing-latticeboltzmannmethod_amd"


@pytest.fixture(scope="module")
def dry():
    pkg = importlib.import_module(PKG)
    pkg.build_all()
    L = C.CDLL(pkg.lib_path())
    L.lbm_debug_choreography.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_int,
                                         C.c_int, C.c_char_p, C.c_int]
    L.lbm_last_error.restype = C.c_char_p
    out = C.create_string_buffer(1 << 16)

    def run(nx, ny, bounds, transport, options, calls, precision=0, dump=0):
        b = (C.c_int * (2 * len(bounds)))(*[v for p in bounds for v in p])
        cl = (C.c_int * (2 * len(calls)))(*[v for p in calls for v in p])
        rc = L.lbm_debug_choreography(nx, ny, b, len(bounds), precision, transport, " ".join(f"{k}={v}" for k, v in options.items()).encode(), cl,
                                      len(calls), dump, out, len(out))
        return rc, (out.value.decode() if rc >= 0 else L.lbm_last_error().decode())
    return run


# (plan options, precision): one entry per kernel family, depth and region shape
PLANS = [(dict(fuse=1), 0), (dict(fuse=2, pair_ty=8), 0), (dict(fuse=3, pair_ty=12), 0), (dict(fuse=3, pair_ty=8), 0),
         (dict(deep=1), 0), (dict(deep=2), 0), (dict(deep=3), 0)] + \
        [(dict(deep=d, arith=a), 0) for d in (6, 7, 9) for a in (0, 1)] + [(dict(deep=8, arith=a), 1) for a in (0, 1)]
CALLS = [[(1, 0)], [(2, 0)], [(6, 0)], [(7, 0)], [(12, 0)], [(19, 0)], [(20, 0)], [(24, 0)], [(13, 0), (6, 0)], [(31, 7)], [(50, 13)], [(64, 8)], [(97, 0)], [(97, 31), (5, 0)]]


def strips_of(heights):
    b, y = [], 0
    for h in heights:
        b.append((y, h))
        y += h
    return b, y


def geometries():
    """(transport, bounds, ny): groups of three strips (bottom / middle / top) with peer copies and over RCCL, one strip as a rank of a
    multi-process run in each position, one strip exchanging with itself."""
    out = []
    for hs in [(12, 12, 12), (13, 24, 17), (22, 23, 45), (36, 14, 64), (44, 96, 28), (128, 50, 128), (191, 192, 200), (600, 12, 100)]:
        b, ny = strips_of(hs)
        out.append((0, b, ny))
    for hs in [(12, 13, 14), (64, 24, 128), (200, 36, 96)]:
        b, ny = strips_of(hs)
        out.append((1, b, ny))
    for h in list(range(12, 80)) + [96, 128, 191, 192, 200, 256, 600]:
        out.append((2, [(h, h)], 3 * h))          # a middle rank
    for h in (12, 44, 128):
        out.append((2, [(0, h)], 2 * h))          # the bottom rank (no south face)
        out.append((2, [(h, h)], 2 * h))          # the top rank (no north face)
    for h in (24, 64, 128, 256):
        out.append((3, [(0, h)], h))
    return out


def test_every_schedule_orders_its_accesses_and_reads_fresh_rows(dry):
    """HEAD: no race and no stale read in the whole enumeration."""
    runs = 0
    for (plan, prec), dh, ov, tp in itertools.product(PLANS, (0, 1, 2), (0, 1, 2), (0, 1)):
        opts = dict(tune=0, nt=1, xcd=1, overlap=ov, deep_halo=dh, trailing_pair=tp, **plan)
        for transport, bounds, ny in geometries():
            for calls in CALLS:
                rc, text = dry(256, ny, bounds, transport, opts, calls, prec)
                runs += 1
                assert rc == 0, f"{opts} transport {transport} bounds {bounds} calls {calls}: rc {rc}\n{text[:3000]}"
    assert runs > 300000


def test_the_edge_band_rule_of_before_the_fix_is_flagged(dry):
    """The geometry as it was before commit 0223fbc (edge band = one band of tiles, not at least the rows that travel; option
    "debug_old_edge_band") is a race the detector names: the neighbour's pull of twelve rows against the interior launch of a
    one- to three-iteration remainder launch that writes rows 7-12."""
    base = dict(tune=0, nt=1, xcd=1, overlap=1, deep_halo=2, deep=7, arith=1)
    b, ny = strips_of((100, 100, 100))
    rc, text = dry(256, ny, b, 0, base, [(20, 0)])
    assert rc == 0, text
    rc, text = dry(256, ny, b, 0, dict(base, debug_old_edge_band=1), [(20, 0)])
    assert rc > 0 and "RACE strip 0 buffer" in text and "copy buf" in text and "main stream: kernel" in text, text
    # ... and the eight-row exchange of the eight-iteration LDS shape, and one rank of a multi-process run (its own send against its interior rows)
    rc, text = dry(256, 384, [(128, 128)], 2, dict(tune=0, nt=1, xcd=1, overlap=1, deep_halo=1, deep=3, debug_old_edge_band=1), [(20, 0)])
    assert rc > 0 and "send buf" in text, text
    rc, text = dry(256, 384, [(128, 128)], 2, dict(tune=0, nt=1, xcd=1, overlap=1, deep_halo=1, deep=3), [(20, 0)])
    assert rc == 0, text


def test_a_frame_that_is_not_refreshed_is_stale(dry):
    """The freshness half: with the exchange cut (the diagnostic option "skip_exchange") the second launch of a middle rank reads
    ghost rows that still hold the previous exchange's iteration — STALE, named with the launch that read them."""
    rc, text = dry(256, 384, [(128, 128)], 2, dict(tune=0, nt=1, xcd=1, overlap=0, deep_halo=1, deep=7, skip_exchange=1), [(20, 0)])
    assert rc > 0 and "STALE strip 0 buffer" in text and "kernel t=6" in text, text


def test_overwriting_edge_rows_under_a_neighbours_pull_is_flagged(dry):
    """A second negative control (option "debug_skip_pull_wait", test only): in a group with peer copies a strip's edge rows are READ by
    its neighbours' pulls, so before a launch overwrites them it must wait for the neighbours' ev_comm. Where launches come in pairs
    between exchanges (three iterations per launch, one exchange per two launches) nothing else orders the two: without the wait the
    checker names the write-after-read — the neighbour's copy against the launch that rewrites the rows two launches later. (With one
    exchange per launch on the side stream the order follows transitively from the pulls themselves, and the checker agrees: no race.)"""
    b, ny = strips_of((100, 100, 100))
    pairs = dict(tune=0, nt=1, xcd=1, overlap=0, deep_halo=1, fuse=3, pair_ty=12)
    rc, text = dry(256, ny, b, 0, pairs, [(40, 0)])
    assert rc == 0, text
    rc, text = dry(256, ny, b, 0, dict(pairs, debug_skip_pull_wait=1), [(40, 0)])
    assert rc > 0 and "RACE strip 0 buffer" in text and "copy buf" in text and "main stream: kernel" in text, text
    every = dict(tune=0, nt=1, xcd=1, overlap=1, deep_halo=1, deep=7, arith=1)
    assert dry(256, ny, b, 0, dict(every, debug_skip_pull_wait=1), [(40, 0)])[0] == 0


def test_the_ranks_of_a_multi_process_run_would_pair_up(dry):
    """exchange_rccl's multi-rank branch (`nranks > 1`: what replaces Grid::exchange_ghost_cells between processes,
    /root/reference/include/LBMGrid.h:249-283) has run on no hardware yet — it needs two GPUs. lbm_debug_p2p_matching runs its posting loops
    DRY on every rank of a run and pairs the transcripts the way RCCL pairs messages (the k-th send to a peer with the peer's k-th receive):
    same number of messages, same counts, same offset inside the block of edge / ghost rows on both ends, same number of exchanges on every
    rank — for two to eight ranks, even and uneven strips, every plan family, deep_halo 0/1/2, whole rows and the trimmed message, fp64 and
    fp32. Ranks that disagree on the message (one trims, one does not) are flagged: the case the pin agreement of lbm_initialise prevents."""
    pkg = importlib.import_module(PKG)
    L = C.CDLL(pkg.lib_path())
    L.lbm_debug_p2p_matching.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int,
                                         C.c_char_p, C.c_int]
    L.lbm_last_error.restype = C.c_char_p
    out = C.create_string_buffer(1 << 14)

    def match(nx, heights, options, calls, precision=0, rank1=None):
        b, ny = strips_of(heights)
        ba = (C.c_int * (2 * len(b)))(*[v for p in b for v in p])
        cl = (C.c_int * (2 * len(calls)))(*[v for p in calls for v in p])
        rc = L.lbm_debug_p2p_matching(nx, ny, ba, len(b), precision, " ".join(f"{k}={v}" for k, v in options.items()).encode(),
                                      " ".join(f"{k}={v}" for k, v in rank1.items()).encode() if rank1 else None, cl, len(calls), out, len(out))
        return rc, (out.value.decode() if rc >= 0 else L.lbm_last_error().decode())

    runs = 0
    for (plan, prec), dh, trim, ov in itertools.product(PLANS, (0, 1, 2), (0, 1), (0, 1)):
        opts = dict(tune=0, nt=1, xcd=1, overlap=ov, deep_halo=dh, halo_trim=trim, trailing_pair=1, **plan)
        for heights in ((128, 128), (40, 24, 100), (12, 13, 14, 200), (128,) * 8):
            for calls in ([(20, 0)], [(97, 31), (5, 0)]):
                rc, text = match(4096 if len(heights) == 8 else 320, heights, opts, calls, prec)
                runs += 1
                assert rc == 0, f"{opts} {heights} {calls}: rc {rc}\n{text[:2000]}"
    assert runs > 1000
    # the payload: six rows x nine sub-rows whole, 45 of them trimmed (nx = 4096 fp64: sub-rows of 4112 elements)
    rc, text = match(4096, (128,) * 8, dict(tune=0, nt=1, xcd=1, overlap=1, deep_halo=1, deep=1, halo_trim=0, trailing_pair=1), [(24, 0)])
    assert rc == 0 and "4 exchanges per rank, 8 messages posted by rank 0" in text, text
    rc, text = match(4096, (128,) * 8, dict(tune=0, nt=1, xcd=1, overlap=1, deep_halo=1, deep=1, halo_trim=1, trailing_pair=1), [(24, 0)])
    assert rc == 0 and "4 exchanges per rank, 40 messages posted by rank 0" in text, text
    # negative controls: rank 1 trims and the others do not; rank 1 exchanges twelve rows per two launches and the others six per launch
    rc, text = match(320, (64, 64, 64), dict(tune=0, nt=1, xcd=1, overlap=1, deep_halo=1, deep=7, halo_trim=0), [(24, 0)], rank1=dict(halo_trim=1))
    assert rc > 0 and "rank 1 posts" in text, text
    rc, text = match(320, (64, 64, 64), dict(tune=0, nt=1, xcd=1, overlap=1, deep_halo=1, deep=7, halo_trim=0), [(24, 0)], rank1=dict(deep_halo=2))
    assert rc > 0, text

