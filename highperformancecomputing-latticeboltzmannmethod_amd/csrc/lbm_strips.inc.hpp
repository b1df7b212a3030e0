// csrc/lbm_strips.inc.hpp — strip halo exchange (RCCL send/recv, in-process groups), the choreography of one launch (edge bands, events, side stream), plan_launch / advance, state initialisation
// (part of the one host translation unit lbm_hip.hip, which includes it in this place; round 4 split a 2 100-line file by concern)
// ---- strip halo exchange ----------------------------------------------------------------------------------
// After a launch has produced the new populations in buf[dst]: my top GR interior rows go to the north neighbour's
// south ghost rows, my bottom GR interior rows to the south neighbour's north ghost rows, all nine populations
// (a fused launch recomputes up to two of the neighbour's rows, which needs every population; per lattice update
// this is the reference's 9 values per edge cell, LBMGrid.h:404-406). Strips always use the row-interleaved layout, in
// which GR rows x 9 sub-rows are ONE contiguous run of GR*pitch elements: one message per face, no packing.
// FaceSpans is the single place the offsets and the count are computed; every transport below uses it.
// rows per face and exchange / rows of each internal face the first launch of a pair recomputes: six and three, or — a deep plan
// with "deep_halo" 2 — twelve and six
inline bool device_transport(const lbm_ctx* c) { return c->group_n > 1 || (c->comm && c->nranks > 1) || c->loopback; }
inline bool deep_pairs(const lbm_ctx* c) { return c->deep && deep_depth(c->deep) <= HR1 && c->deep_halo == 2 && device_transport(c); }
// (a deep LDS shape of seven / eight iterations per launch refreshes seven / eight: the frame holds twelve)
inline int halo_rows(const lbm_ctx* c) {
    if (deep_pairs(c)) return 2 * HR1;
    if (c->deep && deep_depth(c->deep) > HR1 && deep_depth(c->deep) <= GR && device_transport(c)) return deep_depth(c->deep);
    return HR1;
}
inline int ext_rows(const lbm_ctx* c) { return halo_rows(c) / 2; }
struct FaceSpans {
    size_t cnt;       // elements per face message (halo_rows x pitch)
    long top_rows;    // my top HR interior rows      (gy = nyl+GR-HR .. nyl+GR-1)   -> north neighbour's ghost_s
    long bot_rows;    // my bottom HR interior rows   (gy = GR .. GR+HR-1)           -> south neighbour's ghost_n
    long ghost_n;     // my north ghost rows          (gy = nyl+GR .. nyl+GR+HR-1)
    long ghost_s;     // my south ghost rows          (gy = GR-HR .. GR-1)
};
inline FaceSpans face_spans(const lbm_ctx* c) {
    FaceSpans f;
    const int hr = halo_rows(c);
    f.cnt = (size_t)hr * c->pitch;
    f.top_rows = (long)(c->nyl + GR - hr) * c->pitch;
    f.bot_rows = (long)GR * c->pitch;
    f.ghost_n = (long)(c->nyl + GR) * c->pitch;
    f.ghost_s = (long)(GR - hr) * c->pitch;
    return f;
}

// ---- what is queued goes through these: the runtime call, or — dry run (lbm_ctx::rec) — an entry of the choreography record
#define QCHK(expr) do { const int q_ = (expr); if (q_) return q_; } while (0)
#define SETDEV(c) do { if (!(c)->rec) HIPCHK(hipSetDevice((c)->device)); } while (0)
#define LAUNCHED(c) do { if (!(c)->rec) HIPCHK(hipGetLastError()); } while (0)
inline int stream_id(const lbm_ctx* c, hipStream_t s) { return s == c->stream ? 0 : 1; }
inline int q_record(lbm_ctx* c, hipEvent_t e, hipStream_t s) {
    if (c->rec) {
        ChoreoOp o; o.kind = ChoreoOp::RECORD; o.strip = c->group_k; o.stream = stream_id(c, s); o.ev_strip = c->group_k;
        o.ev = e == c->ev_main ? 0 : e == c->ev_edge ? 1 : 2;
        c->rec->ops.push_back(o);
        return LBM_OK;
    }
    HIPCHK(hipEventRecord(e, s));
    return LBM_OK;
}
inline int q_wait(lbm_ctx* c, hipStream_t s, const lbm_ctx* owner, hipEvent_t e) {
    if (c->rec) {
        ChoreoOp o; o.kind = ChoreoOp::WAIT; o.strip = c->group_k; o.stream = stream_id(c, s); o.ev_strip = owner->group_k;
        o.ev = e == owner->ev_main ? 0 : e == owner->ev_edge ? 1 : 2;
        c->rec->ops.push_back(o);
        return LBM_OK;
    }
    HIPCHK(hipStreamWaitEvent(s, e, 0));
    return LBM_OK;
}
// rows of a FaceSpans offset (elements from the buffer's start; row-interleaved layout: a row is `pitch` elements)
inline int span_row(const lbm_ctx* c, long off) { return (int)(off / c->pitch) - GR; }
inline void rec_xfer(lbm_ctx* c, int kind, hipStream_t s, int buf, long write_off, long read_off, int rows, const lbm_ctx* from) {
    ChoreoOp o; o.kind = kind; o.strip = c->group_k; o.stream = stream_id(c, s); o.buf = buf;
    if (write_off >= 0) { o.w0[0] = span_row(c, write_off); o.w1[0] = o.w0[0] + rows; }
    if (read_off >= 0) { o.r0 = span_row(from ? from : c, read_off); o.r1 = o.r0 + rows; }
    o.r_strip = from ? from->group_k : -1;
    c->rec->ops.push_back(o);
}

// The TRIMMED message (round 5; option "halo_trim", measured like the rest of the schedule): what the receiver's launches really read
// of the hr rows of a face. Between two exchanges a strip advances at most hr iterations, so of ghost row g (1 = next to the strip)
// only the populations that can still reach the strip matter: rows 1 .. hr-2 all nine, row hr-1 all but the three that move AWAY from
// the strip (they would feed level 1 of row hr, which nobody computes), row hr the three that move TOWARDS it — 9 hr - 9 of the 9 hr
// sub-rows (45 of 54 at hr = 6: 17 % fewer bytes), in five runs of the row-interleaved layout [row][population][column]. Sub-rows that
// do not travel keep stale but finite values that no valid cell ever reads. Runs are in sub-rows, relative to the first sub-row of the
// block; `south_block`: the block lies BELOW the strip it borders (a sender's top rows seen from the receiver / a receiver's south
// ghost rows): its outermost row is the FIRST in memory and "towards the strip" is north (populations 2, 5, 6). Otherwise (a sender's
// bottom rows / a receiver's north ghost rows) the outermost row is the LAST and "towards the strip" is south (4, 7, 8).
struct FaceRuns { int n; int off[5], cnt[5]; };
inline FaceRuns face_runs(int hr, bool trim, bool south_block) {
    FaceRuns r{};
    if (!trim || hr < 2) { r.n = 1; r.off[0] = 0; r.cnt[0] = Q * hr; return r; }
    r.n = 5;
    if (south_block) {      // row 0 (outermost): {2}, {5,6}; row 1: {0..3}, {5,6}; rows 2 .. hr-1: everything
        const int o[5] = {2, 5, Q + 0, Q + 5, 2 * Q}, n[5] = {1, 2, 4, 2, Q * (hr - 2)};
        for (int k = 0; k < 5; ++k) { r.off[k] = o[k]; r.cnt[k] = n[k]; }
    } else {                // rows 0 .. hr-3: everything, then row hr-2: {0,1}, {3,4}, {7,8}; row hr-1 (outermost): {4}, {7,8}
        const int a = Q * (hr - 2), b = Q * (hr - 1);
        const int o[5] = {0, a + 3, a + 7, b + 4, b + 7}, n[5] = {a + 2, 2, 2, 1, 2};
        for (int k = 0; k < 5; ++k) { r.off[k] = o[k]; r.cnt[k] = n[k]; }
    }
    return r;
}
inline size_t face_payload_elems(const lbm_ctx* c) {      // elements per face and exchange that really travel
    const FaceRuns r = face_runs(halo_rows(c), c->halo_trim != 0, true);
    size_t n = 0;
    for (int k = 0; k < r.n; ++k) n += (size_t)r.cnt[k];
    return n * c->pitch0;
}

// Transports of ONE context: RCCL send/recv between processes (rank r <-> r-1, r+1), or the test-only loopbacks.
template <typename T>
int exchange_rccl(lbm_ctx* c, int dst, hipStream_t s) {
    if (c->skip_exchange) return LBM_OK;
    const FaceSpans f = face_spans(c);
    if (c->rec) {       // dry run: one rank of a multi-process run (its neighbours are other processes), or the device-copy loopback
        const int hr = halo_rows(c);
        if (c->loopback) {
            rec_xfer(c, ChoreoOp::COPY, s, dst, f.ghost_s, f.top_rows, hr, c);
            rec_xfer(c, ChoreoOp::COPY, s, dst, f.ghost_n, f.bot_rows, hr, c);
        } else if (c->nranks > 1) {
            if (c->rank + 1 < c->nranks) { rec_xfer(c, ChoreoOp::SEND, s, dst, -1, f.top_rows, hr, nullptr); rec_xfer(c, ChoreoOp::RECV, s, dst, f.ghost_n, -1, hr, nullptr); }
            if (c->rank > 0) { rec_xfer(c, ChoreoOp::SEND, s, dst, -1, f.bot_rows, hr, nullptr); rec_xfer(c, ChoreoOp::RECV, s, dst, f.ghost_s, -1, hr, nullptr); }
        }
        if (c->loopback || c->nranks <= 1) return LBM_OK;
        // (a rank of a multi-process run: the posting loops below run too — dry: what they would hand to RCCL goes to the transcript)
    }
    T* b = static_cast<T*>(c->buf[dst]);
    const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : ncclFloat;
    const int hr = halo_rows(c);
    const bool trim = c->halo_trim != 0;
    const size_t sub = (size_t)c->pitch0;                        // elements of one sub-row
    // my top rows and my south ghost rows are "south blocks" (below the strip they border), my bottom rows and north ghost rows are not
    const FaceRuns south_runs = face_runs(hr, trim, true), north_runs = face_runs(hr, trim, false);
    if (c->loopback) {   // test transports: my own edge rows become my ghost rows
        if (c->layout != 1) return fail(LBM_ERR_COMM, "loopback requires the row-interleaved layout");
        if (c->loopback == 2) {   // ... through RCCL itself: a one-rank communicator sending to / receiving from rank 0
            if (!c->comm) return fail(LBM_ERR_COMM, "loopback=2 needs lbm_comm_init(c, 0, 1, id)");
            NCCLCHK(ncclGroupStart());        // self send/recv pairs match in posting order
            for (int k = 0; k < south_runs.n; ++k) {
                NCCLCHK(ncclSend(b + f.top_rows + south_runs.off[k] * sub, south_runs.cnt[k] * sub, dt, 0, c->comm, s));
                NCCLCHK(ncclRecv(b + f.ghost_s + south_runs.off[k] * sub, south_runs.cnt[k] * sub, dt, 0, c->comm, s));
            }
            for (int k = 0; k < north_runs.n; ++k) {
                NCCLCHK(ncclSend(b + f.bot_rows + north_runs.off[k] * sub, north_runs.cnt[k] * sub, dt, 0, c->comm, s));
                NCCLCHK(ncclRecv(b + f.ghost_n + north_runs.off[k] * sub, north_runs.cnt[k] * sub, dt, 0, c->comm, s));
            }
            NCCLCHK(ncclGroupEnd());
            return LBM_OK;
        }
        for (int k = 0; k < south_runs.n; ++k)                   // ... or plain device copies on the same stream
            HIPCHK(hipMemcpyAsync(b + f.ghost_s + south_runs.off[k] * sub, b + f.top_rows + south_runs.off[k] * sub, south_runs.cnt[k] * sub * sizeof(T), hipMemcpyDeviceToDevice, s));
        for (int k = 0; k < north_runs.n; ++k)
            HIPCHK(hipMemcpyAsync(b + f.ghost_n + north_runs.off[k] * sub, b + f.bot_rows + north_runs.off[k] * sub, north_runs.cnt[k] * sub * sizeof(T), hipMemcpyDeviceToDevice, s));
        return LBM_OK;
    }
    if (c->nranks <= 1) return LBM_OK;
    if (c->layout != 1) return fail(LBM_ERR_COMM, "strips require the row-interleaved layout");
    // (the k-th send to a peer meets the peer's k-th receive from me: both sides walk the same runs in the same order — held on the CPU by
    // lbm_debug_p2p_matching, which runs these very loops dry on every rank of a run and pairs the transcripts)
    auto post = [&](int kind, long off, long cnt, int peer) -> int {
        if (c->rec) { c->rec->p2p.push_back({kind, peer, off, cnt}); return LBM_OK; }
        if (kind == 0) NCCLCHK(ncclSend(b + off, (size_t)cnt, dt, peer, c->comm, s));
        else NCCLCHK(ncclRecv(b + off, (size_t)cnt, dt, peer, c->comm, s));
        return LBM_OK;
    };
    if (!c->rec) NCCLCHK(ncclGroupStart());
    if (c->rank + 1 < c->nranks) {
        for (int k = 0; k < south_runs.n; ++k) QCHK(post(0, f.top_rows + (long)(south_runs.off[k] * sub), (long)(south_runs.cnt[k] * sub), c->rank + 1));
        for (int k = 0; k < north_runs.n; ++k) QCHK(post(1, f.ghost_n + (long)(north_runs.off[k] * sub), (long)(north_runs.cnt[k] * sub), c->rank + 1));
    }
    if (c->rank > 0) {
        for (int k = 0; k < north_runs.n; ++k) QCHK(post(0, f.bot_rows + (long)(north_runs.off[k] * sub), (long)(north_runs.cnt[k] * sub), c->rank - 1));
        for (int k = 0; k < south_runs.n; ++k) QCHK(post(1, f.ghost_s + (long)(south_runs.off[k] * sub), (long)(south_runs.cnt[k] * sub), c->rank - 1));
    }
    if (c->rec) c->rec->p2p.push_back({2, -1, 0, 0});
    else NCCLCHK(ncclGroupEnd());
    return LBM_OK;
}

inline hipStream_t exchange_stream(const lbm_ctx* c) { return c->overlap ? c->comm_stream : c->stream; }

// Transports of an in-process GROUP of strips (lbm_group_link): every member's exchange is issued by the one host
// thread that drives the group, after every member's edge rows have been queued.
//   peer : each strip PULLS its neighbours' edge rows into its own ghost rows (hipMemcpyPeerAsync over xGMI, a plain
//          device copy when both strips share a device) on its own exchange stream, behind the neighbour's ev_edge;
//   rccl : all members' ncclSend/ncclRecv inside ONE ncclGroupStart/End (one communicator per member, ncclCommInitAll).
// peer transport, one member: pull the neighbours' edge rows of buf[dst] into my ghost rows on my exchange stream
template <typename T>
int pull_halos(lbm_ctx** cs, int n, int k, int dst) {
    lbm_ctx* c = cs[k];
    if (c->skip_exchange) return LBM_OK;
    SETDEV(c);
    const FaceSpans f = face_spans(c);
    T* b = static_cast<T*>(c->buf[dst]);
    hipStream_t s = exchange_stream(c);
    const size_t sub = (size_t)c->pitch0;
    auto pull = [&](lbm_ctx* nb, long nb_rows, long my_ghost, bool south_block) -> int {
        const T* src = static_cast<const T*>(nb->buf[dst]) + nb_rows;
        QCHK(q_wait(c, s, nb, nb->ev_edge));            // the neighbour's edge rows of this launch are written
        if (c->rec) { rec_xfer(c, ChoreoOp::COPY, s, dst, my_ghost, nb_rows, halo_rows(c), nb); return LBM_OK; }
        const FaceRuns r = face_runs(halo_rows(c), c->halo_trim != 0, south_block);
        for (int q = 0; q < r.n; ++q) {
            const size_t o = r.off[q] * sub, bytes = r.cnt[q] * sub * sizeof(T);
            if (nb->device == c->device) HIPCHK(hipMemcpyAsync(b + my_ghost + o, src + o, bytes, hipMemcpyDeviceToDevice, s));
            else HIPCHK(hipMemcpyPeerAsync(b + my_ghost + o, c->device, src + o, nb->device, bytes, s));
        }
        return LBM_OK;
    };
    if (k > 0) { int rc = pull(cs[k - 1], face_spans(cs[k - 1]).top_rows, f.ghost_s, true); if (rc) return rc; }
    if (k + 1 < n) { int rc = pull(cs[k + 1], face_spans(cs[k + 1]).bot_rows, f.ghost_n, false); if (rc) return rc; }
    return LBM_OK;
}

template <typename T>
int exchange_group(lbm_ctx** cs, int n, int dst) {
    if (n < 2 || cs[0]->skip_exchange) return LBM_OK;
    if (cs[0]->group_transport == 1 && cs[0]->rec) {      // dry run: every member's sends (reads) and receives (writes) on its exchange stream
        for (int k = 0; k < n; ++k) {
            lbm_ctx* c = cs[k];
            const FaceSpans f = face_spans(c);
            const int hr = halo_rows(c);
            hipStream_t s = exchange_stream(c);
            if (k + 1 < n) { rec_xfer(c, ChoreoOp::SEND, s, dst, -1, f.top_rows, hr, nullptr); rec_xfer(c, ChoreoOp::RECV, s, dst, f.ghost_n, face_spans(cs[k + 1]).bot_rows, hr, cs[k + 1]); }
            if (k > 0) { rec_xfer(c, ChoreoOp::SEND, s, dst, -1, f.bot_rows, hr, nullptr); rec_xfer(c, ChoreoOp::RECV, s, dst, f.ghost_s, face_spans(cs[k - 1]).top_rows, hr, cs[k - 1]); }
        }
        return LBM_OK;
    }
    if (cs[0]->group_transport == 1) {
        const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : ncclFloat;
        NCCLCHK(ncclGroupStart());
        for (int k = 0; k < n; ++k) {
            lbm_ctx* c = cs[k];
            const FaceSpans f = face_spans(c);
            T* b = static_cast<T*>(c->buf[dst]);
            hipStream_t s = exchange_stream(c);
            const size_t sub = (size_t)c->pitch0;
            const FaceRuns sr = face_runs(halo_rows(c), c->halo_trim != 0, true), nr = face_runs(halo_rows(c), c->halo_trim != 0, false);
            if (k + 1 < n) {
                for (int q = 0; q < sr.n; ++q) NCCLCHK(ncclSend(b + f.top_rows + sr.off[q] * sub, sr.cnt[q] * sub, dt, k + 1, c->comm, s));
                for (int q = 0; q < nr.n; ++q) NCCLCHK(ncclRecv(b + f.ghost_n + nr.off[q] * sub, nr.cnt[q] * sub, dt, k + 1, c->comm, s));
            }
            if (k > 0) {
                for (int q = 0; q < nr.n; ++q) NCCLCHK(ncclSend(b + f.bot_rows + nr.off[q] * sub, nr.cnt[q] * sub, dt, k - 1, c->comm, s));
                for (int q = 0; q < sr.n; ++q) NCCLCHK(ncclRecv(b + f.ghost_s + sr.off[q] * sub, sr.cnt[q] * sub, dt, k - 1, c->comm, s));
            }
        }
        NCCLCHK(ncclGroupEnd());
        return LBM_OK;
    }
    for (int k = 0; k < n; ++k) {
        int rc = pull_halos<T>(cs, n, k, dst);
        if (rc) return rc;
    }
    return LBM_OK;
}

// ---- one launch, in phases --------------------------------------------------------------------------------
// A launch advances `depth` iterations (1, or 2/3 fused). Strips (a context with internal faces) issue launches in
// pairs between halo exchanges: KIND_EXTENDED (first of a pair: the strip's rows plus ext_rows ghost rows per internal
// face, no exchange afterwards) and KIND_EXCHANGE (a normal launch followed by the exchange of GR rows); without
// deep halos every launch is KIND_EXCHANGE. KIND_LOCAL: no neighbour to talk to.
//
// KIND_EXCHANGE with overlap. The E rows next to each neighbour ("edge bands": E = GR for one iteration,
// one band of the fused kernel otherwise) contain the GR rows that travel. They are updated by ONE launch on the side
// stream, followed there by the exchange; the remaining interior rows are updated concurrently on the main stream:
//   side stream : wait(ev_main: everything queued on the main stream so far) -> edge bands -> record(ev_edge)
//                 -> exchange -> record(ev_comm)
//   main stream : record(ev_main) ... wait(ev_edge of the PREVIOUS group) -> interior rows
// Hazards: edge(n) and interior(n) both read rows the other kind wrote in group n-1 (ev_main / ev_edge); edge(n) reads
// the ghost rows recv(n-1) wrote and recv(n) overwrites ghost rows edge(n-1) read, send(n) reads what edge(n) wrote,
// edge(n+1) overwrites rows send(n-1) read (all ordered by the side stream itself); interior(n) overwrites rows of
// the buffer edge(n-1) read (ev_edge). Interior rows read no ghost row (E >= GR) and write no edge row. Consumers on
// the main stream (forces, snapshots) first wait for ev_comm (join_comm). In a group with the peer transport a strip's
// edge rows are additionally read by its NEIGHBOURS' pulls: before they are overwritten the launching stream waits for
// the neighbours' ev_comm (their last pull).
// Without overlap the whole launch and the exchange run on the main stream (ev_edge / ev_comm are recorded all the
// same: the group transports order themselves by them).
enum { KIND_LOCAL = 0, KIND_EXTENDED = 1, KIND_EXCHANGE = 2 };
struct Launch { int depth, kind, src, dst, t; };

inline int join_comm(lbm_ctx* c);
inline bool face_south(const lbm_ctx* c) { return c->p.y_start > 0 || c->loopback; }
inline bool face_north(const lbm_ctx* c) { return c->p.y_start + c->nyl < c->p.ny || c->loopback; }

template <typename T>
void launch_depth(const lbm_ctx* c, const KArgs<T>& a, int depth, hipStream_t s) {
    if (c->rec) {
        ChoreoOp o; o.kind = ChoreoOp::KERNEL; o.strip = c->group_k; o.stream = stream_id(c, s); o.buf = a.dst == c->buf[0] ? 0 : 1;
        o.t = a.t; o.depth = depth;
        o.w0[0] = a.y_lo; o.w1[0] = a.y_lo + a.y_cnt; o.w0[1] = a.y_lo2; o.w1[1] = a.y_lo2 + a.y_cnt2;
        c->rec->ops.push_back(o);
        return;
    }
    if (depth > 1) launch_fused_rows<T>(c, a, depth, s);
    else launch_rows<T, MODE_STEP>(c, a, s);
}

// Decide the next launch of a context that still has `remaining` iterations to go in this call. Fusion: d iterations are
// fused only when the plan allows it, when none of the iterations t+1 .. t+d-1 is a force-output iteration (their
// post-collision states never exist in memory) and when at least one more iteration follows inside this call, so that
// the last launch of every lbm_step call is a single iteration and buf[cur^1] holds the previous iteration's
// populations (macro snapshot / f_current accessors) — unless "trailing_pair" lifts that rule. The last launch of a call
// is never the first of a pair, so every call ends with valid ghost rows. Every rank derives the same sequence from
// (steps_done, remaining, output_frequency). Without a device transport (host-staged halos: the caller exchanges after
// every call) a call may therefore contain at most two launches.
inline int plan_launch(lbm_ctx* c, int remaining, int of, bool transport, bool strip_logic, Launch* L) {
    const int t = c->steps_done;
    int depth = 1;
    const bool any_face = strip_logic && (face_south(c) || face_north(c));
    bool deep_plan = false;      // a deep plan exchanges after every launch (no extended first launch of a pair)
    c->deep_now = false;
    if (c->fuse > 1) {
        const int room = remaining - (c->trailing_pair ? 0 : 1);       // iterations a fused launch may take now
        int dmax = std::min(c->fuse, any_face ? 3 : 4);  // (four: k_step4_tile, no faces)
        // a deep plan: D iterations while D fit, then the four-/three-/two-iteration tile kernels for what is left
        // (a strip with faces: its ghost rows go GR deep and are refreshed after every launch, so a deep launch of up to GR
        // iterations works there too — the register kernel with five / six iterations and the 64x16 LDS shape with six, not the
        // seven / eight ones)
        // Whether the faces are EXCHANGED in this call (strip_logic) or not (the plan probe), a context whose rows end at an
        // internal face has GR rows beyond them and no more: the seven- / eight-iteration shapes would read past the frame.
        const bool phys_face = face_south(c) || face_north(c);
        // (with a device transport the exchange refreshes as many rows as the plan's launches are deep, up to the frame's
        // twelve: the seven- / eight-iteration LDS shapes run on strips too; host-staged halos carry six)
        const int hmax = (device_transport(c) || !strip_logic) ? GR : HR1;
        const int deep = (c->deep && (!phys_face || deep_depth(c->deep) <= hmax)) ? deep_depth(c->deep) : 0;
        if (deep) {
            // `seg` iterations may be fused from here: up to the next force-output iteration (its post-collision state
            // must exist in memory) and the end of the call. A long segment takes the plan's depth; near its end the
            // cheapest split into launches of the available depths is taken instead (20 = 7 + 7 + 6 in registers, 6 + 6 + 4 + 4
            // on an LDS shape, rather than 6 + 6 + 6 + 2: the two- and one-iteration kernels run at half and a third of the
            // fused rate).
            const int seg = of > 0 ? std::min(room, of - t % of) : room;
            // depths the plan's kernel family offers: the register kernel five and six iterations anywhere, seven on a
            // context without strip faces (a strip's ghost rows go six deep); the LDS shapes their own depth only
            auto in_family = [&](int d) {
                if (d == deep) return true;
                if (!deep_is_col(c->deep)) return false;
                if (deep_is_tall(c->deep)) return d == 6 || (d == 8 && !phys_face);       // (a strip exchanges seven rows on this plan)
                return d == 5 || d == 6 || (d == 7 && !phys_face);
            };
            int fam_min = deep;
            for (int d = 2; d < deep; ++d) if (in_family(d)) { fam_min = d; break; }
            dmax = std::min(any_face ? 3 : 4, fam_min - 1);
            if (seg >= 4 * deep) depth = deep;
            else if (seg >= 2) {
                const double* per_it = c->depth_rel;                             // depth 1..4 relative to the deep kernel
                constexpr double LAUNCH = 0.25;   // what one more launch costs, in iterations of the deep kernel (kernel boundary + a partly filled last round)
                double best[64];
                int first[64];
                best[0] = 0.0; first[0] = 0;
                for (int r = 1; r <= seg; ++r) {
                    best[r] = 1e30; first[r] = 1;
                    for (int d = 1; d <= std::min(r, 8); ++d) {
                        const bool fam = in_family(d);
                        if (!fam && d > dmax) continue;
                        const double cst = best[r - d] + (fam ? 1.0 : per_it[d - 1]) * d + LAUNCH;
                        if (cst < best[r] - 1e-12 || (d == deep && cst < best[r] + 1e-12)) { best[r] = cst; first[r] = d; }
                    }
                }
                depth = first[seg];
            }
            c->deep_now = depth > 1 && in_family(depth);
            dmax = depth;       // (decided: the generic rule below only confirms it)
            deep_plan = true;
        }
        // A three-iteration plan leaves a one- or two-iteration launch at the end of a call whose length is not a
        // multiple of three, which runs at half the rate. Where the four-iteration kernel is usable, one (remainder 1)
        // or two (remainder 2) four-iteration launches absorb it: 20 = 4 + 4 + 3 + 3 + 3 + 3.
        if (c->fuse == 3 && dmax == 3 && !any_face && of <= 0 && (room % 3 == 1 ? room >= 4 : (room % 3 == 2 && room >= 8)))
            dmax = 4;
        for (int d = dmax; d >= 2 && depth == 1; --d) {
            if (room < d) continue;
            bool ok = true;
            for (int j = 1; j < d; ++j) ok = ok && !(of > 0 && (t + j) % of == 0);
            if (ok) depth = d;
        }
    }
    const bool faces = strip_logic && (face_south(c) || face_north(c));
    const bool last = remaining - depth <= 0;
    L->depth = depth; L->src = c->cur; L->dst = c->cur ^ 1; L->t = t;
    // (an extended launch recomputes ext_rows ghost rows and leaves as many valid ones: launches of up to ext_rows iterations only;
    // a deep plan with a device transport exchanges after every launch instead)
    // (deep plans pair up with "deep_halo" 2 only, and only with a device transport: host-staged halos carry six rows)
    const bool pairs = c->deep_halo && (deep_plan && transport ? deep_pairs(c) : depth <= HR1 / 2);
    if (faces && pairs && !last && !c->mid_pair && depth <= ext_rows(c)) L->kind = KIND_EXTENDED;
    else {
        if (faces && !transport && !last)
            return fail(LBM_ERR_ARG, "a strip with host-staged halos can take at most two launches per lbm_step call "
                                     "(exchange the edge rows, then call again)");
        L->kind = transport ? KIND_EXCHANGE : KIND_LOCAL;
    }
    return LBM_OK;
}

// Everything of a launch that precedes its exchange.
template <typename T>
int issue_before(lbm_ctx* c, const Launch& L) {
    KArgs<T> a = make_kargs<T>(c, L.src, L.dst, L.t);
    const int rev = (c->alternate && (c->launches_total & 1)) ? 1 : 0;
    if (L.kind == KIND_LOCAL) {
        a.reverse = rev;
        launch_depth<T>(c, a, L.depth, c->stream);
        LAUNCHED(c);
        return LBM_OK;
    }
    // one tile band — and at least the rows that travel: the exchange reads them behind ev_edge, i.e. behind the EDGE launch only.
    // (Round 4: with twelve- and eight-row exchanges a remainder launch of one iteration, whose band used to be six rows, left the
    // rows beyond them to the interior launch — a race with the neighbour's pull that showed as a 1-in-15 mismatch of a threaded
    // group across force outputs.)
    const int band = c->deep_now ? deep_rows(c, c->deep, L.depth) : L.depth > 1 ? c->pair_ty : HR1;
    const int E = c->debug_old_edge_band ? band : std::max(band, halo_rows(c));      // (TEST ONLY: the rule as it was before the fix)
    if (L.kind == KIND_EXTENDED) {   // all rows of the strip PLUS the ext_rows ghost rows next to each internal face
        const int es = face_south(c) ? ext_rows(c) : 0, en = face_north(c) ? ext_rows(c) : 0;
        int e0 = face_south(c) ? E : 0, e1 = face_north(c) ? E : 0;
        if (c->overlap == 2 && c->comm_issued && e0 + e1 < c->nyl) {
            // Schedule 2: the exchange that follows the previous launch is still in flight on the side stream. The rows
            // that do not depend on it start now on the main stream; the bands next to the faces (and the extension)
            // follow the exchange on the side stream. The next launch waits for ev_edge.
            a.y_lo = e0; a.y_cnt = c->nyl - e0 - e1; a.reverse = rev;
            launch_depth<T>(c, a, L.depth, c->stream);
            LAUNCHED(c);
            KArgs<T> b = make_kargs<T>(c, L.src, L.dst, L.t);
            b.y_lo = -es; b.y_cnt = e0 + es; b.y_lo2 = c->nyl - e1; b.y_cnt2 = e1 + en;
            if (b.y_cnt == 0) { b.y_lo = b.y_lo2; b.y_cnt = b.y_cnt2; b.y_cnt2 = 0; }
            launch_depth<T>(c, b, L.depth, c->comm_stream);
            LAUNCHED(c);
            QCHK(q_record(c, c->ev_edge, c->comm_stream));
            c->ext_split_pending = true;
            return LBM_OK;
        }
        int rc = join_comm(c);       // the last exchange (and the edge bands before it) live on the side stream
        if (rc) return rc;
        a.y_lo = -es;
        a.y_cnt = c->nyl + es + en;
        a.reverse = rev;
        launch_depth<T>(c, a, L.depth, c->stream);
        LAUNCHED(c);
        return LBM_OK;
    }
    hipStream_t es = exchange_stream(c);
    auto wait_for_neighbour_pulls = [&](hipStream_t s) -> int {   // group / peer: my edge rows of buf[dst] may still be being read
        if (c->debug_skip_pull_wait) return LBM_OK;               // (TEST ONLY: the dry-run checker must name the race this leaves)
        for (lbm_ctx* nb : {c->nb_south, c->nb_north})
            if (nb && c->group_transport == 0 && nb->comm_issued) QCHK(q_wait(c, s, nb, nb->ev_comm));
        return LBM_OK;
    };
    if (c->overlap != 1) {
        // 0: launch and exchange on the main stream. 2: the launch on the main stream, the exchange on the side stream
        // behind it (ev_main) — it is the NEXT launch's interior rows that overlap with it.
        int rc = join_comm(c);       // (2) the edge part of a split extended launch / the exchange of a shallow-halo run
        if (rc) return rc;
        rc = wait_for_neighbour_pulls(c->stream);
        if (rc) return rc;
        a.reverse = rev;
        launch_depth<T>(c, a, L.depth, c->stream);
        LAUNCHED(c);
        QCHK(q_record(c, c->ev_edge, c->stream));
        if (c->overlap == 2) {
            QCHK(q_record(c, c->ev_main, c->stream));
            QCHK(q_wait(c, c->comm_stream, c, c->ev_main));
        }
        return LBM_OK;
    }
    const bool has_s = face_south(c), has_n = face_north(c);
    // (edge bands on the 64x16 LDS tile of one-cell threads with the interior in registers — the shortest edge launch — were
    // measured: one rank of eight / four / two 8.70 / 11.48 / 17.61 us per iteration against 8.63 / 11.38 / 17.39 for the
    // register kernel throughout: the interior blocks share the CUs with the edge blocks either way. Not kept.)
    int e0 = has_s ? E : 0, e1 = has_n ? E : 0;
    if (e0 + e1 >= c->nyl) { e0 = c->nyl; e1 = 0; }          // short strip: everything is edge
    QCHK(q_record(c, c->ev_main, c->stream));
    QCHK(q_wait(c, c->comm_stream, c, c->ev_main));
    if (c->comm_issued) QCHK(q_wait(c, c->stream, c, c->ev_edge));   // ev_edge still is the previous group's
    int rc = wait_for_neighbour_pulls(es);
    if (rc) return rc;
    a.reverse = 0;
    a.y_lo = 0; a.y_cnt = e0; a.y_lo2 = c->nyl - e1; a.y_cnt2 = e1;
    if (e0 == 0) { a.y_lo = a.y_lo2; a.y_cnt = e1; a.y_cnt2 = 0; }
    launch_depth<T>(c, a, L.depth, c->comm_stream);
    LAUNCHED(c);
    QCHK(q_record(c, c->ev_edge, c->comm_stream));
    c->edge_rows[0] = e0; c->edge_rows[1] = e1;
    return LBM_OK;
}

// Everything of a launch that follows its exchange, and the bookkeeping.
template <typename T>
int issue_after(lbm_ctx* c, const Launch& L) {
    if (L.kind == KIND_EXCHANGE) {
        QCHK(q_record(c, c->ev_comm, exchange_stream(c)));
        c->comm_issued = true;
        if (c->overlap == 1) {
            const int e0 = c->edge_rows[0], e1 = c->edge_rows[1];
            if (c->nyl - e0 - e1 > 0) {
                KArgs<T> a = make_kargs<T>(c, L.src, L.dst, L.t);
                a.y_lo = e0; a.y_cnt = c->nyl - e0 - e1;
                a.reverse = (c->alternate && (c->launches_total & 1)) ? 1 : 0;
                launch_depth<T>(c, a, L.depth, c->stream);
                LAUNCHED(c);
            }
        }
    }
    c->mid_pair = (L.kind == KIND_EXTENDED);
    c->cur = L.dst;
    c->steps_done = L.t + L.depth;
    c->launches_total++;
    c->last_was_pair = L.depth > 1;
    c->restored = false;
    return LBM_OK;
}

// Advance ONE context by up to `remaining` iterations with one launch; returns the iterations taken (1..3) or <0.
template <typename T>
int advance(lbm_ctx* c, int remaining, int of, bool transport, bool strip_logic = true) {
    Launch L;
    int rc = plan_launch(c, remaining, of, transport, strip_logic, &L);
    if (rc) return rc;
    rc = issue_before<T>(c, L);
    if (rc) return rc;
    if (L.kind == KIND_EXCHANGE) {
        rc = exchange_rccl<T>(c, L.dst, exchange_stream(c));
        if (rc) return rc;
    }
    rc = issue_after<T>(c, L);
    if (rc) return rc;
    return L.depth;
}

// Make everything issued so far (both streams) visible to work queued on the compute stream afterwards.
inline int join_comm(lbm_ctx* c) {
    if (c->ext_split_pending) {   // overlap 2: the edge bands of the last extended launch (queued behind the exchange)
        QCHK(q_wait(c, c->stream, c, c->ev_edge));
        c->ext_split_pending = false;
    }
    if (c->comm_issued) QCHK(q_wait(c, c->stream, c, c->ev_comm));
    return LBM_OK;
}

template <typename T>
int init_state(lbm_ctx* c) {
    InitArgs<T> ia;
    ia.a = static_cast<T*>(c->buf[0]);
    ia.b = static_cast<T*>(c->buf[1]);
    ia.plane = (long)c->plane; ia.pitch = c->pitch; ia.xoff = c->xoff;
    ia.nx = c->nx; ia.ny_loc = c->nyl; ia.ny_glob = c->p.ny; ia.y_start = c->p.y_start;
    ia.cyl_x = c->cyl_x; ia.cyl_y = c->cyl_y; ia.cyl_r2 = (double)(c->cyl_r * c->cyl_r);
    for (int i = 0; i < Q; ++i) ia.feq_in[i] = (T)c->feq_in[i];
    ia.solid_count = c->d_solid_count;
    HIPCHK(hipMemsetAsync(c->d_solid_count, 0, sizeof(int), c->stream));
    dim3 grid((c->nx + 2 + 255) / 256, c->nyl + 2 * GR), block(256);
    hipLaunchKernelGGL((k_init<T>), grid, block, 0, c->stream, ia);
    HIPCHK(hipGetLastError());
    // collision_step of iteration 0: initial state (buf 0) -> P_0 (buf 1)
    int rc = launch_step<T>(c, 0, 1, 0, MODE_COLLIDE_ONLY, c->stream);
    if (rc) return rc;
    c->cur = 1;
    c->steps_done = 0;
    return LBM_OK;
}

inline void free_buffers(lbm_ctx* c) {
    for (int k = 0; k < 2; ++k)
        if (c->buf[k]) { (void)hipFree(c->buf[k]); c->buf[k] = nullptr; }
}
inline int alloc_buffers(lbm_ctx* c) {
    free_buffers(c);
    HIPCHK(hipMalloc(&c->buf[0], buffer_bytes(c)));
    HIPCHK(hipMalloc(&c->buf[1], buffer_bytes(c)));
    return LBM_OK;
}

