// csrc/lbm_choreo.inc.hpp — dry run of the strip choreography and its checker (lbm_debug_choreography; no device needed)
// (part of the one host translation unit lbm_hip.hip, which includes it after lbm_steps.inc.hpp)
//
// Round 4 introduced — and fixed — a race of the launch choreography: with twelve- and eight-row exchanges a one-iteration remainder
// launch kept a six-row edge band, so rows 7-12 of the rows that travel were written by the INTERIOR launch, which no event orders
// with the neighbour's pull. It showed as 3 mismatching runs in 40. The choreography is pure host arithmetic plus a fixed pattern of
// event records and waits, so the class can be checked exhaustively without a GPU: the functions that issue a launch group run here
// on contexts without a device (lbm_ctx::rec: every runtime call becomes an entry of a list), and the list is replayed below:
//   ordering  — vector clocks per stream; an event record snapshots its stream's clock, a wait joins it. Every row of every buffer
//               of every strip remembers its last writer and the readers since; an access that conflicts with one of them (same row,
//               at least one write) without a happens-before path is a RACE. This is what the reference gets for free from
//               MPI_Waitall (LBMGrid.h:278-280) and what ev_main / ev_edge / ev_comm must provide here.
//   freshness — every row carries the iteration of the data it holds; a launch of d iterations from t must find iteration t in every
//               row within d rows of the rows it writes (its dependency cone), a force launch iteration t in the strip's rows. A ghost
//               frame that is exchanged too rarely or not deep enough for the launches between two exchanges is STALE.
// The host order of a group is the order of its three phases (every strip's issue_before, the exchange, every strip's issue_after),
// which is what the rendezvous of the threaded path enforce; within a phase the strips touch no common event.
struct ChoreoViolation { int kind; int op_a, op_b; int strip, buf, row; int want, have; };   // kind 0 race, 1 stale

struct ChoreoChecker {
    struct RowState { int w_stream = -1; unsigned w_seq = 0; int w_op = -1; int level = INT_MIN; };
    static constexpr int CONST_LEVEL = INT_MAX;       // a physical wall's ghost rows: the initial equilibrium, never written (N2)
    int nstrips = 0, nstreams = 0;
    std::vector<int> nyl;
    std::vector<std::vector<RowState>> rows[2];       // [buf][strip][gy]
    std::vector<std::vector<unsigned>> readers[2];    // [buf][strip][gy * nstreams + stream] = seq of the stream's last read (0: none)
    std::vector<std::vector<int>> reader_op[2];
    std::vector<std::vector<unsigned>> vc;            // per stream
    std::vector<std::vector<unsigned>> evc;           // per (strip, event): the clock its last record snapshot; empty = never recorded
    std::vector<ChoreoViolation> bad;

    void init(lbm_ctx** cs, int n) {
        nstrips = n; nstreams = 2 * n;
        nyl.resize((size_t)n);
        vc.assign((size_t)nstreams, std::vector<unsigned>((size_t)nstreams, 0u));
        evc.assign((size_t)3 * n, std::vector<unsigned>());
        for (int b = 0; b < 2; ++b) { rows[b].resize((size_t)n); readers[b].resize((size_t)n); reader_op[b].resize((size_t)n); }
        for (int k = 0; k < n; ++k) {
            const lbm_ctx* c = cs[k];
            nyl[(size_t)k] = c->nyl;
            const int tot = c->nyl + 2 * GR;
            for (int b = 0; b < 2; ++b) {
                rows[b][(size_t)k].assign((size_t)tot, RowState());
                readers[b][(size_t)k].assign((size_t)tot * nstreams, 0u);
                reader_op[b][(size_t)k].assign((size_t)tot * nstreams, -1);
                // physical walls: constant ghost rows in both buffers
                for (int gy = 0; gy < tot; ++gy) {
                    const int y = gy - GR;
                    const bool wall = (y < 0 && !face_south(c)) || (y >= c->nyl && !face_north(c));
                    if (wall) rows[b][(size_t)k][(size_t)gy].level = CONST_LEVEL;
                }
            }
            // the state lbm_initialise (+ the first exchange) leaves: P_0 in buf[cur], the ghost frame filled halo_rows deep
            const int hr = halo_rows(c);
            for (int y = -hr; y < c->nyl + hr; ++y) {
                RowState& r = rows[c->cur][(size_t)k][(size_t)(y + GR)];
                if (r.level != CONST_LEVEL) r.level = 0;
            }
        }
    }
    bool hb(int w_stream, unsigned w_seq, int s) const { return w_stream < 0 || vc[(size_t)s][(size_t)w_stream] >= w_seq; }
    void race(int op_a, int op_b, int strip, int buf, int row) { if (bad.size() < 64) bad.push_back({0, op_a, op_b, strip, buf, row, 0, 0}); else bad.back().want++; }
    void read_row(int op, int s, int strip, int buf, int y) {
        if (y < -GR || y >= nyl[(size_t)strip] + GR) return;
        const size_t gy = (size_t)(y + GR);
        const RowState& r = rows[buf][(size_t)strip][gy];
        if (!hb(r.w_stream, r.w_seq, s)) race(r.w_op, op, strip, buf, y);
        readers[buf][(size_t)strip][gy * nstreams + s] = vc[(size_t)s][(size_t)s];
        reader_op[buf][(size_t)strip][gy * nstreams + s] = op;
    }
    void write_row(int op, int s, int strip, int buf, int y, int level) {
        if (y < -GR || y >= nyl[(size_t)strip] + GR) return;
        const size_t gy = (size_t)(y + GR);
        RowState& r = rows[buf][(size_t)strip][gy];
        if (!hb(r.w_stream, r.w_seq, s)) race(r.w_op, op, strip, buf, y);
        for (int q = 0; q < nstreams; ++q) {
            unsigned& seq = readers[buf][(size_t)strip][gy * nstreams + q];
            if (seq && !hb(q, seq, s)) race(reader_op[buf][(size_t)strip][gy * nstreams + q], op, strip, buf, y);
            seq = 0;
        }
        r.w_stream = s; r.w_seq = vc[(size_t)s][(size_t)s]; r.w_op = op;
        if (r.level != CONST_LEVEL) r.level = level;
    }
    int level_of(int strip, int buf, int y) const {
        if (y < -GR || y >= nyl[(size_t)strip] + GR) return INT_MIN;
        return rows[buf][(size_t)strip][(size_t)(y + GR)].level;
    }
    void need_level(int op, int strip, int buf, int y, int want) {
        if (y < -GR || y >= nyl[(size_t)strip] + GR) return;
        const int have = level_of(strip, buf, y);
        if (have != want && have != CONST_LEVEL && bad.size() < 64) bad.push_back({1, op, -1, strip, buf, y, want, have});
    }
    void run(const std::vector<ChoreoOp>& ops) {
        for (int idx = 0; idx < (int)ops.size(); ++idx) {
            const ChoreoOp& o = ops[(size_t)idx];
            const int s = 2 * o.strip + o.stream;
            std::vector<unsigned>& me = vc[(size_t)s];
            me[(size_t)s]++;
            switch (o.kind) {
                case ChoreoOp::RECORD: evc[(size_t)(3 * o.ev_strip + o.ev)] = me; break;
                case ChoreoOp::WAIT: {
                    const std::vector<unsigned>& e = evc[(size_t)(3 * o.ev_strip + o.ev)];
                    for (size_t q = 0; q < e.size(); ++q) me[q] = std::max(me[q], e[q]);
                    break;
                }
                case ChoreoOp::KERNEL: {
                    const int src = o.buf ^ 1;
                    for (int k = 0; k < 2; ++k)
                        for (int y = o.w0[k] - o.depth; y < o.w1[k] + o.depth && o.w1[k] > o.w0[k]; ++y) { read_row(idx, s, o.strip, src, y); need_level(idx, o.strip, src, y, o.t); }
                    for (int k = 0; k < 2; ++k)
                        for (int y = o.w0[k]; y < o.w1[k]; ++y) write_row(idx, s, o.strip, o.buf, y, o.t + o.depth);
                    break;
                }
                case ChoreoOp::COPY:
                    for (int j = 0; j < o.r1 - o.r0; ++j) {
                        read_row(idx, s, o.r_strip, o.buf, o.r0 + j);
                        write_row(idx, s, o.strip, o.buf, o.w0[0] + j, level_of(o.r_strip, o.buf, o.r0 + j));
                    }
                    break;
                case ChoreoOp::SEND:
                    for (int y = o.r0; y < o.r1; ++y) read_row(idx, s, o.strip, o.buf, y);
                    break;
                case ChoreoOp::RECV:
                    for (int j = 0; j < o.w1[0] - o.w0[0]; ++j) {
                        // the data: the sender's rows (a member of this group), or — another process — the edge rows of a strip that runs the
                        // same schedule in lockstep: they hold the iteration of the rows I send across the same face in this exchange
                        int level;
                        const int hr = o.w1[0] - o.w0[0];
                        if (o.r_strip >= 0) level = level_of(o.r_strip, o.buf, o.r0 + j);
                        else level = o.w0[0] < 0 ? level_of(o.strip, o.buf, j) : level_of(o.strip, o.buf, nyl[(size_t)o.strip] - hr + j);
                        write_row(idx, s, o.strip, o.buf, o.w0[0] + j, level);
                    }
                    break;
                case ChoreoOp::FORCES:
                    for (int y = o.r0; y < o.r1; ++y) { read_row(idx, s, o.strip, o.buf, y); need_level(idx, o.strip, o.buf, y, o.t); }
                    break;
                default: break;
            }
        }
    }
};

inline std::string choreo_op_text(const ChoreoOp& o, int idx) {
    static const char* kinds[] = {"kernel", "record", "wait", "copy", "send", "recv", "forces"};
    static const char* evs[] = {"ev_main", "ev_edge", "ev_comm"};
    char b[256];
    int n = snprintf(b, sizeof(b), "#%d strip %d %s stream: %s", idx, o.strip, o.stream ? "side" : "main", kinds[o.kind]);
    if (o.kind == ChoreoOp::KERNEL)
        n += snprintf(b + n, sizeof(b) - n, " t=%d depth=%d writes buf %d rows [%d,%d)%s", o.t, o.depth, o.buf, o.w0[0], o.w1[0],
                      o.w1[1] > o.w0[1] ? (" + [" + std::to_string(o.w0[1]) + "," + std::to_string(o.w1[1]) + ")").c_str() : "");
    else if (o.kind == ChoreoOp::RECORD || o.kind == ChoreoOp::WAIT) n += snprintf(b + n, sizeof(b) - n, " %s of strip %d", evs[o.ev], o.ev_strip);
    else if (o.kind == ChoreoOp::COPY) n += snprintf(b + n, sizeof(b) - n, " buf %d: rows [%d,%d) of strip %d -> rows [%d,%d)", o.buf, o.r0, o.r1, o.r_strip, o.w0[0], o.w1[0]);
    else if (o.kind == ChoreoOp::SEND) n += snprintf(b + n, sizeof(b) - n, " buf %d rows [%d,%d)", o.buf, o.r0, o.r1);
    else if (o.kind == ChoreoOp::RECV) n += snprintf(b + n, sizeof(b) - n, " buf %d rows [%d,%d)", o.buf, o.w0[0], o.w1[0]);
    else if (o.kind == ChoreoOp::FORCES) n += snprintf(b + n, sizeof(b) - n, " t=%d reads buf %d rows [%d,%d)", o.t, o.buf, o.r0, o.r1);
    return b;
}

// A context without a device: the fields lbm_create / lbm_initialise would fill, streams / events / buffers as distinct non-null tokens.
inline lbm_ctx* choreo_fake_ctx(int nx, int ny, int y_start, int rows, int precision, int k) {
    lbm_ctx* c = new lbm_ctx();
    c->p.nx = nx; c->p.ny = ny; c->p.y_start = y_start; c->p.local_ny = rows; c->p.precision = precision;
    c->p.tau = 0.6; c->p.inlet_velocity = 0.05; c->p.cylinder_x = 0.2; c->p.cylinder_y = 0.5; c->p.cylinder_radius = 0.05;
    c->nx = nx; c->nyl = rows;
    c->esize = precision == LBM_PRECISION_F32 ? 4 : 8;
    const int per128 = (int)(128 / c->esize);
    c->xoff = per128;
    c->pitch0 = round_up(c->xoff + nx + 1, per128);
    configure_layout(c, 1);
    c->cyl_x = (int)(0.2 * nx); c->cyl_y = (int)(0.5 * ny); c->cyl_r = (int)(0.05 * ny);
    const uintptr_t base = 0x10000u * (uintptr_t)(k + 1);
    c->stream = (hipStream_t)(base + 0x10); c->comm_stream = (hipStream_t)(base + 0x20);
    c->ev_main = (hipEvent_t)(base + 0x30); c->ev_edge = (hipEvent_t)(base + 0x40); c->ev_comm = (hipEvent_t)(base + 0x50);
    c->buf[0] = (void*)(base + 0x1000); c->buf[1] = (void*)(base + 0x2000);
    c->group_k = k;
    c->tune = 0;
    c->log_cap = 1 << 20;
    return c;
}
