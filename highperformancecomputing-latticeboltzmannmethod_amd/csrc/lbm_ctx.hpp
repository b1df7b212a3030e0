// csrc/lbm_ctx.hpp — error reporting, the host threads of a group of strips (GroupPool) and struct lbm_ctx: everything a context owns
// (part of the one host translation unit lbm_hip.hip, which includes it in this place; round 4 split a 2 100-line file by concern)
#pragma once
namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(LBM_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                                          hipGetErrorString(e_));                                 \
    } while (0)
#define NCCLCHK(expr)                                                                               \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) return fail(LBM_ERR_COMM, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                                           ncclGetErrorString(r_));                                 \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

// A stall must name itself (a GPU tool sat silent for seven minutes and was killed with nothing on record): with
// LBM_TRACE=<file> in the environment the library appends one line per coarse event (create / comm_init / initialise / step /
// group_step / destroy, begin and end) with a timestamp, pid and thread, flushed at once. Off: one cached getenv.
inline void lbm_trace(const char* what, const char* fmt = nullptr, ...) {
    static FILE* fp = [] { const char* f = getenv("LBM_TRACE"); return (f && *f) ? fopen(f, "a") : (FILE*)nullptr; }();
    if (!fp) return;
    static std::mutex mu;
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    char detail[256] = "";
    if (fmt) { va_list ap; va_start(ap, fmt); vsnprintf(detail, sizeof(detail), fmt, ap); va_end(ap); }
    std::lock_guard<std::mutex> lk(mu);
    fprintf(fp, "%.6f pid %d tid %lx %s %s\n", t, (int)getpid(), (unsigned long)std::hash<std::thread::id>{}(std::this_thread::get_id()) & 0xffffff, what, detail);
    fflush(fp);
}
// the bound of every host-side wait of the library (rendezvous of a group's threads, completion of a group job, lbm_sync,
// the drains of lbm_destroy): LBM_WAIT_TIMEOUT_MS, default five minutes — below the seven minutes of silence after which the
// GPU harness kills a command, far above anything a healthy run waits for (a full hardware queue drains in seconds)
inline long default_wait_timeout_ms() {
    static const long v = [] { const char* e = getenv("LBM_WAIT_TIMEOUT_MS"); const long x = e ? atol(e) : 0; return x > 0 ? x : 300000L; }();
    return v;
}

// The host threads of an in-process group of strips (lbm_group_link): one per strip beyond the first, created ONCE and parked
// on a condition variable between lbm_group_step calls (a call used to spawn and join n-1 std::threads: Solver::run issues one
// call per output chunk, ~1 ms of GPU work at N = 8). Strip 0 is driven by the calling thread. Inside a job the n threads move
// in lockstep through `arrive`; whether a phase aborts is decided ONCE per rendezvous, by the last thread to arrive, from
// the error state as it stood then — so every thread takes the same branch and nobody is left waiting at the next rendezvous
// (a thread that failed after a rendezvous used to make a slower one return early).
// EVERY WAIT IS BOUNDED (round 5; rounds 3-4 used std::barrier and an untimed condition variable): a rendezvous that is not
// complete after `timeout_ms`, or a job whose threads have not all returned by then, turns into LBM_ERR_TIMEOUT with the strips
// that are missing and the launch / rendezvous each was last seen at, and poisons the pool: later jobs are refused, and the
// contexts of a poisoned group whose threads never came back are leaked by lbm_destroy rather than freed under a thread that
// may still wake up (the state the threads share lives on the heap and outlives the pool; a job's own state lives in its closure).
struct GroupPool {
    struct State {
        const int n;
        std::mutex mu;
        std::condition_variable cv_job, cv_done, cv_bar;
        std::function<void(State&, int)> job;      // a COPY of the caller's closure (which owns its state through a shared_ptr)
        unsigned long gen = 0;
        int pending = 0;
        bool stop = false;
        long timeout_ms = default_wait_timeout_ms();
        // rendezvous: arrivals are counted lock-free and a waiter spins for a few microseconds before it blocks (a launch of an
        // N = 8 strip is ~20 us of GPU work and passes three rendezvous)
        std::atomic<int> arrived{0};
        std::atomic<unsigned long> bgen{0};
        std::atomic<bool> broken{false};
        std::vector<std::atomic<int>> at;          // per strip: 4 * launch + rendezvous it last arrived at (for the message)
        std::atomic<int> err{0};
        int phase_err = 0;                 // written by the last arrival of a rendezvous only: the same for every thread of a phase
        std::mutex emu;
        std::string msg;
        explicit State(int n_) : n(n_), at((size_t)n_) { for (auto& a : at) a.store(-1); }
        void report(int rc, const char* text) {
            if (rc == 0) return;
            std::lock_guard<std::mutex> lk(emu);
            if (err.load() == 0) { msg = text; err.store(rc); }
        }
        std::string missing(int mine) {    // who has not arrived where strip `mine` waits
            std::string s;
            for (int k = 0; k < n; ++k) {
                const int a = at[(size_t)k].load();
                if (a >= mine) continue;
                char b[96];
                if (a < 0) snprintf(b, sizeof(b), "%sstrip %d (not yet at any rendezvous of this call)", s.empty() ? "" : ", ", k);
                else snprintf(b, sizeof(b), "%sstrip %d (last seen at rendezvous %d of launch %d)", s.empty() ? "" : ", ", k, a % 4, a / 4);
                s += b;
            }
            return s.empty() ? std::string("nobody (a late arrival)") : s;
        }
        // rendezvous `phase` (1..3) of launch `launch` of this call; true: some strip had failed by the time the last one arrived,
        // or the rendezvous timed out — EVERY thread sees true and leaves
        bool arrive(int i, int launch, int phase) {
            const int mine = 4 * launch + phase;
            at[(size_t)i].store(mine);
            if (broken.load()) return true;
            const unsigned long g = bgen.load(std::memory_order_acquire);
            if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
                arrived.store(0, std::memory_order_relaxed);
                phase_err = err.load();
                { std::lock_guard<std::mutex> lk(mu); bgen.store(g + 1, std::memory_order_release); }
                cv_bar.notify_all();
                return phase_err != 0 || broken.load();
            }
            for (int k = 0; k < 4000; ++k) {
                if (bgen.load(std::memory_order_acquire) != g) return phase_err != 0 || broken.load();
                __builtin_ia32_pause();
            }
            std::unique_lock<std::mutex> lk(mu);
            if (!cv_bar.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return bgen.load(std::memory_order_acquire) != g || broken.load(); })) {
                broken.store(true);
                char b[512];
                snprintf(b, sizeof(b), "strip %d waited %ld ms at rendezvous %d of launch %d of this call for: %s; the group is unusable, destroy it",
                         i, timeout_ms, phase, launch, missing(mine).c_str());
                report(LBM_ERR_TIMEOUT, b);
                lbm_trace("TIMEOUT", "%s", b);
                cv_bar.notify_all();
                return true;
            }
            return phase_err != 0 || broken.load();
        }
    };
    std::shared_ptr<State> S;
    std::vector<std::thread> th;
    explicit GroupPool(int n_) : S(std::make_shared<State>(n_)) {
        for (int i = 1; i < n_; ++i) th.emplace_back([s = S, i] { loop(*s, i); });
    }
    // all threads parked? (a poisoned pool: waits up to `grace_ms` for the stragglers first)
    bool quiesce(long grace_ms) {
        std::unique_lock<std::mutex> lk(S->mu);
        return S->cv_done.wait_for(lk, std::chrono::milliseconds(grace_ms), [&] { return S->pending == 0; });
    }
    ~GroupPool() {
        const bool idle = quiesce(S->broken.load() ? std::max(5000L, S->timeout_ms) : 0x7fffffffL);
        { std::lock_guard<std::mutex> lk(S->mu); S->stop = true; S->job = nullptr; }
        S->cv_job.notify_all();
        for (auto& t : th) { if (idle) t.join(); else t.detach(); }      // (a thread that never came back keeps the shared state alive)
    }
    static void loop(State& s, int i) {
        unsigned long seen = 0;
        for (;;) {
            std::function<void(State&, int)> f;
            {
                std::unique_lock<std::mutex> lk(s.mu);
                s.cv_job.wait(lk, [&] { return s.stop || s.gen != seen; });      // (parked: the one wait without a bound, ended by ~GroupPool)
                if (s.stop) return;
                seen = s.gen; f = s.job;
            }
            f(s, i);
            s.at[(size_t)i].store(INT_MAX);
            { std::lock_guard<std::mutex> lk(s.mu); if (--s.pending == 0) s.cv_done.notify_all(); }
        }
    }
    // run f(S, 0) .. f(S, n-1), one strip per thread; returns when all are done — or LBM_ERR_TIMEOUT after timeout_ms
    int run(std::function<void(State&, int)> f, long timeout_ms) {
        State& s = *S;
        if (s.broken.load()) return fail(LBM_ERR_TIMEOUT, "this group timed out earlier (%s): destroy it", s.msg.c_str());
        s.err.store(0); s.phase_err = 0; s.msg.clear(); s.arrived.store(0);
        for (auto& a : s.at) a.store(-1);
        { std::lock_guard<std::mutex> lk(s.mu); s.timeout_ms = timeout_ms > 0 ? timeout_ms : default_wait_timeout_ms(); s.job = f; s.pending = s.n - 1; ++s.gen; }
        s.cv_job.notify_all();
        f(s, 0);
        s.at[0].store(INT_MAX);
        std::unique_lock<std::mutex> lk(s.mu);
        if (!s.cv_done.wait_for(lk, std::chrono::milliseconds(s.timeout_ms), [&] { return s.pending == 0; })) {
            s.broken.store(true);
            char b[512];
            snprintf(b, sizeof(b), "%d strip thread(s) have not come back %ld ms after strip 0 finished its part: %s; the group is unusable, destroy it",
                     s.pending, s.timeout_ms, s.missing(INT_MAX).c_str());
            s.report(LBM_ERR_TIMEOUT, b);
            lbm_trace("TIMEOUT", "%s", b);
            s.cv_bar.notify_all();
        } else s.job = nullptr;
        if (s.err.load() != LBM_OK) return fail(s.err.load(), "%s", s.msg.c_str());
        return LBM_OK;
    }
};

// Dry run of the strip choreography (lbm_debug_choreography, include/lbm_hip.h; no device needed). With `rec` set on a context the
// functions that issue a launch group — plan_launch, issue_before, the exchanges, issue_after, join_comm, the force launch — append
// what they WOULD queue instead of calling the runtime: kernels with the rows they write (and, through their depth, read), event
// records, cross-stream waits, copies, sends and receives, each with its stream. lbm_choreo.inc.hpp replays the list with vector
// clocks and reports every pair of conflicting accesses that no event orders, and every launch that reads a row of the wrong
// iteration: the class of round 4's edge-band race (a band shorter than the rows that travel), found then by a 1-in-15 flake.
struct ChoreoOp {
    enum { KERNEL = 0, RECORD = 1, WAIT = 2, COPY = 3, SEND = 4, RECV = 5, FORCES = 6 };
    int kind = 0;
    int strip = 0, stream = 0;      // the issuing strip, 0 main / 1 side stream
    int ev_strip = 0, ev = 0;       // RECORD / WAIT: the event's owner and 0 ev_main, 1 ev_edge, 2 ev_comm
    int buf = 0;                    // KERNEL: the buffer written (it reads buf ^ 1); COPY / SEND / RECV / FORCES: the buffer touched
    int t = 0, depth = 0;           // KERNEL: first iteration and iterations; FORCES: the iteration
    int w0[2] = {0, 0}, w1[2] = {0, 0};   // rows written, [w0, w1) in local rows (ghost rows: < 0 or >= nyl); KERNEL: two ranges
    int r0 = 0, r1 = 0, r_strip = -1;     // COPY / SEND / FORCES: rows read and whose (RECV: where the data comes from; -1: another process)
};
// ... and, for one rank of a multi-process run, the transcript of what exchange_rccl's posting loops WOULD hand to RCCL, in posting order:
// kind 0 send / 1 recv / 2 end of a group call; `off`, `cnt` in elements of the buffer. lbm_debug_p2p_matching holds the transcripts of
// neighbouring ranks against each other (RCCL pairs the k-th send to a peer with the peer's k-th receive from the sender).
struct ChoreoP2P { int kind, peer; long off, cnt; };
struct Choreo { std::vector<ChoreoOp> ops; std::vector<ChoreoP2P> p2p; };

struct lbm_ctx {
    lbm_params p{};
    int device = 0;
    hipStream_t stream = nullptr;       // compute stream (all kernels)
    hipStream_t comm_stream = nullptr;  // halo exchange (RCCL send/recv)
    hipEvent_t ev_edge = nullptr, ev_comm = nullptr, ev_main = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    int nx = 0, nyl = 0, xoff = 0;
    int pitch0 = 0;          // elements of one sub-row (ghost columns + 128-B padding included)
    int pitch = 0;           // ROW stride: elements between consecutive rows of one plane
    size_t plane = 0;        // PLANE stride: elements between the same cell of consecutive planes
    size_t total = 0;        // elements per population buffer
    int layout = 0;          // 0 planar, 1 row-interleaved (see lbm_kernels.hpp)
    size_t esize = 8;        // bytes per element
    void* buf[2] = {nullptr, nullptr};
    int cur = 0;             // buf[cur] = P_{steps_done}; buf[cur^1] = P_{steps_done-1} (or the initial state)
    void* scratch = nullptr; // f_current snapshot (lazy)
    double* d_macro = nullptr;   // rho | ux | uy (lazy), each nx*nyl
    unsigned long long* d_maxbits = nullptr;
    int* d_unstable = nullptr;
    int* d_solid_count = nullptr;
    void* d_feq = nullptr;          // the nine initial-equilibrium values in the element type (fused kernels)
    double* d_force_now = nullptr;  // 3 doubles
    double* d_force_log = nullptr;  // capacity x 3 doubles
    int log_cap = 0, log_count = 0;
    int steps_done = 0;
    bool initialised = false;
    double feq_in[Q];
    int cyl_x = 0, cyl_y = 0, cyl_r = 0;
    // options
    int alternate = 1;   // walk the rows bottom-up / top-down on alternate steps (Infinity Cache reuse)
    int use_nt = 0;      // non-temporal stores in the step kernel
    int use_ntl = 0;     // non-temporal level-1 loads in the register kernel (k_stepc_col)
    int fuse = 1;        // iterations fused per launch where the schedule allows: 1, 2 (k_step2_tile) or 3 (k_step3_tile)
    int pair_ty = 8;     // tile height of the fused kernels (8 or 12)
    int xcd = 0;         // fused kernels: remap blocks so that each XCD walks a contiguous run of tiles
    bool deep_now = false;   // the launch being issued is the plan's deep launch (set by plan_launch)
    int deep = 0;        // 1..3: k_stepd_tile shape (6/7/8 iterations per launch on an LDS-filling tile); 6/7: k_stepc_col (registers)
    int arith = 0;       // collision arithmetic: 0 strict IEEE op-by-op (bit-identical to the oracle), 1 contracted (FMA +
                         // one reciprocal, as the reference's -ffast-math -mfma build permits); see lbm_kernels.hpp Arith
    int num_cus = 256;   // compute units of the device (what counts as a small grid: one round of blocks)
    int loopback = 0;    // TEST ONLY: the strip is its own north and south neighbour (exercises the overlap choreography):
                         // 1 = device copies, 2 = RCCL send/recv to self on a one-rank communicator
    int deep_halo = 1;       // strips: 1 = one exchange of six rows per TWO launches of up to three iterations (the first launch of a pair is
                             // extended); deep plans exchange after every launch. 2 = deep plans too: twelve rows per two launches of up to six
    int halo_trim = 0;       // strips: 1 = an exchange carries only the sub-rows the receiver's launches read (9 hr - 9 of the 9 hr sub-rows of a
                             // face, in five runs: the outermost ghost row is read for its three inbound populations only, the next one
                             // for all but its three outbound ones); 0 = all nine populations of every row, one contiguous message
    bool trim_pinned = false;
    int trailing_pair = 0;   // allow an lbm_step call to END on a fused launch (host-staged strips)
    bool mid_pair = false;        // the last launch was the extended first launch of a pair (no exchange after it)
    bool last_was_pair = false;   // the last launch fused several iterations: buf[cur^1] is older than steps_done-1
    bool restored = false;   // state came from lbm_load_state: no previous-iteration buffer until the next step
    int tune = 1;        // lbm_initialise times the candidate plans on this device and keeps the fastest
    char plan_desc[512] = "";
    char plan_opts[128] = "";    // the plan as lbm_set_option pairs ("layout=1 nt=0 ..."): with tune=0 they reproduce it in another process
    double depth_rel[4] = {2.8, 1.6, 1.12, 1.08};   // cost per iteration of a 1- / 2- / 3- / 4-iteration launch relative to the plan's deep
                                                    // launch (plan_launch's tail split); measured by choose_plan on a single domain,
                                                    // these defaults — 4096x1024 fp64, round 2 — elsewhere (strips: every rank must split alike)
    bool depth_rel_measured = false;
    int timing = 0;
    int overlap = 1;
    bool overlap_pinned = false, deep_pinned = false;   // set through lbm_set_option: the strip tuner leaves them alone
    int skip_exchange = 0;   // DIAGNOSTIC: issue every launch but no halo traffic (times the compute side of a strip run; results invalid)
    char sched_desc[800] = "";
    int timed_launches = 0, timed_steps = 0;
    long launches_total = 0;
    // communicator
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    bool comm_issued = false;   // ev_comm has been recorded at least once
    double* d_red = nullptr;
    // in-process group of strips (lbm_group_link): neighbours, transport (0 peer copies, 1 RCCL), size
    lbm_ctx* nb_south = nullptr;
    lbm_ctx* nb_north = nullptr;
    int group_transport = 0, group_n = 1, group_k = 0;
    int group_threads = 1;   // a group is driven by one host thread per strip (0: the calling thread issues for every strip)
    long wait_timeout_ms = 0;    // bound of the host-side waits (0: LBM_WAIT_TIMEOUT_MS or five minutes); option "wait_timeout_ms"
    // TEST ONLY (options "debug_fault_launch" / "_point" / "_stall_ms"): in launch `launch` of the next lbm_group_step call this strip
    // fails (stall_ms 0: an injected LBM_ERR_HIP) or sleeps, at point 0 (before the first rendezvous), 1 or 2 (between rendezvous)
    int debug_fault_launch = -1, debug_fault_point = 0, debug_fault_stall_ms = 0;
    Choreo* rec = nullptr;        // dry run (lbm_debug_choreography): record what would be queued, call no runtime function
    int debug_old_edge_band = 0;  // TEST ONLY: the edge-band height as it was before round 4's fix (the detector must flag it)
    int debug_skip_pull_wait = 0; // TEST ONLY: a launch does not wait for its neighbours' pulls before overwriting its edge rows (the detector must flag it)
    bool owns_comm = true;
    std::shared_ptr<GroupPool> pool;   // the group's host threads (shared by its members)
    int edge_rows[2] = {0, 0};   // edge-band heights of the launch in flight (issue_before -> issue_after)
    bool ext_split_pending = false;   // overlap 2: the edge part of the last extended launch is queued on the side stream (ev_edge)
    // hipGraph replay of launch groups (a strip with a device transport on a deep plan; see replay_groups)
    int use_graph = 1;               // option "graph"
    hipGraphExec_t gexec = nullptr;  // GRAPH_GROUPS consecutive launch groups captured from the eager path
    int gkey[6] = {0, 0, 0, 0, 0, 0};   // what the capture depended on: cur, overlap, deep_halo, deep, use_nt, skip_exchange
    int giters = 0;                  // iterations one replay advances
    bool graph_failed = false;       // capture was refused once (e.g. by the transport): eager from then on
    hipEvent_t gev_main = nullptr, gev_edge = nullptr, gev_comm = nullptr;   // the capture's own events (a captured event must not be waited for eagerly)
    int* d_tbase = nullptr;          // device word the kernels' iteration numbers are relative to (KArgs::t_base)
    int tbase_host = 0;              // its value as of the work queued so far
    long graph_replays = 0;
    char graph_note[128] = "";       // why the graph path was given up, if it was
    // host-staged halo staging (device side)
    double* d_halo = nullptr;  // 4 faces-in-flight x [HR1][9][nx] doubles
};

// A wait that cannot be bounded can at least name itself: a blocking runtime call registers here, and a watchdog thread (started with the
// first registration, one per process, never joined) reports every wait that outlives the default bound — ONE line on stderr and in the
// LBM_TRACE file, with the strip, the stream and the iteration the queue reaches. The call itself is not interrupted.
struct Watchdog {
    struct Item { unsigned long id; char text[200]; std::chrono::steady_clock::time_point t0; bool reported; };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Item> items;
    unsigned long next = 1;
    bool started = false;
    static Watchdog& get() { static Watchdog* w = new Watchdog(); return *w; }       // (leaked on purpose: the thread outlives static destruction)
    unsigned long enter(const char* text) {
        std::lock_guard<std::mutex> lk(mu);
        if (!started) { started = true; std::thread([this] { loop(); }).detach(); }
        Item it{next++, "", std::chrono::steady_clock::now(), false};
        snprintf(it.text, sizeof(it.text), "%s", text);
        items.push_back(it);
        return it.id;
    }
    void leave(unsigned long id) {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t k = 0; k < items.size(); ++k) if (items[k].id == id) { items.erase(items.begin() + (long)k); break; }
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait_for(lk, std::chrono::milliseconds(500));
            const auto now = std::chrono::steady_clock::now();
            for (Item& it : items) {
                const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(now - it.t0).count();
                if (!it.reported && ms > default_wait_timeout_ms()) {
                    it.reported = true;
                    fprintf(stderr, "lbm_hip: STALL: %s — waiting for %ld ms (pid %d)\n", it.text, ms, (int)getpid());
                    fflush(stderr);
                    lbm_trace("STALL", "%s — waiting for %ld ms", it.text, ms);
                }
            }
        }
    }
};

// hipStreamSynchronize with a bound: polls hipStreamQuery (busy for the first 200 us, then yielding, then sleeping 100 us at a time) and
// gives up with LBM_ERR_TIMEOUT after `limit_ms`, saying which stream of which strip was still busy at which iteration. Used where the
// time does not matter (the drains of lbm_initialise / lbm_destroy: default bound) and by lbm_sync when the caller has set
// "wait_timeout_ms". lbm_sync's DEFAULT is the blocking hipStreamSynchronize under the watchdog above: polling the stream while a short
// timed window runs costs throughput (round 4 measured 147.8 against 153.9 GLUPS in the driver's 20-step window, profiles/r04/README.md §3).
inline int wait_stream(const lbm_ctx* c, hipStream_t s, const char* what) {
    const long limit_ms = c->wait_timeout_ms > 0 ? c->wait_timeout_ms : default_wait_timeout_ms();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return LBM_OK;
        if (e != hipErrorNotReady) return fail(LBM_ERR_HIP, "hipStreamQuery(%s) -> %s", what, hipGetErrorString(e));
        (void)hipGetLastError();
        const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (us > limit_ms * 1000L) {
            const int rc = fail(LBM_ERR_TIMEOUT, "%s of the strip of rows %d..%d on device %d still busy after %ld ms (work queued up to iteration %d)",
                                what, c->p.y_start, c->p.y_start + c->nyl, c->device, limit_ms, c->steps_done);
            lbm_trace("TIMEOUT", "%s", g_err);
            return rc;
        }
        if (us > 5000) std::this_thread::sleep_for(std::chrono::microseconds(100));
        else if (us > 200) std::this_thread::yield();
    }
}
// the blocking form, named by the watchdog if it outlives the default bound
inline int sync_stream_blocking(const lbm_ctx* c, hipStream_t s, const char* what) {
    char text[200];
    snprintf(text, sizeof(text), "hipStreamSynchronize(%s) of the strip of rows %d..%d on device %d, work queued up to iteration %d", what, c->p.y_start,
             c->p.y_start + c->nyl, c->device, c->steps_done);
    const unsigned long id = Watchdog::get().enter(text);
    const hipError_t e = hipStreamSynchronize(s);
    Watchdog::get().leave(id);
    if (e != hipSuccess) return fail(LBM_ERR_HIP, "hipStreamSynchronize(%s) -> %s", what, hipGetErrorString(e));
    return LBM_OK;
}
