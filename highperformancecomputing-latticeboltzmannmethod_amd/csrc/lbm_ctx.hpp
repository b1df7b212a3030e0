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

// The host threads of an in-process group of strips (lbm_group_link): one per strip beyond the first, created ONCE and parked
// on a condition variable between lbm_group_step calls (a call used to spawn and join n-1 std::threads: Solver::run issues one
// call per output chunk, ~1 ms of GPU work at N = 8). Strip 0 is driven by the calling thread. Inside a job the n threads move
// in lockstep through `sync`; whether a phase aborts is decided ONCE per rendezvous, in the barrier's completion step, from
// the error state as it stood when the last thread arrived — so every thread takes the same branch and nobody is left
// waiting at the next rendezvous (a thread that failed after a rendezvous used to make a slower one return early).
struct GroupPool {
    struct Snap {
        GroupPool* p;
        void operator()() noexcept { p->phase_err = p->err.load(); }
    };
    const int n;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    const std::function<void(int)>* job = nullptr;
    unsigned long gen = 0;
    int pending = 0;
    bool stop = false;
    std::atomic<int> err{0};
    int phase_err = 0;                 // written by the barrier's completion step only: the same for every thread of a phase
    std::mutex emu;
    std::string msg;
    std::barrier<Snap> sync;
    explicit GroupPool(int n_) : n(n_), sync(n_, Snap{this}) {
        for (int i = 1; i < n; ++i) th.emplace_back([this, i] { loop(i); });
    }
    ~GroupPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_job.notify_all();
        for (auto& t : th) t.join();
    }
    void loop(int i) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)>* f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen; f = job;
            }
            (*f)(i);
            { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cv_done.notify_one(); }
        }
    }
    // run f(0) .. f(n-1), one strip per thread; returns when all are done
    void run(const std::function<void(int)>& f) {
        err.store(0); phase_err = 0; msg.clear();
        { std::lock_guard<std::mutex> lk(mu); job = &f; pending = n - 1; ++gen; }
        cv_job.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    void report(int rc, const char* text) {
        if (rc == 0) return;
        std::lock_guard<std::mutex> lk(emu);
        if (err.load() == 0) { msg = text; err.store(rc); }
    }
    // rendezvous; true: some strip had failed by the time the last one arrived — EVERY thread sees true and leaves
    bool arrive() { sync.arrive_and_wait(); return phase_err != 0; }
};

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
    int variant = 0;     // (unused since round 4: it chose the retired k_step_vec; the option is accepted so that old plan strings still load)
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
    char sched_desc[640] = "";
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

