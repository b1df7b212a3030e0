// csrc/lbm_hip.hip — host side of liblbm_hip.so: the C-ABI of include/lbm_hip.h over the gfx950 kernels in
// lbm_kernels.hpp. Plain HIP runtime + RCCL; no torch types, no CPU fallback.
#include "lbm_kernels.hpp"
#include "lbm_col_api.hpp"
#include "lbm_plan.hpp"
#include "../../include/lbm_hip.h"

#include <rccl/rccl.h>

#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <barrier>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace lbmk;

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
    int variant = 0;     // single-iteration kernel: 0 = k_step_vec when nx % V == 0, 1 = k_step_site
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
    int deep_halo = 1;       // strips: one exchange of GR rows per TWO launches (the first launch of a pair is extended)
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
    double* d_halo = nullptr;  // 4 faces-in-flight x [GR][9][nx] doubles
};

namespace {

template <typename T>
KArgs<T> make_kargs(const lbm_ctx* c, int src, int dst, int t) {
    KArgs<T> a;
    a.src = static_cast<const T*>(c->buf[src]);
    a.dst = static_cast<T*>(c->buf[dst]);
    a.plane = (long)c->plane;
    a.pitch = c->pitch;
    a.xoff = c->xoff;
    a.nx = c->nx;
    a.ny_loc = c->nyl;
    a.ny_glob = c->p.ny;
    a.y_start = c->p.y_start;
    a.cyl_x = c->cyl_x;
    a.cyl_y = c->cyl_y;
    a.cyl_r2 = (double)(c->cyl_r * c->cyl_r);
    a.tau_inv = (T)(1.0 / c->p.tau);
    a.u_in = (T)c->p.inlet_velocity;
    a.unstable_t = c->d_unstable;
    a.t = t - c->tbase_host;
    a.t_base = c->d_tbase;
    a.y_lo = 0;
    a.y_cnt = c->nyl;
    a.y_lo2 = 0;
    a.y_cnt2 = 0;
    a.reverse = 0;
    return a;
}

template <typename T> constexpr int vec_width() { return (int)(16 / sizeof(T)); }

// true: the 16-byte-per-lane kernel k_step_vec runs; false: the generic one-site-per-thread k_step_site
inline bool use_vec(const lbm_ctx* c) {
    const int v = (int)(16 / c->esize);
    if (c->variant == 1) return false;
    return c->nx % v == 0;
}

// Strides of the two layouts. Planar: plane stride = whole rows rounded up to k*64 KiB + 4 KiB (nine planes whose
// stride is a multiple of 64 KiB put the nine accesses of a wave on the same HBM channel group: 5.2-5.4 TB/s at +0
// vs 5.9-6.1 TB/s at +1..8 KiB, 4096x1024 fp64). Row-interleaved: the nine sub-rows of a lattice row are adjacent.
inline void configure_layout(lbm_ctx* c, int layout) {
    c->layout = layout;
    if (layout == 1) {
        c->plane = (size_t)c->pitch0;
        c->pitch = Q * c->pitch0;
        c->total = (size_t)c->pitch * (c->nyl + 2 * GR);
    } else {
        const size_t raw = (size_t)c->pitch0 * (c->nyl + 2 * GR) * c->esize;
        const size_t w = 65536;
        c->plane = ((raw + w - 1) / w * w + 4096) / c->esize;
        c->pitch = c->pitch0;
        c->total = (size_t)Q * c->plane;
    }
}
inline size_t buffer_bytes(const lbm_ctx* c) { return c->total * c->esize + 256; }  // +slack: displaced vector load

// Launch one step-family kernel over the local rows [a.y_lo, a.y_lo + a.y_cnt). Instantiated: MODE_STEP in both store
// policies and both arithmetic modes, MODE_COLLIDE_ONLY in both arithmetic modes, MODE_STREAM_ONLY once (no collision in it).
template <typename T, int MODE>
void launch_rows(const lbm_ctx* c, const KArgs<T>& a, hipStream_t s) {
    constexpr int V = vec_width<T>();
    const bool nt = (MODE == MODE_STEP) && c->use_nt;
    const bool fast = (MODE != MODE_STREAM_ONLY) && c->arith == AR_CONTRACTED;
    const bool vec = use_vec(c);
    const dim3 grid(vec ? (c->nx / V + 255) / 256 : (c->nx + 255) / 256, a.y_cnt + a.y_cnt2), block(256);
#define LBM_K1(NT_, AR_) do { if (vec) hipLaunchKernelGGL((k_step_vec<T, V, MODE, NT_, AR_>), grid, block, 0, s, a); \
                              else hipLaunchKernelGGL((k_step_site<T, MODE, NT_, AR_>), grid, block, 0, s, a); } while (0)
    if constexpr (MODE == MODE_STEP) {
        if (fast) { if (nt) LBM_K1(true, AR_CONTRACTED); else LBM_K1(false, AR_CONTRACTED); }
        else { if (nt) LBM_K1(true, AR_STRICT); else LBM_K1(false, AR_STRICT); }
    } else if constexpr (MODE == MODE_COLLIDE_ONLY) {
        if (fast) LBM_K1(false, AR_CONTRACTED); else LBM_K1(false, AR_STRICT);
    } else {
        LBM_K1(false, AR_STRICT);
    }
#undef LBM_K1
}

// "deep" plans: shape id -> iterations per launch and tile. 1..3: LDS-image tiles (k_stepd_tile: six / seven iterations on
// 64x16 tiles, eight on 32x32; what a grid of a single round of blocks picks); 6 / 7: the register-resident column kernel
// (k_stepc_col, R = 4 rows per thread x 8 waves = a 64 x 32 region per block, two blocks per CU) with five / six iterations —
// a plan of either uses both depths, and on a context without strip faces seven iterations too, for what a segment leaves
// over (20 = 7 + 7 + 6; at 4096x1024 fp64 seven iterations run at 160.6 GLUPS against 161.7 for six — eight, 154.3, are
// not built). Ids 4 / 5 were round 2's 32x16 LDS tiles: retired.
// rows of one band of tiles of a launch of `depth` iterations (the edge bands of a strip are one band each)
inline int deep_rows(const lbm_ctx* c, int id, int depth) {
    if (deep_is_col(id)) return col_tile_h(depth, col_rows_per_thread((int)c->esize, c->arith == 0));
    return id == 3 ? 32 : 16;
}
// A fused kernel over the local rows [a.y_lo, a.y_lo + a.y_cnt): iterations a.t .. a.t + depth - 1 (depth 2..8).
template <typename T>
void launch_fused_rows(const lbm_ctx* c, const KArgs<T>& a, int depth, hipStream_t s) {
    const int shape = c->deep;
    K2Extra<T> e;
    e.feq_in = static_cast<const T*>(c->d_feq);
    e.small = ((c->total + (size_t)c->pitch) * c->esize + 1024 < (size_t(1) << 32)) ? 1 : 0;   // 32-bit byte offsets (+ one row of slack)
    e.xcd = c->xcd;
    e.nt = c->use_nt;
    e.ntl = c->use_ntl;
    const bool fast = c->arith == AR_CONTRACTED;
    if (c->deep_now && deep_is_col(shape)) {    // D iterations with the lattice in registers (k_stepc_col, lbm_col.hip)
        launch_col<T>(a, e, depth, c->use_nt != 0, fast, s);
        return;
    }
    if (c->deep_now) {    // D iterations on a deep LDS tile (k_stepd_tile; whole-domain launches of small grids)
#define LBM_KD(TX_, TY_, D_) do { \
        dim3 gridd((c->nx + TX_ - 1) / TX_, (a.y_cnt + TY_ - 1) / TY_ + (a.y_cnt2 + TY_ - 1) / TY_); \
        if (fast) hipLaunchKernelGGL((k_stepd_tile<T, TX_, TY_, D_, AR_CONTRACTED>), gridd, dim3(TX_ * TY_), 0, s, a, e); \
        else hipLaunchKernelGGL((k_stepd_tile<T, TX_, TY_, D_, AR_STRICT>), gridd, dim3(TX_ * TY_), 0, s, a, e); } while (0)
        switch (shape) {
            case 1: LBM_KD(64, 16, 6); break;
            case 2: LBM_KD(64, 16, 7); break;
            default: LBM_KD(32, 32, 8); break;
        }
#undef LBM_KD
        return;
    }
    const int ty = c->pair_ty;
    dim3 grid((c->nx + 63) / 64, (a.y_cnt + ty - 1) / ty + (a.y_cnt2 + ty - 1) / ty);
#define LBM_KT(K_, TY_, NTH_, G_) do { if (fast) hipLaunchKernelGGL((K_<T, TY_, NTH_, AR_CONTRACTED>), G_, dim3(NTH_), 0, s, a, e); \
                                       else hipLaunchKernelGGL((K_<T, TY_, NTH_, AR_STRICT>), G_, dim3(NTH_), 0, s, a, e); } while (0)
    if (depth == 4) {   // four iterations: 64x8 tiles only (LDS)
        dim3 grid4((c->nx + 63) / 64, (a.y_cnt + 7) / 8 + (a.y_cnt2 + 7) / 8);
        // fp64: 70.5 KB of LDS per block = two blocks per CU, so 1024 threads fill the 32 wave slots; fp32 (35 KB) fills them
        // with four 512-thread blocks (measured: 1024 threads -14 % in fp32, +3 % in fp64)
        constexpr int N4 = sizeof(T) == 8 ? 1024 : 512;
        LBM_KT(k_step4_tile, 8, N4, grid4);
    } else if (depth == 3) {
        if (ty == 12) LBM_KT(k_step3_tile, 12, 1024, grid); else LBM_KT(k_step3_tile, 8, 512, grid);
    } else {
        if (ty == 12) LBM_KT(k_step2_tile, 12, 768, grid); else LBM_KT(k_step2_tile, 8, 512, grid);
    }
#undef LBM_KT
}
inline bool pair_possible(const lbm_ctx*) { return true; }   // partial tiles cover any nx
template <typename T>
int launch_step(lbm_ctx* c, int src, int dst, int t, int mode, hipStream_t s) {
    KArgs<T> a = make_kargs<T>(c, src, dst, t);
    a.reverse = ((mode == MODE_STEP || mode >= 100) && c->alternate && (c->launches_total & 1)) ? 1 : 0;
    switch (mode) {
        case MODE_STEP: launch_rows<T, MODE_STEP>(c, a, s); break;
        case 102: launch_fused_rows<T>(c, a, 2, s); break;     // iterations t, t+1
        case 103: launch_fused_rows<T>(c, a, 3, s); break;     // iterations t, t+1, t+2
        case 104: launch_fused_rows<T>(c, a, 4, s); break;     // iterations t .. t+3 (k_step4_tile, no strip faces)
        case MODE_COLLIDE_ONLY: launch_rows<T, MODE_COLLIDE_ONLY>(c, a, s); break;
        default: break;
    }
    HIPCHK(hipGetLastError());
    return LBM_OK;
}

template <typename T>
int launch_forces(lbm_ctx* c, double* out, int t) {
    ForceArgs<T> f;
    f.cur = static_cast<const T*>(c->buf[c->cur]);
    f.plane = (long)c->plane; f.pitch = c->pitch; f.xoff = c->xoff;
    f.nx = c->nx; f.ny_loc = c->nyl; f.ny_glob = c->p.ny; f.y_start = c->p.y_start;
    f.cyl_x = c->cyl_x; f.cyl_y = c->cyl_y; f.cyl_r = c->cyl_r; f.cyl_r2 = (double)(c->cyl_r * c->cyl_r);
    f.x0 = std::max(0, c->cyl_x - c->cyl_r - 1);
    f.x1 = std::min(c->nx - 1, c->cyl_x + c->cyl_r + 1);
    f.y0 = std::max(0, c->cyl_y - c->cyl_r - 1 - c->p.y_start);
    f.y1 = std::min(c->nyl - 1, c->cyl_y + c->cyl_r + 1 - c->p.y_start);
    f.out = out; f.t = t;
    hipLaunchKernelGGL((k_forces<T>), dim3(1), dim3(1024), 0, c->stream, f);
    HIPCHK(hipGetLastError());
    return LBM_OK;
}

// ---- strip halo exchange ----------------------------------------------------------------------------------
// After a launch has produced the new populations in buf[dst]: my top GR interior rows go to the north neighbour's
// south ghost rows, my bottom GR interior rows to the south neighbour's north ghost rows, all nine populations
// (a fused launch recomputes up to two of the neighbour's rows, which needs every population; per lattice update
// this is the reference's 9 values per edge cell, LBMGrid.h:404-406). Strips always use the row-interleaved layout, in
// which GR rows x 9 sub-rows are ONE contiguous run of GR*pitch elements: one message per face, no packing.
// FaceSpans is the single place the offsets and the count are computed; every transport below uses it.
struct FaceSpans {
    size_t cnt;       // elements per face message (GR rows x pitch)
    long top_rows;    // my top GR interior rows      (gy = nyl .. nyl+GR-1)    -> north neighbour's ghost_s
    long bot_rows;    // my bottom GR interior rows   (gy = GR .. 2GR-1)        -> south neighbour's ghost_n
    long ghost_n;     // my north ghost rows          (gy = nyl+GR .. nyl+2GR-1)
    long ghost_s;     // my south ghost rows          (gy = 0 .. GR-1)
};
inline FaceSpans face_spans(const lbm_ctx* c) {
    FaceSpans f;
    f.cnt = (size_t)GR * c->pitch;
    f.top_rows = (long)c->nyl * c->pitch;
    f.bot_rows = (long)GR * c->pitch;
    f.ghost_n = (long)(c->nyl + GR) * c->pitch;
    f.ghost_s = 0;
    return f;
}

// Transports of ONE context: RCCL send/recv between processes (rank r <-> r-1, r+1), or the test-only loopbacks.
template <typename T>
int exchange_rccl(lbm_ctx* c, int dst, hipStream_t s) {
    if (c->skip_exchange) return LBM_OK;
    const FaceSpans f = face_spans(c);
    T* b = static_cast<T*>(c->buf[dst]);
    const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : ncclFloat;
    if (c->loopback) {   // test transports: my own edge rows become my ghost rows
        if (c->layout != 1) return fail(LBM_ERR_COMM, "loopback requires the row-interleaved layout");
        if (c->loopback == 2) {   // ... through RCCL itself: a one-rank communicator sending to / receiving from rank 0
            if (!c->comm) return fail(LBM_ERR_COMM, "loopback=2 needs lbm_comm_init(c, 0, 1, id)");
            NCCLCHK(ncclGroupStart());        // self send/recv pairs match in posting order
            NCCLCHK(ncclSend(b + f.top_rows, f.cnt, dt, 0, c->comm, s));
            NCCLCHK(ncclRecv(b + f.ghost_s, f.cnt, dt, 0, c->comm, s));
            NCCLCHK(ncclSend(b + f.bot_rows, f.cnt, dt, 0, c->comm, s));
            NCCLCHK(ncclRecv(b + f.ghost_n, f.cnt, dt, 0, c->comm, s));
            NCCLCHK(ncclGroupEnd());
            return LBM_OK;
        }
        const size_t bytes = f.cnt * sizeof(T);                   // ... or plain device copies on the same stream
        HIPCHK(hipMemcpyAsync(b + f.ghost_s, b + f.top_rows, bytes, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(b + f.ghost_n, b + f.bot_rows, bytes, hipMemcpyDeviceToDevice, s));
        return LBM_OK;
    }
    if (c->nranks <= 1) return LBM_OK;
    if (c->layout != 1) return fail(LBM_ERR_COMM, "strips require the row-interleaved layout");
    NCCLCHK(ncclGroupStart());
    if (c->rank + 1 < c->nranks) {
        NCCLCHK(ncclSend(b + f.top_rows, f.cnt, dt, c->rank + 1, c->comm, s));
        NCCLCHK(ncclRecv(b + f.ghost_n, f.cnt, dt, c->rank + 1, c->comm, s));
    }
    if (c->rank > 0) {
        NCCLCHK(ncclSend(b + f.bot_rows, f.cnt, dt, c->rank - 1, c->comm, s));
        NCCLCHK(ncclRecv(b + f.ghost_s, f.cnt, dt, c->rank - 1, c->comm, s));
    }
    NCCLCHK(ncclGroupEnd());
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
    HIPCHK(hipSetDevice(c->device));
    const FaceSpans f = face_spans(c);
    T* b = static_cast<T*>(c->buf[dst]);
    hipStream_t s = exchange_stream(c);
    const size_t bytes = f.cnt * sizeof(T);
    auto pull = [&](lbm_ctx* nb, long nb_rows, long my_ghost) -> int {
        const T* src = static_cast<const T*>(nb->buf[dst]) + nb_rows;
        HIPCHK(hipStreamWaitEvent(s, nb->ev_edge, 0));            // the neighbour's edge rows of this launch are written
        if (nb->device == c->device) HIPCHK(hipMemcpyAsync(b + my_ghost, src, bytes, hipMemcpyDeviceToDevice, s));
        else HIPCHK(hipMemcpyPeerAsync(b + my_ghost, c->device, src, nb->device, bytes, s));
        return LBM_OK;
    };
    if (k > 0) { int rc = pull(cs[k - 1], face_spans(cs[k - 1]).top_rows, f.ghost_s); if (rc) return rc; }
    if (k + 1 < n) { int rc = pull(cs[k + 1], face_spans(cs[k + 1]).bot_rows, f.ghost_n); if (rc) return rc; }
    return LBM_OK;
}

template <typename T>
int exchange_group(lbm_ctx** cs, int n, int dst) {
    if (n < 2 || cs[0]->skip_exchange) return LBM_OK;
    if (cs[0]->group_transport == 1) {
        const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : ncclFloat;
        NCCLCHK(ncclGroupStart());
        for (int k = 0; k < n; ++k) {
            lbm_ctx* c = cs[k];
            const FaceSpans f = face_spans(c);
            T* b = static_cast<T*>(c->buf[dst]);
            hipStream_t s = exchange_stream(c);
            if (k + 1 < n) {
                NCCLCHK(ncclSend(b + f.top_rows, f.cnt, dt, k + 1, c->comm, s));
                NCCLCHK(ncclRecv(b + f.ghost_n, f.cnt, dt, k + 1, c->comm, s));
            }
            if (k > 0) {
                NCCLCHK(ncclSend(b + f.bot_rows, f.cnt, dt, k - 1, c->comm, s));
                NCCLCHK(ncclRecv(b + f.ghost_s, f.cnt, dt, k - 1, c->comm, s));
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
// pairs between halo exchanges: KIND_EXTENDED (first of a pair: the strip's rows plus EXT ghost rows per internal
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
constexpr int EXT = 3;   // rows of each internal face recomputed by the first launch of a pair
inline bool face_south(const lbm_ctx* c) { return c->p.y_start > 0 || c->loopback; }
inline bool face_north(const lbm_ctx* c) { return c->p.y_start + c->nyl < c->p.ny || c->loopback; }

template <typename T>
void launch_depth(const lbm_ctx* c, const KArgs<T>& a, int depth, hipStream_t s) {
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
        const int deep = (c->deep && (!phys_face || deep_depth(c->deep) <= GR)) ? deep_depth(c->deep) : 0;
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
    // (an extended launch recomputes EXT ghost rows and leaves GR - EXT valid ones: launches of up to EXT iterations only;
    // a deep plan with a device transport exchanges after every launch instead)
    if (faces && c->deep_halo && !last && !c->mid_pair && depth <= EXT && !(deep_plan && transport)) L->kind = KIND_EXTENDED;
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
        HIPCHK(hipGetLastError());
        return LBM_OK;
    }
    const int E = c->deep_now ? deep_rows(c, c->deep, L.depth) : L.depth > 1 ? c->pair_ty : GR;   // one tile band
    if (L.kind == KIND_EXTENDED) {   // all rows of the strip PLUS the EXT ghost rows next to each internal face
        const int es = face_south(c) ? EXT : 0, en = face_north(c) ? EXT : 0;
        int e0 = face_south(c) ? E : 0, e1 = face_north(c) ? E : 0;
        if (c->overlap == 2 && c->comm_issued && e0 + e1 < c->nyl) {
            // Schedule 2: the exchange that follows the previous launch is still in flight on the side stream. The rows
            // that do not depend on it start now on the main stream; the bands next to the faces (and the extension)
            // follow the exchange on the side stream. The next launch waits for ev_edge.
            a.y_lo = e0; a.y_cnt = c->nyl - e0 - e1; a.reverse = rev;
            launch_depth<T>(c, a, L.depth, c->stream);
            HIPCHK(hipGetLastError());
            KArgs<T> b = make_kargs<T>(c, L.src, L.dst, L.t);
            b.y_lo = -es; b.y_cnt = e0 + es; b.y_lo2 = c->nyl - e1; b.y_cnt2 = e1 + en;
            if (b.y_cnt == 0) { b.y_lo = b.y_lo2; b.y_cnt = b.y_cnt2; b.y_cnt2 = 0; }
            launch_depth<T>(c, b, L.depth, c->comm_stream);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->ev_edge, c->comm_stream));
            c->ext_split_pending = true;
            return LBM_OK;
        }
        int rc = join_comm(c);       // the last exchange (and the edge bands before it) live on the side stream
        if (rc) return rc;
        a.y_lo = -es;
        a.y_cnt = c->nyl + es + en;
        a.reverse = rev;
        launch_depth<T>(c, a, L.depth, c->stream);
        HIPCHK(hipGetLastError());
        return LBM_OK;
    }
    hipStream_t es = exchange_stream(c);
    auto wait_for_neighbour_pulls = [&](hipStream_t s) -> int {   // group / peer: my edge rows of buf[dst] may still be being read
        for (lbm_ctx* nb : {c->nb_south, c->nb_north})
            if (nb && c->group_transport == 0 && nb->comm_issued) HIPCHK(hipStreamWaitEvent(s, nb->ev_comm, 0));
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
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev_edge, c->stream));
        if (c->overlap == 2) {
            HIPCHK(hipEventRecord(c->ev_main, c->stream));
            HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_main, 0));
        }
        return LBM_OK;
    }
    const bool has_s = face_south(c), has_n = face_north(c);
    // (edge bands on the 64x16 LDS tile of one-cell threads with the interior in registers — the shortest edge launch — were
    // measured: one rank of eight / four / two 8.70 / 11.48 / 17.61 us per iteration against 8.63 / 11.38 / 17.39 for the
    // register kernel throughout: the interior blocks share the CUs with the edge blocks either way. Not kept.)
    int e0 = has_s ? E : 0, e1 = has_n ? E : 0;
    if (e0 + e1 >= c->nyl) { e0 = c->nyl; e1 = 0; }          // short strip: everything is edge
    HIPCHK(hipEventRecord(c->ev_main, c->stream));
    HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_main, 0));
    if (c->comm_issued) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_edge, 0));   // ev_edge still is the previous group's
    int rc = wait_for_neighbour_pulls(es);
    if (rc) return rc;
    a.reverse = 0;
    a.y_lo = 0; a.y_cnt = e0; a.y_lo2 = c->nyl - e1; a.y_cnt2 = e1;
    if (e0 == 0) { a.y_lo = a.y_lo2; a.y_cnt = e1; a.y_cnt2 = 0; }
    launch_depth<T>(c, a, L.depth, c->comm_stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev_edge, c->comm_stream));
    c->edge_rows[0] = e0; c->edge_rows[1] = e1;
    return LBM_OK;
}

// Everything of a launch that follows its exchange, and the bookkeeping.
template <typename T>
int issue_after(lbm_ctx* c, const Launch& L) {
    if (L.kind == KIND_EXCHANGE) {
        HIPCHK(hipEventRecord(c->ev_comm, exchange_stream(c)));
        c->comm_issued = true;
        if (c->overlap == 1) {
            const int e0 = c->edge_rows[0], e1 = c->edge_rows[1];
            if (c->nyl - e0 - e1 > 0) {
                KArgs<T> a = make_kargs<T>(c, L.src, L.dst, L.t);
                a.y_lo = e0; a.y_cnt = c->nyl - e0 - e1;
                a.reverse = (c->alternate && (c->launches_total & 1)) ? 1 : 0;
                launch_depth<T>(c, a, L.depth, c->stream);
                HIPCHK(hipGetLastError());
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
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_edge, 0));
        c->ext_split_pending = false;
    }
    if (c->comm_issued) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
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

// ---- plan: pick layout / kernel / store policy / traversal by measurement --------------------------------
// The step is a pure 18-stream copy with arithmetic attached; which formulation the memory system likes best
// depends on the grid (working set vs the 256 MiB Infinity Cache, row length vs channel interleave) and even on
// where the allocation landed physically (measured: the same planar plan runs at 100 us or 110 us per step at
// 4096x1024 fp64 depending on the allocation). All candidates compute bit-identical results, so lbm_initialise
// times each one on the real buffers (12 warm-up iterations, then the faster of two 36-iteration windows) and keeps the fastest together
// with the very allocation it was measured on.

inline void apply_plan(lbm_ctx* c, const Plan& pl) {
    configure_layout(c, pl.layout);
    c->variant = pl.variant; c->use_nt = pl.nt; c->use_ntl = pl.ntl; c->alternate = pl.alternate; c->fuse = pl.fuse > 0 ? pl.fuse : 1; c->xcd = pl.xcd;
    if (pl.ty) c->pair_ty = pl.ty;
    c->deep = pl.deep;
}

template <typename T>
int time_plan(lbm_ctx* c, float* ms_out, int window = 36) {
    int rc = init_state<T>(c);
    if (rc) return rc;
    auto run = [&](int n) -> int {
        for (int k = 0; k < n;) {
            // far from the end of a call (and with a room that is a multiple of three, so that a three-iteration plan is not
            // handed the four-iteration kernel for a remainder): every launch has the plan's own depth — a window is no
            // multiple of 7 or 8. No strip logic: the probe times local launches.
            const int took = advance<T>(c, 3 * (1 << 18) + (c->trailing_pair ? 0 : 1), 0, false, false);
            if (took < 0) return took;
            k += took;
        }
        return LBM_OK;
    };
    rc = run(12);
    if (rc) return rc;
    *ms_out = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {      // the faster of two windows of `window` iterations (36: six to a dozen fused launches each)
        const int t0 = c->steps_done;
        HIPCHK(hipEventRecord(c->ev_t0, c->stream));
        rc = run(window);
        if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_t1, c->stream));
        HIPCHK(hipEventSynchronize(c->ev_t1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
        *ms_out = std::min(*ms_out, ms / (float)(c->steps_done - t0));     // per iteration
    }
    return LBM_OK;
}

template <typename T>
int choose_plan(lbm_ctx* c) {
    const bool strips = (c->comm && c->nranks > 1) || c->group_n > 1 || c->loopback;
    const Plan fixed = {(strips || c->loopback) ? 1 : c->layout, c->variant, c->use_nt, c->alternate, c->fuse, c->pair_ty, c->xcd,
                        "fixed by options", c->deep, c->use_ntl};
    const bool p2 = pair_possible(c);
    (void)p2;
    const bool vec_ok = (c->nx % vec_width<T>() == 0);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    configure_layout(c, 1);
    const size_t need = 2 * buffer_bytes(c);
    // tiny grids are launch/latency bound (nothing to choose); huge ones cannot afford a second live allocation
    const bool can_tune = c->tune && (size_t)c->nx * c->nyl >= (1u << 16) && 2 * need + (1u << 28) < free_b;
    // the strip rule (lbm_plan.hpp): a function of the global grid and the number of strips only, so that every rank —
    // measuring or not — issues the same launch depths
    const int nstrips = c->group_n > 1 ? c->group_n : (c->comm && c->nranks > 1) ? c->nranks : 1;
    PlanQuery q;
    q.nx = c->nx; q.nyl = c->nyl; q.ny_glob = c->p.ny; q.esize = (int)c->esize; q.num_cus = c->num_cus; q.nstrips = nstrips; q.strips = strips;
    q.vec_ok = vec_ok; q.tune = c->tune != 0; q.can_tune = can_tune; q.faces = face_south(c) || face_north(c);
    const std::vector<Plan> cand = plan_candidates(q, fixed);
    free_buffers(c);                                // a second lbm_initialise starts from no population buffers
    // First round: every candidate once; the three fastest keep their allocations. Final round: those three again with
    // longer windows (candidates within 2 % of each other are common and the first round cannot tell them apart).
    struct Kept { int k; float ms; void* buf[2]; };
    std::vector<Kept> top;
    auto drop_all = [&]() { for (Kept& t : top) for (void*& q : t.buf) if (q) { (void)hipFree(q); q = nullptr; } top.clear(); };
    const bool room = can_tune && 4 * need + (1u << 28) < free_b;     // three kept allocations + the one being probed
    const size_t keep = room ? 3 : 1;
    for (size_t k = 0; k < cand.size(); ++k) {
        apply_plan(c, cand[k]);
        c->buf[0] = c->buf[1] = nullptr;            // keep the best allocations alive while the next one is probed
        int rc = alloc_buffers(c);
        if (rc) { free_buffers(c); drop_all(); return rc; }
        float ms = 0.f;
        if (cand.size() > 1) {
            rc = time_plan<T>(c, &ms);
            if (rc) { free_buffers(c); drop_all(); return rc; }
        }
        top.push_back({(int)k, ms, {c->buf[0], c->buf[1]}});
        c->buf[0] = c->buf[1] = nullptr;
        std::stable_sort(top.begin(), top.end(), [](const Kept& x, const Kept& y) { return x.ms < y.ms; });
        while (top.size() > keep) { for (void* q : top.back().buf) if (q) (void)hipFree(q); top.pop_back(); }
    }
    std::string finalists;
    if (top.size() > 1) {
        for (Kept& t : top) {
            apply_plan(c, cand[(size_t)t.k]);
            c->buf[0] = t.buf[0]; c->buf[1] = t.buf[1];
            // (longer windows: the finalists are often 2-3 % apart — store policy, walk direction — and the alternating walk
            // only shows what it gains from the Infinity Cache once a few launches have gone both ways. Round 3 took the faster
            // of two 120-iteration windows and picked three different plans in three sessions at 16384x4096 fp32: now the MEDIAN
            // of three windows of at least 50 ms each — time_plan returns the faster of two halves, so six in all.)
            const int window = std::max(120, (int)std::ceil(25.0 / std::max(1e-4, (double)t.ms)));
            float w[3] = {0.f, 0.f, 0.f};
            int rc = LBM_OK;
            for (int r = 0; r < 3 && !rc; ++r) rc = time_plan<T>(c, &w[r], window);
            c->buf[0] = c->buf[1] = nullptr;
            if (rc) { drop_all(); return rc; }
            std::sort(w, w + 3);
            t.ms = w[1];
        }
        std::stable_sort(top.begin(), top.end(), [](const Kept& x, const Kept& y) { return x.ms < y.ms; });
        for (const Kept& t : top) {
            char fb[160];
            snprintf(fb, sizeof(fb), "%s%s %.2f", finalists.empty() ? "" : "; ", cand[(size_t)t.k].name.c_str(), t.ms * 1e3f);
            finalists += fb;
        }
        while (top.size() > 1) { for (void* q : top.back().buf) if (q) (void)hipFree(q); top.pop_back(); }
    }
    const int best = top[0].k;
    const float best_ms = top[0].ms;
    void* best_buf[2] = {top[0].buf[0], top[0].buf[1]};
    apply_plan(c, cand[best]);
    c->buf[0] = best_buf[0]; c->buf[1] = best_buf[1];
    c->launches_total = 0;
    c->last_was_pair = false;
    if (cand.size() > 1 && !finalists.empty())
        snprintf(c->plan_desc, sizeof(c->plan_desc), "%s (fastest of %zu measured, %.1f us/iteration; finalists, median of three windows, us/iteration: %s)",
                 cand[best].name.c_str(), cand.size(), best_ms * 1e3f, finalists.c_str());
    else if (cand.size() > 1) snprintf(c->plan_desc, sizeof(c->plan_desc), "%s (fastest of %zu measured, %.1f us/iteration)",
                                       cand[best].name.c_str(), cand.size(), best_ms * 1e3f);
    else snprintf(c->plan_desc, sizeof(c->plan_desc), "%s", cand[best].name.c_str());
    snprintf(c->plan_opts, sizeof(c->plan_opts), "%s", plan_option_string(c->layout, c->variant, c->use_nt, c->alternate, c->pair_ty, c->xcd, c->fuse, c->deep, c->use_ntl).c_str());
    if (cand.size() > 1 && c->deep && !strips && best_ms > 0.f) {
        // What the shallow launches cost on THIS grid and allocation, for plan_launch's split of a segment's last iterations
        // (a single domain only: the strips of a run must all split alike, so they keep the fixed table).
        const Plan keep = cand[best];
        for (int d = 1; d <= 4; ++d) {
            Plan q = keep;
            q.deep = 0; q.fuse = d; q.ty = d == 4 ? 8 : 12;
            apply_plan(c, q);
            configure_layout(c, keep.layout);
            float ms = 0.f;
            int rc = time_plan<T>(c, &ms);
            if (rc) return rc;
            c->depth_rel[d - 1] = std::max(1.0, (double)ms / (double)best_ms);
        }
        c->depth_rel_measured = true;
        apply_plan(c, keep);
        c->launches_total = 0;
        c->last_was_pair = false;
    }
    return LBM_OK;
}

template <typename T> int do_steps(lbm_ctx** cs, int n, int nsteps, int of);
int allreduce_doubles(lbm_ctx* c, double* vals, int n, int op);

// What the ranks must agree on before the collective trials of tune_strip_schedule: packed so that ONE MIN-reduction yields the
// minimum and (negated) the maximum of every pin. v = {go, pin_overlap or -1, -(pin_overlap or -1), pin_deep or -1, -(...)}.
inline void strip_pins_pack(bool go, bool overlap_pinned, int overlap, bool deep_pinned, int deep_halo, double v[5]) {
    const double po = overlap_pinned ? (double)overlap : -1.0, pd = deep_pinned ? (double)deep_halo : -1.0;
    v[0] = go ? 1.0 : 0.0; v[1] = po; v[2] = -po; v[3] = pd; v[4] = -pd;
}
// after the MIN-reduction: false = the ranks disagree (some pinned, some not, or to different values)
inline bool strip_pins_agree(const double v[5], int* go, int* overlap_pinned, int* overlap, int* deep_pinned, int* deep_halo) {
    if (v[1] != -v[2] || v[3] != -v[4]) return false;
    *go = v[0] > 0.5;
    *overlap_pinned = v[1] >= 0.0; if (*overlap_pinned) *overlap = (int)v[1];
    *deep_pinned = v[3] >= 0.0; if (*deep_pinned) *deep_halo = (int)v[3];
    return true;
}

// Strip schedule by measurement (one rank of a multi-process run; collective: every rank runs the same trials and sees
// the same reduced timings, so all ranks choose alike). The schedules — exchange overlapped with the interior rows of the
// same launch (1), of the next, extended launch (2) or serialised (0); one exchange per two launches (deep halo) or per
// launch — compute identical results; which is fastest depends on the strip height and on the link (overlap costs two extra
// launches and three events per group, which a short strip cannot hide). Each candidate: 60 warm-up + 240 timed iterations
// (forty launch groups of six) with the real transport, MAX over the ranks; then the two fastest are timed again, twice,
// and the faster of the two wins (candidates 2-3 % apart are common: round 2's single window of four groups could not rank
// them). What travels per exchange and face is the same in every schedule — GR rows x 9 populations, one contiguous message
// — so the payload per iteration depends on the iterations between two exchanges only; lbm_strip_schedule() reports it.
template <typename T>
int tune_strip_schedule(lbm_ctx* c) {
    const bool multi = c->comm && (c->nranks > 1 || c->loopback == 2);
    auto describe = [&](const char* how, int tried, double us_per_it) {
        // iterations between two exchanges: a deep launch (up to GR iterations) exchanges after every launch; the
        // three-iteration plans after every launch, or after every second one with the deep halo
        const bool deep_launches = c->deep && deep_depth(c->deep) <= GR;
        const int its = deep_launches ? deep_depth(c->deep) : std::min(c->fuse, 3) * (c->deep_halo ? 2 : 1);
        const double face_bytes = (double)GR * c->pitch * c->esize;
        int n = snprintf(c->sched_desc, sizeof(c->sched_desc), "overlap=%d deep_halo=%d (%s", c->overlap, c->deep_halo, how);
        if (tried > 0 && n > 0 && n < (int)sizeof(c->sched_desc))
            n += snprintf(c->sched_desc + n, sizeof(c->sched_desc) - n, " of %d measured, %.2f us/iteration", tried, us_per_it);
        if (n > 0 && n < (int)sizeof(c->sched_desc))
            snprintf(c->sched_desc + n, sizeof(c->sched_desc) - n, "); %.0f B per face and exchange = %.0f B per face and iteration (%d iterations per exchange)",
                     face_bytes, face_bytes / std::max(its, 1), its);
    };
    describe(multi ? "fixed by options" : "default", 0, 0.0);
    if (!multi) return LBM_OK;
    {   // The trials below are COLLECTIVE (send/recv with the neighbours, an all-reduce per schedule): whether they run — and WHICH
        // of them run: a pinned half of the schedule removes trials — must be the same decision on every rank. Strips may differ
        // in height (191 rows over 8 ranks: seven of 24 and one of 23) and, in principle, in their options, so the decision and
        // the pins are reduced over the ranks first (one MIN over {go, pin, -pin, ...}): all of them tune the same list, or the
        // call fails on every rank alike (ranks with different pins ran different numbers of collective trials and
        // the first multi-process lbm_initialise hung in RCCL instead of returning an error).
        double v[5];
        strip_pins_pack(c->tune && c->nyl >= 4 * GR, c->overlap_pinned, c->overlap, c->deep_pinned, c->deep_halo, v);
        int rc = allreduce_doubles(c, v, 5, 2);      // MIN
        if (rc) return rc;
        int go = 0, po = 0, pd = 0, ov = c->overlap, dh = c->deep_halo;
        if (!strip_pins_agree(v, &go, &po, &ov, &pd, &dh))
            return fail(LBM_ERR_ARG, "the ranks of this run pin different strip schedules (lbm_set_option overlap / deep_halo): set the same on every rank");
        c->overlap_pinned = po != 0; c->deep_pinned = pd != 0;
        if (po) c->overlap = ov;
        if (pd) c->deep_halo = dh;
        if (!go || (po && pd)) return LBM_OK;
    }
    const int keep_tp = c->trailing_pair;
    c->trailing_pair = 1;
    constexpr int WARM = 60, TIMED = 240;      // (the warm-up is long enough to take the one-off graph capture of a schedule)
    auto trial = [&](int o, int d, double* worst_ms) -> int {
        c->overlap = o; c->deep_halo = d;
        int rc = do_steps<T>(&c, 1, WARM, 0);
        if (rc) return rc;
        rc = join_comm(c);
        if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_t0, c->stream));
        rc = do_steps<T>(&c, 1, TIMED, 0);
        if (rc) return rc;
        rc = join_comm(c);
        if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_t1, c->stream));
        HIPCHK(hipEventSynchronize(c->ev_t1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
        *worst_ms = (double)ms;
        return allreduce_doubles(c, worst_ms, 1, 1);   // MAX over the ranks: the job advances at the pace of its slowest strip
    };
    static const int variants[5][2] = {{1, 1}, {0, 1}, {2, 1}, {1, 0}, {0, 0}};   // (overlap, deep_halo); 2 needs the deep halo
    struct Res { int o, d; double ms; };
    std::vector<Res> res;
    for (int v = 0; v < 5; ++v) {
        Res r{variants[v][0], variants[v][1], 0.0};
        if (c->overlap_pinned && r.o != c->overlap) continue;        // (a pinned half of the schedule stays as set)
        if (c->deep_pinned && r.d != c->deep_halo) continue;
        int rc = trial(r.o, r.d, &r.ms);
        if (rc) return rc;
        res.push_back(r);
    }
    const int tried = (int)res.size();
    std::string trials;      // every schedule's first-round time (MAX over the ranks), for the log: the margin of the choice
    for (const Res& r : res) {
        char tb[48];
        snprintf(tb, sizeof(tb), "%so%dd%d %.2f", trials.empty() ? "" : ", ", r.o, r.d, r.ms * 1e3 / TIMED);
        trials += tb;
    }
    std::stable_sort(res.begin(), res.end(), [](const Res& x, const Res& y) { return x.ms < y.ms; });
    if (res.size() > 2) res.resize(2);
    if (res.size() == 2) {
        for (Res& r : res) {        // every rank re-times the same two in the same order (the reduced timings are identical everywhere)
            double a = 0.0, b = 0.0;
            int rc = trial(r.o, r.d, &a);
            if (!rc) rc = trial(r.o, r.d, &b);
            if (rc) return rc;
            r.ms = std::min(a, b);
        }
        std::stable_sort(res.begin(), res.end(), [](const Res& x, const Res& y) { return x.ms < y.ms; });
    }
    c->trailing_pair = keep_tp;
    if (!res.empty()) {
        c->overlap = res[0].o; c->deep_halo = res[0].d;
        describe("fastest", tried, res[0].ms * 1e3 / TIMED);
        const size_t n = strlen(c->sched_desc);
        if (n + 1 < sizeof(c->sched_desc))
            snprintf(c->sched_desc + n, sizeof(c->sched_desc) - n, "; first round, us/iteration by (overlap, deep_halo): %s", trials.c_str());
    }
    // back to iteration 0 with fresh halos
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->comm_stream));
    c->mid_pair = false; c->comm_issued = false; c->ext_split_pending = false; c->launches_total = 0; c->last_was_pair = false;
    const int big = INT_MAX;
    HIPCHK(hipMemcpyAsync(c->d_unstable, &big, sizeof(int), hipMemcpyHostToDevice, c->stream));
    int rc = init_state<T>(c);
    if (rc) return rc;
    return exchange_rccl<T>(c, c->cur, c->stream);
}

template <typename T>
int do_initialise(lbm_ctx* c) {
    int rc = choose_plan<T>(c);
    if (rc) return rc;
    const int big = INT_MAX;
    HIPCHK(hipMemcpyAsync(c->d_unstable, &big, sizeof(int), hipMemcpyHostToDevice, c->stream));
    rc = init_state<T>(c);
    if (rc) return rc;
    if ((c->comm || c->loopback) && c->group_n <= 1) {   // (a group exchanges once all members are initialised)
        rc = exchange_rccl<T>(c, c->cur, c->stream);
        if (rc) return rc;
        rc = tune_strip_schedule<T>(c);
        if (rc) return rc;
    }
    return LBM_OK;
}

// ---- hipGraph replay of launch groups ------------------------------------------------------------------------
// A strip of an N = 8 run (4096 x 128) advances six iterations in ~25 us of GPU time, and one launch group — edge-band
// launch, three event records, three cross-stream waits, one RCCL group, interior launch — costs the host 33-43 us to issue:
// the strip is host-bound (round 2's proxy: 7.7 us per iteration against 5.9-6.1 on the GPU). GRAPH_GROUPS consecutive groups
// are therefore captured ONCE from the very code that issues them eagerly (plan_launch / issue_before / exchange_rccl /
// issue_after under hipStreamBeginCapture on the main stream; the side stream joins the capture through the first event wait
// and is joined back before the capture ends) and replayed with one hipGraphLaunch. An even number of groups returns the
// buffer parity, so one graph serves every replay; the kernels' iteration numbers (first-unstable bookkeeping) are relative
// to a device word that the graph itself advances (k_add_int). Where capture is refused — a transport that cannot be
// captured, an in-process group (its cross-device event waits belong to other captures) — the eager path runs as before.
constexpr int GRAPH_GROUPS = 4;
__global__ void k_add_int(int* p, int v) { *p += v; }
__global__ void k_set_int(int* p, int v) { *p = v; }


inline void graph_drop(lbm_ctx* c) {
    if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    c->giters = 0;
}

// May the next `GRAPH_GROUPS` groups of this context be replayed? (a single context with a device transport on a deep plan
// that exchanges after every launch, far from the end of the call and from any force output)
inline bool graph_wanted(const lbm_ctx* c, int remaining, int of, bool transport) {
    if (!c->use_graph || c->graph_failed || !transport || c->group_n > 1) return false;
    // RCCL send/recv between REAL peers under stream capture has never run anywhere (this round's boxes have one GPU; the
    // one-rank communicator sending to itself captures and replays fine): a multi-rank run takes the graph path only when
    // asked to ("graph" 2) — a refused capture falls back, a hang in an untested collective path would not.
    if (c->nranks > 1 && c->use_graph < 2) return false;
    if (!c->deep || deep_depth(c->deep) > GR || c->mid_pair || c->overlap == 2) return false;
    if (!(face_south(c) || face_north(c))) return false;
    const int depth = deep_depth(c->deep), iters = GRAPH_GROUPS * depth;
    if (remaining < iters + 4 * depth + 1) return false;                       // (plan_launch splits the END of a segment differently)
    if (of > 0 && (c->steps_done % of == 0 || of - c->steps_done % of < iters + 4 * depth + 1)) return false;
    return true;
}

// Replay (capturing first, if need be) GRAPH_GROUPS launch groups. Returns the iterations advanced, 0 if the graph path is
// not available (the caller issues eagerly), < 0 on error.
template <typename T>
int replay_groups(lbm_ctx* c, int remaining, int of, bool transport) {
    const int key[6] = {c->cur, c->overlap, c->deep_halo, c->deep, c->use_nt, c->skip_exchange};
    if (c->gexec && memcmp(key, c->gkey, sizeof(key)) != 0) graph_drop(c);
    // everything queued so far, on both streams, precedes the graph: join the side stream into the main one
    int rc = join_comm(c);
    if (rc) return rc;
    if (c->tbase_host != c->steps_done) {      // the graph's launches carry iteration numbers relative to the device word
        hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, c->d_tbase, c->steps_done);
        c->tbase_host = c->steps_done;
    }
    if (!c->gexec) {
        struct Saved { int cur, steps_done; long launches_total; bool comm_issued, mid_pair, last_was_pair, ext_split, restored; int e0, e1;
                       hipEvent_t ev_main, ev_edge, ev_comm; } sv{c->cur, c->steps_done, c->launches_total, c->comm_issued, c->mid_pair,
                       c->last_was_pair, c->ext_split_pending, c->restored, c->edge_rows[0], c->edge_rows[1], c->ev_main, c->ev_edge, c->ev_comm};
        auto restore = [&]() {
            c->cur = sv.cur; c->steps_done = sv.steps_done; c->launches_total = sv.launches_total; c->comm_issued = sv.comm_issued;
            c->mid_pair = sv.mid_pair; c->last_was_pair = sv.last_was_pair; c->ext_split_pending = sv.ext_split; c->restored = sv.restored;
            c->edge_rows[0] = sv.e0; c->edge_rows[1] = sv.e1; c->ev_main = sv.ev_main; c->ev_edge = sv.ev_edge; c->ev_comm = sv.ev_comm;
        };
        // the capture records and waits for its OWN events, and starts with nothing to wait for (joined above)
        c->ev_main = c->gev_main; c->ev_edge = c->gev_edge; c->ev_comm = c->gev_comm;
        c->comm_issued = false; c->ext_split_pending = false;
        hipGraph_t graph = nullptr;
        auto note = [&](const char* what, hipError_t e) {
            if (!c->graph_note[0]) snprintf(c->graph_note, sizeof(c->graph_note), "%s: %s", what, e == hipSuccess ? g_err : hipGetErrorString(e));
        };
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        bool ok = e == hipSuccess;
        if (!ok) note("hipStreamBeginCapture", e);
        int iters = 0;
        if (ok) {
            for (int g = 0; g < GRAPH_GROUPS && ok; ++g) {
                const int took = advance<T>(c, remaining - iters, of, transport, true);
                ok = took == deep_depth(c->deep);
                if (!ok) { char b[64]; snprintf(b, sizeof(b), "group %d took %d iterations", g, took); note(b, hipSuccess); }
                iters += took > 0 ? took : 0;
            }
            if (ok) { ok = join_comm(c) == LBM_OK; if (!ok) note("join", hipSuccess); }   // the side stream rejoins the origin of the capture
            if (ok) hipLaunchKernelGGL(k_add_int, dim3(1), dim3(1), 0, c->stream, c->d_tbase, iters);
            e = hipStreamEndCapture(c->stream, &graph);
            if (ok && (e != hipSuccess || !graph)) note("hipStreamEndCapture", e);
            ok = ok && e == hipSuccess && graph != nullptr;
        }
        (void)hipGetLastError();
        if (ok) { e = hipGraphInstantiate(&c->gexec, graph, nullptr, nullptr, 0); ok = e == hipSuccess; if (!ok) note("hipGraphInstantiate", e); }
        if (graph) (void)hipGraphDestroy(graph);
        const int cur_after = c->cur;
        restore();
        if (ok && cur_after != sv.cur) { ok = false; note("odd number of buffer flips", hipSuccess); }
        if (!ok) {        // refused: eager from now on
            graph_drop(c);
            c->graph_failed = true;
            (void)hipGetLastError();
            return 0;
        }
        memcpy(c->gkey, key, sizeof(key));
        c->giters = iters;
    }
    HIPCHK(hipGraphLaunch(c->gexec, c->stream));
    // the host-side state as the eager path would have left it; every stream of the graph was joined into the main one
    c->steps_done += c->giters;
    c->tbase_host += c->giters;
    c->launches_total += GRAPH_GROUPS;
    c->last_was_pair = true;
    c->restored = false;
    c->comm_issued = false;
    c->ext_split_pending = false;
    c->mid_pair = false;
    c->graph_replays++;
    return c->giters;
}

// `nsteps` iterations of n strips driven in lockstep by this thread (n == 1: a context on its own, which may talk to
// other PROCESSES through its RCCL communicator). Per launch: every member's part before the exchange, the exchange,
// every member's part after it.
template <typename T>
int do_steps(lbm_ctx** cs, int n, int nsteps, int of) {
    for (int i = 0; i < n; ++i) {
        lbm_ctx* c = cs[i];
        HIPCHK(hipSetDevice(c->device));
        if (c->timing) HIPCHK(hipEventRecord(c->ev_t0, c->stream));
        if (c->steps_done != cs[0]->steps_done) return fail(LBM_ERR_ARG, "the strips of a group are at different iterations");
    }
    lbm_ctx* c0 = cs[0];
    const bool transport = n > 1 || (c0->comm && c0->nranks > 1) || c0->loopback;   // a device transport is attached
    int launches = 0;
    std::vector<Launch> L((size_t)n);
    if (n > 1 && c0->group_threads && c0->pool) {
        // One host thread per strip: a launch costs a strip ~10 runtime calls (kernels, events, copies), which one thread
        // issuing for 8 GPUs in turn cannot hide behind 20 us kernels. Three rendezvous per launch: every strip's
        // ev_edge is recorded before anybody pulls, every pull is queued before anybody records ev_comm / launches the
        // interior, and every ev_comm is recorded before the next launch looks at its neighbours'.
        GroupPool& P = *c0->pool;
        const std::function<void(int)> worker = [&](int i) {
            lbm_ctx* c = cs[i];
            (void)hipSetDevice(c->device);
            for (int k = 0; k < nsteps;) {
                const int t = c->steps_done;
                int rc = LBM_OK;
                if (of > 0 && t % of == 0) {
                    if (c->log_count >= c->log_cap) rc = fail(LBM_ERR_ARG, "force log full (%d rows): drain it", c->log_cap);
                    if (!rc) rc = join_comm(c);
                    if (!rc) rc = launch_forces<T>(c, c->d_force_log + 3L * c->log_count, t);
                    if (!rc) c->log_count++;
                }
                if (!rc) rc = plan_launch(c, nsteps - k, of, transport, true, &L[(size_t)i]);
                if (!rc) rc = issue_before<T>(c, L[(size_t)i]);
                P.report(rc, g_err);
                if (P.arrive()) return;
                if (L[(size_t)i].depth != L[0].depth || L[(size_t)i].kind != L[0].kind)
                    P.report(fail(LBM_ERR_ARG, "the strips of a group disagree on the next launch (different options?)"), g_err);
                else if (L[0].kind == KIND_EXCHANGE) {
                    int rc2 = LBM_OK;
                    if (c0->group_transport == 0) rc2 = pull_halos<T>(cs, n, i, L[0].dst);
                    else if (i == 0) rc2 = exchange_group<T>(cs, n, L[0].dst);      // RCCL: one group call, one thread
                    P.report(rc2, g_err);
                }
                if (P.arrive()) return;
                P.report(issue_after<T>(c, L[(size_t)i]), g_err);
                if (P.arrive()) return;
                k += L[(size_t)i].depth;       // (its own copy: strip 0 may already be planning the next launch into L[0])
                if (i == 0) ++launches;
            }
        };
        P.run(worker);
        if (P.err.load() != LBM_OK) return fail(P.err.load(), "%s", P.msg.c_str());
        HIPCHK(hipSetDevice(c0->device));
    } else
    for (int k = 0; k < nsteps;) {
        const int t = c0->steps_done;
        if (n == 1 && graph_wanted(c0, nsteps - k, of, transport)) {
            const int took = replay_groups<T>(c0, nsteps - k, of, transport);
            if (took < 0) return took;
            if (took > 0) { k += took; launches += GRAPH_GROUPS; continue; }
        }
        for (int i = 0; i < n; ++i) {
            lbm_ctx* c = cs[i];
            HIPCHK(hipSetDevice(c->device));
            if (of > 0 && t % of == 0) {
                if (c->log_count >= c->log_cap) return fail(LBM_ERR_ARG, "force log full (%d rows): drain it", c->log_cap);
                int rc = join_comm(c);      // the edge bands of the previous launch live on the side stream
                if (rc) return rc;
                rc = launch_forces<T>(c, c->d_force_log + 3L * c->log_count, t);
                if (rc) return rc;
                c->log_count++;
            }
            int rc = plan_launch(c, nsteps - k, of, transport, true, &L[i]);
            if (rc) return rc;
            if (L[i].depth != L[0].depth || L[i].kind != L[0].kind)
                return fail(LBM_ERR_ARG, "the strips of a group disagree on the next launch (different options?)");
            rc = issue_before<T>(c, L[i]);
            if (rc) return rc;
        }
        if (L[0].kind == KIND_EXCHANGE) {
            int rc = n > 1 ? exchange_group<T>(cs, n, L[0].dst) : exchange_rccl<T>(c0, L[0].dst, exchange_stream(c0));
            if (rc) return rc;
        }
        for (int i = 0; i < n; ++i) {
            HIPCHK(hipSetDevice(cs[i]->device));
            int rc = issue_after<T>(cs[i], L[i]);
            if (rc) return rc;
        }
        k += L[0].depth;
        ++launches;
    }
    for (int i = 0; i < n; ++i) {
        lbm_ctx* c = cs[i];
        if (!c->timing) continue;
        HIPCHK(hipSetDevice(c->device));
        int jr = join_comm(c);
        if (jr) return jr;
        HIPCHK(hipEventRecord(c->ev_t1, c->stream));
        c->timed_launches = launches;
        c->timed_steps = nsteps;
    }
    return LBM_OK;
}

// The halos of a freshly initialised / restored group: every member's edge rows of buf[cur] to its neighbours.
template <typename T>
int refresh_group_halos(lbm_ctx** cs, int n) {
    for (int i = 0; i < n; ++i) {
        lbm_ctx* c = cs[i];
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipStreamSynchronize(c->comm_stream));
        HIPCHK(hipEventRecord(c->ev_edge, exchange_stream(c)));
        if (c->cur != cs[0]->cur) return fail(LBM_ERR_ARG, "the strips of a group are in different buffer phases");
    }
    int rc = exchange_group<T>(cs, n, cs[0]->cur);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        lbm_ctx* c = cs[i];
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipEventRecord(c->ev_comm, exchange_stream(c)));
        c->comm_issued = true;
        c->mid_pair = false;
    }
    for (int i = 0; i < n; ++i) {
        HIPCHK(hipSetDevice(cs[i]->device));
        HIPCHK(hipStreamSynchronize(exchange_stream(cs[i])));
    }
    return LBM_OK;
}

template <typename T>
int do_macros(lbm_ctx* c, bool want_max) {
    const size_t n = (size_t)c->nx * c->nyl;
    if (!c->d_macro) HIPCHK(hipMalloc(&c->d_macro, 3 * n * sizeof(double)));
    MacroArgs<T> m;
    m.old = static_cast<const T*>(c->buf[c->cur ^ 1]);
    m.plane = (long)c->plane; m.pitch = c->pitch; m.xoff = c->xoff;
    m.nx = c->nx; m.ny_loc = c->nyl; m.ny_glob = c->p.ny; m.y_start = c->p.y_start;
    m.cyl_x = c->cyl_x; m.cyl_y = c->cyl_y; m.cyl_r2 = (double)(c->cyl_r * c->cyl_r);
    m.u_in = (T)c->p.inlet_velocity;
    m.initial = (c->steps_done == 0);
    m.rho = c->d_macro; m.ux = c->d_macro + n; m.uy = c->d_macro + 2 * n;
    m.max_usq_bits = want_max ? c->d_maxbits : nullptr;
    if (want_max) HIPCHK(hipMemsetAsync(c->d_maxbits, 0, sizeof(unsigned long long), c->stream));
    dim3 grid((c->nx + 255) / 256, c->nyl), block(256);
    hipLaunchKernelGGL((k_macros<T>), grid, block, 0, c->stream, m);
    HIPCHK(hipGetLastError());
    return LBM_OK;
}

template <typename T>
int do_populations(lbm_ctx* c, int which, double* aos) {
    const int tnx = c->nx + 2, tny = c->nyl + 2;
    const void* srcbuf = c->buf[c->cur ^ 1];
    const bool initial = (c->steps_done == 0);
    if (which == 0 && !initial) {
        if (!c->scratch) HIPCHK(hipMalloc(&c->scratch, buffer_bytes(c)));
        KArgs<T> a = make_kargs<T>(c, c->cur ^ 1, c->cur ^ 1, 0);
        a.dst = static_cast<T*>(c->scratch);
        launch_rows<T, MODE_STREAM_ONLY>(c, a, c->stream);
        HIPCHK(hipGetLastError());
        srcbuf = c->scratch;
    }
    std::vector<T> host(c->total);
    HIPCHK(hipMemcpyAsync(host.data(), srcbuf, c->total * c->esize, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int gy = 0; gy < tny; ++gy)
        for (int gx = 0; gx < tnx; ++gx) {
            const bool ghost = (gy == 0 || gy == tny - 1 || gx == 0 || gx == tnx - 1);
            double* o = aos + ((size_t)gy * tnx + gx) * Q;
            // ghost cells of f_current are never written after Grid::initialise (LBMGrid.h:196-213); those of
            // f_next keep the initial equilibrium until the first exchange_ghost_cells
            const bool analytic = ghost && (which == 0 || initial);
            for (int i = 0; i < Q; ++i)
                o[i] = analytic ? (double)(T)c->feq_in[i]
                                : (double)host[(size_t)i * c->plane + (size_t)(gy + GR - 1) * c->pitch + c->xoff + gx - 1];
        }
    return LBM_OK;
}

template <typename T>
int do_halo_export(lbm_ctx* c, double* south_out, double* north_out) {
    const T* base = static_cast<const T*>(c->buf[c->cur]);
    const size_t n = (size_t)GR * Q * c->nx;
    dim3 grid((c->nx + 255) / 256, GR * Q), block(256);
    if (south_out) {   // my bottom GR interior rows
        hipLaunchKernelGGL((k_halo_pack<T>), grid, block, 0, c->stream, base, (long)c->plane, c->pitch, c->xoff, c->nx, GR,
                           c->d_halo);
        HIPCHK(hipMemcpyAsync(south_out, c->d_halo, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    if (north_out) {   // my top GR interior rows
        hipLaunchKernelGGL((k_halo_pack<T>), grid, block, 0, c->stream, base, (long)c->plane, c->pitch, c->xoff, c->nx,
                           c->nyl, c->d_halo + n);
        HIPCHK(hipMemcpyAsync(north_out, c->d_halo + n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return LBM_OK;
}

template <typename T>
int do_halo_import(lbm_ctx* c, const double* south_in, const double* north_in) {
    T* base = static_cast<T*>(c->buf[c->cur]);
    const size_t n = (size_t)GR * Q * c->nx;
    dim3 grid((c->nx + 255) / 256, GR * Q), block(256);
    if (south_in) {    // -> south ghost rows gy = 0 .. GR-1
        HIPCHK(hipMemcpyAsync(c->d_halo + 2 * n, south_in, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL((k_halo_unpack<T>), grid, block, 0, c->stream, base, (long)c->plane, c->pitch, c->xoff, c->nx, 0,
                           c->d_halo + 2 * n);
    }
    if (north_in) {    // -> north ghost rows gy = nyl+GR .. nyl+2GR-1
        HIPCHK(hipMemcpyAsync(c->d_halo + 3 * n, north_in, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL((k_halo_unpack<T>), grid, block, 0, c->stream, base, (long)c->plane, c->pitch, c->xoff, c->nx,
                           c->nyl + GR, c->d_halo + 3 * n);
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return LBM_OK;
}

// Grid::f_current(x,y,i) written by a client (LBMGrid.h:115): the pre-collision state of the next iteration. The interior
// cells of `aos` are packed into the scratch buffer and re-collided into buf[cur] (collision_step skips solid cells).
template <typename T>
int do_set_f_current(lbm_ctx* c, const double* aos) {
    if (!c->scratch) HIPCHK(hipMalloc(&c->scratch, buffer_bytes(c)));
    std::vector<T> host(c->total);
    const int tnx = c->nx + 2;
    for (int y = 0; y < c->nyl; ++y)
        for (int x = 0; x < c->nx; ++x) {
            const double* v = aos + ((size_t)(y + 1) * tnx + (x + 1)) * Q;
            for (int i = 0; i < Q; ++i) host[(size_t)i * c->plane + (size_t)(y + GR) * c->pitch + c->xoff + x] = (T)v[i];
        }
    HIPCHK(hipMemcpyAsync(c->scratch, host.data(), c->total * c->esize, hipMemcpyHostToDevice, c->stream));
    KArgs<T> a = make_kargs<T>(c, c->cur, c->cur, c->steps_done);
    a.src = static_cast<const T*>(c->scratch);
    launch_rows<T, MODE_COLLIDE_ONLY>(c, a, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LBM_OK;
}

#define DISPATCH(c, call_d, call_f) ((c)->p.precision == LBM_PRECISION_F32 ? (call_f) : (call_d))

}  // namespace

namespace {
int allreduce_doubles(lbm_ctx* c, double* vals, int n, int op) {
    HIPCHK(hipMemcpyAsync(c->d_red, vals, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const ncclRedOp_t rop = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
    NCCLCHK(ncclAllReduce(c->d_red, c->d_red, n, ncclDouble, rop, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(vals, c->d_red, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LBM_OK;
}
}  // namespace

// ---- checkpoint / restart (the reference keeps its state in memory only) ------------------------
// File: header {magic "LBMCKPT1", nx, ny, y_start, local_ny, precision, steps_done, tau, inlet_velocity, cylinder_*}
// followed by the post-collision populations P_{steps_done} of the strip's interior, [9][local_ny][nx] in the
// element type. The state is complete: ghost cells and solid cells are reconstructed by lbm_initialise.
namespace {
struct CkptHeader {
    char magic[8];
    int nx, ny, y_start, local_ny, precision, steps_done;
    double tau, inlet_velocity, cylinder_x, cylinder_y, cylinder_radius;
};

template <typename T>
int do_save(lbm_ctx* c, FILE* fp) {
    std::vector<T> row((size_t)c->nx);
    std::vector<T> host(c->total);
    HIPCHK(hipMemcpyAsync(host.data(), c->buf[c->cur], c->total * c->esize, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < Q; ++i)
        for (int y = 0; y < c->nyl; ++y) {
            const T* src = host.data() + (size_t)i * c->plane + (size_t)(y + GR) * c->pitch + c->xoff;
            if (fwrite(src, sizeof(T), (size_t)c->nx, fp) != (size_t)c->nx) return fail(LBM_ERR_ARG, "short write");
        }
    return LBM_OK;
}

template <typename T>
int do_load(lbm_ctx* c, FILE* fp, int steps_done) {
    // buf[cur] already holds P_0 with all ghost/solid constants in place: overwrite the interior, then rebuild the
    // previous-iteration buffer's ghost frame is not needed (it is only read by snapshots after the next step).
    std::vector<T> host(c->total);
    HIPCHK(hipMemcpyAsync(host.data(), c->buf[c->cur], c->total * c->esize, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < Q; ++i)
        for (int y = 0; y < c->nyl; ++y) {
            T* dst = host.data() + (size_t)i * c->plane + (size_t)(y + GR) * c->pitch + c->xoff;
            if (fread(dst, sizeof(T), (size_t)c->nx, fp) != (size_t)c->nx) return fail(LBM_ERR_ARG, "short read");
        }
    HIPCHK(hipMemcpyAsync(c->buf[c->cur], host.data(), c->total * c->esize, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->steps_done = steps_done;
    c->restored = true;
    return LBM_OK;
}
}  // namespace

extern "C" {

const char* lbm_last_error(void) { return g_err; }

int lbm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lbm_create(const lbm_params* p, int device, lbm_ctx** out) {
    if (!p || !out) return fail(LBM_ERR_ARG, "null argument");
    if (p->nx < 1 || p->ny < 1 || !(p->tau > 0.0)) return fail(LBM_ERR_ARG, "bad nx/ny/tau");
    if (p->precision != LBM_PRECISION_F64 && p->precision != LBM_PRECISION_F32)
        return fail(LBM_ERR_ARG, "bad precision %d", p->precision);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(LBM_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(LBM_ERR_ARG, "device %d out of range (%d devices)", device, ndev);
    lbm_ctx* c = new (std::nothrow) lbm_ctx();
    if (!c) return fail(LBM_ERR_ALLOC, "out of host memory");
    c->p = *p;
    if (c->p.local_ny <= 0) c->p.local_ny = p->ny - p->y_start;
    if (c->p.y_start < 0 || c->p.y_start + c->p.local_ny > p->ny || c->p.local_ny < 1) {
        delete c;
        return fail(LBM_ERR_ARG, "strip [%d,%d) outside [0,%d)", p->y_start, p->y_start + p->local_ny, p->ny);
    }
    if (c->p.local_ny > 65535) {   // one grid row (grid.y) per lattice row in the single-iteration kernels
        delete c;
        return fail(LBM_ERR_ARG, "a strip is limited to 65535 rows (got %d): cut the domain into more strips", p->local_ny);
    }
    c->device = device;
    c->nx = p->nx;
    c->nyl = c->p.local_ny;
    c->esize = p->precision == LBM_PRECISION_F32 ? 4 : 8;
    const int per128 = (int)(128 / c->esize);
    c->xoff = per128;                                      // interior x=0 starts a 128-byte line
    c->pitch0 = round_up(c->xoff + c->nx + 1, per128);     // ghost column x=nx fits, rows stay 128-B aligned
    configure_layout(c, 0);
    // LBMConfig.h:61-65: truncation toward zero
    c->cyl_x = (int)(p->cylinder_x * p->nx);
    c->cyl_y = (int)(p->cylinder_y * p->ny);
    c->cyl_r = (int)(p->cylinder_radius * p->ny);
    {   // f_eq(rho=1, u=(u_in,0)) as Grid::initialise evaluates it (LBMUtils.h:9-12,22-65)
        const double ux = p->inlet_velocity, uy = 0.0, rho = 1.0;
        const double usq = ux * ux + uy * uy, t3 = 1.5 * usq;
        c->feq_in[0] = wgt<double>(0) * rho * (1.0 - 1.5 * usq);
        for (int i = 1; i < Q; ++i) {
            const double cu = (double)cx(i) * ux + (double)cy(i) * uy;
            c->feq_in[i] = (wgt<double>(i) * rho) * (((1.0 + 3.0 * cu) - t3) + 4.5 * (cu * cu));
        }
    }
    c->log_cap = p->force_log_capacity > 0 ? p->force_log_capacity : 4096;
    auto bail = [&](int code) { lbm_destroy(c); return code; };
#define HIPTRY(expr)                                                                                             \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            return bail(fail(LBM_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)));  \
    } while (0)
    HIPTRY(hipSetDevice(device));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus;
    }
    HIPTRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {   // the side stream carries the edge bands + exchange, which sit on the critical path: highest priority
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIPTRY(hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, hi));
    }
    HIPTRY(hipEventCreateWithFlags(&c->ev_edge, hipEventDisableTiming | hipEventDisableSystemFence));   // device-side ordering only
    HIPTRY(hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming | hipEventDisableSystemFence));   // device-side ordering only
    HIPTRY(hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming | hipEventDisableSystemFence));   // device-side ordering only
    HIPTRY(hipEventCreate(&c->ev_t0));
    HIPTRY(hipEventCreate(&c->ev_t1));
    // the population buffers are allocated by lbm_initialise (the plan decides their layout)
    HIPTRY(hipMalloc(&c->d_unstable, sizeof(int)));
    HIPTRY(hipMalloc(&c->d_tbase, sizeof(int)));
    HIPTRY(hipMemset(c->d_tbase, 0, sizeof(int)));
    HIPTRY(hipEventCreateWithFlags(&c->gev_main, hipEventDisableTiming));
    HIPTRY(hipEventCreateWithFlags(&c->gev_edge, hipEventDisableTiming));
    HIPTRY(hipEventCreateWithFlags(&c->gev_comm, hipEventDisableTiming));
    HIPTRY(hipMalloc(&c->d_solid_count, sizeof(int)));
    HIPTRY(hipMalloc(&c->d_maxbits, sizeof(unsigned long long)));
    HIPTRY(hipMalloc(&c->d_force_now, 3 * sizeof(double)));
    HIPTRY(hipMalloc(&c->d_feq, Q * sizeof(double)));
    if (p->precision == LBM_PRECISION_F32) {
        float v[Q];
        for (int i = 0; i < Q; ++i) v[i] = (float)c->feq_in[i];
        HIPTRY(hipMemcpy(c->d_feq, v, sizeof(v), hipMemcpyHostToDevice));
    } else {
        HIPTRY(hipMemcpy(c->d_feq, c->feq_in, Q * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPTRY(hipMalloc(&c->d_force_log, 3 * sizeof(double) * c->log_cap));
    HIPTRY(hipMalloc(&c->d_halo, 4 * GR * Q * sizeof(double) * (size_t)c->nx));
    HIPTRY(hipMalloc(&c->d_red, 64 * sizeof(double)));
#undef HIPTRY
    *out = c;
    return LBM_OK;
}

void lbm_destroy(lbm_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    // (a graph that captured RCCL operations holds the communicator: ncclCommDestroy waits for it to go away first)
    if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    if (c->comm) ncclCommDestroy(c->comm);
    for (lbm_ctx* nb : {c->nb_south, c->nb_north}) {  // a destroyed member leaves its group
        if (!nb) continue;
        // the neighbour's exchange stream may still hold a pull (hipMemcpyPeerAsync) that READS this member's edge rows, and
        // freeing a buffer only waits for work of this member's own device: drain the neighbour's streams first
        (void)hipSetDevice(nb->device);
        if (nb->comm_stream) (void)hipStreamSynchronize(nb->comm_stream);
        if (nb->stream) (void)hipStreamSynchronize(nb->stream);
        if (nb->nb_south == c) nb->nb_south = nullptr;
        if (nb->nb_north == c) nb->nb_north = nullptr;
    }
    (void)hipSetDevice(c->device);
    c->pool.reset();
    void* ptrs[] = {c->buf[0], c->buf[1], c->scratch, c->d_macro, c->d_maxbits, c->d_unstable, c->d_tbase, c->d_solid_count, c->d_feq,
                    c->d_force_now, c->d_force_log, c->d_halo, c->d_red};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    hipEvent_t evs[] = {c->ev_edge, c->ev_comm, c->ev_main, c->ev_t0, c->ev_t1, c->gev_main, c->gev_edge, c->gev_comm};
    for (hipEvent_t e : evs)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    delete c;
}

int lbm_initialise(lbm_ctx* c, int* solid_count_out) {
    if (!c) return fail(LBM_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));        // (re-)initialisation starts from quiet streams
    HIPCHK(hipStreamSynchronize(c->comm_stream));
    c->steps_done = 0;
    c->log_count = 0;
    c->mid_pair = false;
    c->comm_issued = false;
    c->ext_split_pending = false;
    graph_drop(c);                 // (a captured graph holds the addresses of the buffers this call is about to replace)
    c->graph_failed = false;
    c->graph_note[0] = 0;
    if (c->scratch) { (void)hipFree(c->scratch); c->scratch = nullptr; }
    int rc = DISPATCH(c, do_initialise<double>(c), do_initialise<float>(c));
    if (rc) return rc;
    int sc = 0;
    HIPCHK(hipMemcpyAsync(&sc, c->d_solid_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (solid_count_out) *solid_count_out = sc;
    c->initialised = true;
    return LBM_OK;
}

int lbm_step(lbm_ctx* c, int nsteps, int output_frequency) {
    if (!c || !c->initialised) return fail(LBM_ERR_ARG, "context not initialised");
    if (nsteps < 0) return fail(LBM_ERR_ARG, "nsteps < 0");
    if (c->group_n > 1) return fail(LBM_ERR_ARG, "this context is a member of a group: use lbm_group_step");
    HIPCHK(hipSetDevice(c->device));
    return DISPATCH(c, do_steps<double>(&c, 1, nsteps, output_frequency), do_steps<float>(&c, 1, nsteps, output_frequency));
}

int lbm_sync(lbm_ctx* c) {
    if (!c) return fail(LBM_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->comm_stream));
    return LBM_OK;
}

int lbm_steps_done(const lbm_ctx* c) { return c ? c->steps_done : -1; }

int lbm_first_unstable_step(lbm_ctx* c, int* t_out) {
    if (!c || !t_out) return fail(LBM_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    { int jr = join_comm(c); if (jr) return jr; }
    int v = INT_MAX;
    HIPCHK(hipMemcpyAsync(&v, c->d_unstable, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *t_out = (v == INT_MAX) ? -1 : v;
    return LBM_OK;
}

int lbm_get_forces(lbm_ctx* c, double* fx, double* fy) {
    if (!c || !c->initialised) return fail(LBM_ERR_ARG, "context not initialised");
    HIPCHK(hipSetDevice(c->device));
    { int jr = join_comm(c); if (jr) return jr; }
    int rc = DISPATCH(c, launch_forces<double>(c, c->d_force_now, c->steps_done),
                      launch_forces<float>(c, c->d_force_now, c->steps_done));
    if (rc) return rc;
    double h[3];
    HIPCHK(hipMemcpyAsync(h, c->d_force_now, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (fx) *fx = h[1];
    if (fy) *fy = h[2];
    return LBM_OK;
}

int lbm_drain_force_log(lbm_ctx* c, lbm_force_row* rows, int max_rows) {
    if (!c || (!rows && max_rows > 0)) return fail(LBM_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    const int n = std::min(max_rows, c->log_count);
    if (n < c->log_count) return fail(LBM_ERR_ARG, "force log holds %d rows, buffer takes %d", c->log_count, max_rows);
    std::vector<double> h(3 * (size_t)std::max(n, 1));
    if (n > 0) HIPCHK(hipMemcpyAsync(h.data(), c->d_force_log, 3 * sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < n; ++k) {
        rows[k].timestep = (int)h[3 * k];
        rows[k].fx = h[3 * k + 1];
        rows[k].fy = h[3 * k + 2];
    }
    c->log_count = 0;
    return n;
}

int lbm_get_macros(lbm_ctx* c, double* rho, double* ux, double* uy) {
    if (!c || !c->initialised) return fail(LBM_ERR_ARG, "context not initialised");
    if (c->last_was_pair || c->restored) return fail(LBM_ERR_ARG, "snapshot unavailable: the previous iteration's populations are not resident (fused trailing launch or restored state); take one more lbm_step(c,1,..)");
    HIPCHK(hipSetDevice(c->device));
    { int jr = join_comm(c); if (jr) return jr; }
    int rc = DISPATCH(c, do_macros<double>(c, false), do_macros<float>(c, false));
    if (rc) return rc;
    const size_t n = (size_t)c->nx * c->nyl;
    if (rho) HIPCHK(hipMemcpyAsync(rho, c->d_macro, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (ux) HIPCHK(hipMemcpyAsync(ux, c->d_macro + n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (uy) HIPCHK(hipMemcpyAsync(uy, c->d_macro + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LBM_OK;
}

int lbm_max_velocity_sq(lbm_ctx* c, double* out) {
    if (!c || !c->initialised || !out) return fail(LBM_ERR_ARG, "bad argument");
    if (c->last_was_pair || c->restored) return fail(LBM_ERR_ARG, "snapshot unavailable: the previous iteration's populations are not resident (fused trailing launch or restored state); take one more lbm_step(c,1,..)");
    HIPCHK(hipSetDevice(c->device));
    { int jr = join_comm(c); if (jr) return jr; }
    int rc = DISPATCH(c, do_macros<double>(c, true), do_macros<float>(c, true));
    if (rc) return rc;
    unsigned long long bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, c->d_maxbits, sizeof(bits), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(out, &bits, sizeof(double));
    return LBM_OK;
}

int lbm_get_populations(lbm_ctx* c, int which, double* aos) {
    if (!c || !c->initialised || !aos || (which != 0 && which != 1)) return fail(LBM_ERR_ARG, "bad argument");
    if (c->last_was_pair || c->restored) return fail(LBM_ERR_ARG, "snapshot unavailable: the previous iteration's populations are not resident (fused trailing launch or restored state); take one more lbm_step(c,1,..)");
    HIPCHK(hipSetDevice(c->device));
    { int jr = join_comm(c); if (jr) return jr; }
    return DISPATCH(c, do_populations<double>(c, which, aos), do_populations<float>(c, which, aos));
}

int lbm_set_f_current(lbm_ctx* c, const double* aos) {
    if (!c || !c->initialised || !aos) return fail(LBM_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->comm_stream));
    int rc = DISPATCH(c, do_set_f_current<double>(c, aos), do_set_f_current<float>(c, aos));
    if (rc) return rc;
    if ((c->comm || c->loopback) && c->group_n <= 1)   // (a group: lbm_group_refresh_halos once every member is set)
        rc = DISPATCH(c, exchange_rccl<double>(c, c->cur, c->stream), exchange_rccl<float>(c, c->cur, c->stream));
    return rc;
}

int lbm_get_solid(lbm_ctx* c, unsigned char* mask) {
    if (!c || !mask) return fail(LBM_ERR_ARG, "null argument");
    const double r2 = (double)(c->cyl_r * c->cyl_r);
    for (int y = 0; y < c->nyl; ++y)
        for (int x = 0; x < c->nx; ++x) {
            const double dx = x - c->cyl_x, dy = (c->p.y_start + y) - c->cyl_y;
            mask[(size_t)y * c->nx + x] = (dx * dx + dy * dy <= r2) ? 1 : 0;
        }
    return LBM_OK;
}

int lbm_comm_unique_id(void* id128) {
    if (!id128) return fail(LBM_ERR_ARG, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return LBM_OK;
}

int lbm_comm_init(lbm_ctx* c, int rank, int nranks, const void* id128) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(LBM_ERR_ARG, "bad argument");
    if (c->initialised) return fail(LBM_ERR_ARG, "attach the communicator before lbm_initialise");
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    NCCLCHK(ncclCommInitRank(&c->comm, nranks, id, rank));
    c->rank = rank;
    c->nranks = nranks;
    if (nranks > 1 && c->nyl < 2 * GR) return fail(LBM_ERR_ARG, "a strip with neighbours needs at least %d rows", 2 * GR);
    return LBM_OK;
}

int lbm_comm_allreduce(lbm_ctx* c, double* vals, int n, int op) {
    if (!c || !vals || n < 1 || n > 64) return fail(LBM_ERR_ARG, "bad argument");
    if (!c->comm) return (c->nranks == 1) ? LBM_OK : fail(LBM_ERR_COMM, "no communicator");
    HIPCHK(hipSetDevice(c->device));
    return allreduce_doubles(c, vals, n, op);
}

int lbm_runtime_versions(int* rccl, int* hip_runtime, int* hip_driver) {
    int v = 0;
    if (rccl) { NCCLCHK(ncclGetVersion(&v)); *rccl = v; }
    if (hip_runtime) { HIPCHK(hipRuntimeGetVersion(&v)); *hip_runtime = v; }
    if (hip_driver) { HIPCHK(hipDriverGetVersion(&v)); *hip_driver = v; }
    return LBM_OK;
}

const char* lbm_strip_schedule(const lbm_ctx* c) {
    if (!c) return "";
    static thread_local char out[448];
    if (c->graph_failed) snprintf(out, sizeof(out), "%s; launch groups issued call by call (graph capture given up: %s)", c->sched_desc, c->graph_note);
    else if (c->graph_replays > 0) snprintf(out, sizeof(out), "%s; %ld hipGraph replays of %d launch groups", c->sched_desc, c->graph_replays, GRAPH_GROUPS);
    else snprintf(out, sizeof(out), "%s", c->sched_desc);
    return out;
}

int lbm_device_memory(int device, unsigned long long* free_bytes, unsigned long long* total_bytes) {
    size_t f = 0, t = 0;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return LBM_OK;
}

// ---- in-process groups of strips ----------------------------------------------------------------------------
namespace {
int check_group(lbm_ctx** cs, int n, bool linked) {
    if (!cs || n < 1) return fail(LBM_ERR_ARG, "empty group");
    for (int k = 0; k < n; ++k) {
        if (!cs[k]) return fail(LBM_ERR_ARG, "null context in group");
        if (linked && (cs[k]->group_n != n || cs[k]->group_k != k)) return fail(LBM_ERR_ARG, "not the group these contexts were linked as");
    }
    return LBM_OK;
}
}  // namespace

int lbm_group_link(lbm_ctx** cs, int n, int transport) {
    int rc = check_group(cs, n, false);
    if (rc) return rc;
    if (transport != 0 && transport != 1) return fail(LBM_ERR_ARG, "transport must be 0 (peer copies) or 1 (RCCL)");
    for (int k = 0; k < n; ++k) {
        lbm_ctx* c = cs[k];
        if (c->initialised || c->comm || c->group_n > 1) return fail(LBM_ERR_ARG, "link fresh contexts (before lbm_initialise, without a communicator)");
        if (c->p.precision != cs[0]->p.precision || c->nx != cs[0]->nx || c->p.ny != cs[0]->p.ny)
            return fail(LBM_ERR_ARG, "the strips of a group must share nx, ny and the precision");
        const int expect = k == 0 ? 0 : cs[k - 1]->p.y_start + cs[k - 1]->nyl;
        if (c->p.y_start != expect) return fail(LBM_ERR_ARG, "strip %d starts at row %d, expected %d (bottom to top, contiguous)", k, c->p.y_start, expect);
        if (n > 1 && c->nyl < 2 * GR) return fail(LBM_ERR_ARG, "a strip with neighbours needs at least %d rows", 2 * GR);
    }
    if (cs[n - 1]->p.y_start + cs[n - 1]->nyl != cs[0]->p.ny) return fail(LBM_ERR_ARG, "the strips do not cover all %d rows", cs[0]->p.ny);
    if (n == 1) return LBM_OK;
    if (transport == 1) {   // one communicator per member, created together (ncclCommInitAll wants distinct devices)
        std::vector<int> devs((size_t)n);
        std::vector<ncclComm_t> comms((size_t)n);
        for (int k = 0; k < n; ++k) devs[(size_t)k] = cs[k]->device;
        NCCLCHK(ncclCommInitAll(comms.data(), n, devs.data()));
        for (int k = 0; k < n; ++k) { cs[k]->comm = comms[(size_t)k]; cs[k]->rank = k; cs[k]->nranks = n; }
    } else {
        for (int k = 0; k < n; ++k) {
            lbm_ctx* c = cs[k];
            HIPCHK(hipSetDevice(c->device));
            for (lbm_ctx* nb : {k > 0 ? cs[k - 1] : nullptr, k + 1 < n ? cs[k + 1] : nullptr}) {
                if (!nb || nb->device == c->device) continue;
                int can = 0;
                HIPCHK(hipDeviceCanAccessPeer(&can, c->device, nb->device));
                if (can) {
                    const hipError_t e = hipDeviceEnablePeerAccess(nb->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(LBM_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d): %s", c->device, nb->device, hipGetErrorString(e));
                    (void)hipGetLastError();
                }   // (without peer access hipMemcpyPeerAsync stages through the host)
            }
            // the neighbours' streams wait on these events and read what the kernels before them wrote: system-scope release
            for (hipEvent_t* ev : {&c->ev_edge, &c->ev_comm}) {
                HIPCHK(hipEventDestroy(*ev));
                HIPCHK(hipEventCreateWithFlags(ev, hipEventDisableTiming));
            }
        }
    }
    std::shared_ptr<GroupPool> pool(new (std::nothrow) GroupPool(n));
    if (!pool) return fail(LBM_ERR_ALLOC, "out of host memory");
    for (int k = 0; k < n; ++k) {
        cs[k]->nb_south = k > 0 ? cs[k - 1] : nullptr;
        cs[k]->nb_north = k + 1 < n ? cs[k + 1] : nullptr;
        cs[k]->group_transport = transport;
        cs[k]->group_n = n;
        cs[k]->group_k = k;
        cs[k]->pool = pool;
    }
    return LBM_OK;
}

int lbm_group_initialise(lbm_ctx** cs, int n, int* solid_total) {
    int rc = check_group(cs, n, true);
    if (rc) return rc;
    int total = 0;
    for (int k = 0; k < n; ++k) {
        int sc = 0;
        rc = lbm_initialise(cs[k], &sc);
        if (rc) return rc;
        total += sc;
    }
    if (solid_total) *solid_total = total;
    if (n > 1) rc = DISPATCH(cs[0], refresh_group_halos<double>(cs, n), refresh_group_halos<float>(cs, n));
    return rc;
}

int lbm_group_step(lbm_ctx** cs, int n, int nsteps, int output_frequency) {
    int rc = check_group(cs, n, true);
    if (rc) return rc;
    if (nsteps < 0) return fail(LBM_ERR_ARG, "nsteps < 0");
    for (int k = 0; k < n; ++k)
        if (!cs[k]->initialised) return fail(LBM_ERR_ARG, "context not initialised");
    return DISPATCH(cs[0], do_steps<double>(cs, n, nsteps, output_frequency), do_steps<float>(cs, n, nsteps, output_frequency));
}

int lbm_group_refresh_halos(lbm_ctx** cs, int n) {
    int rc = check_group(cs, n, true);
    if (rc || n == 1) return rc;
    return DISPATCH(cs[0], refresh_group_halos<double>(cs, n), refresh_group_halos<float>(cs, n));
}

int lbm_halo_export(lbm_ctx* c, double* south_out, double* north_out) {
    if (!c || !c->initialised) return fail(LBM_ERR_ARG, "context not initialised");
    HIPCHK(hipSetDevice(c->device));
    { int jr = join_comm(c); if (jr) return jr; }
    return DISPATCH(c, do_halo_export<double>(c, south_out, north_out), do_halo_export<float>(c, south_out, north_out));
}

int lbm_halo_import(lbm_ctx* c, const double* south_in, const double* north_in) {
    if (!c || !c->initialised) return fail(LBM_ERR_ARG, "context not initialised");
    HIPCHK(hipSetDevice(c->device));
    { int jr = join_comm(c); if (jr) return jr; }
    return DISPATCH(c, do_halo_import<double>(c, south_in, north_in), do_halo_import<float>(c, south_in, north_in));
}

int lbm_save_state(lbm_ctx* c, const char* path) {
    if (!c || !c->initialised || !path) return fail(LBM_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    { int jr = join_comm(c); if (jr) return jr; }
    FILE* fp = fopen(path, "wb");
    if (!fp) return fail(LBM_ERR_ARG, "cannot open %s for writing", path);
    CkptHeader h{};
    memcpy(h.magic, "LBMCKPT1", 8);
    h.nx = c->nx; h.ny = c->p.ny; h.y_start = c->p.y_start; h.local_ny = c->nyl; h.precision = c->p.precision;
    h.steps_done = c->steps_done; h.tau = c->p.tau; h.inlet_velocity = c->p.inlet_velocity;
    h.cylinder_x = c->p.cylinder_x; h.cylinder_y = c->p.cylinder_y; h.cylinder_radius = c->p.cylinder_radius;
    int rc = fwrite(&h, sizeof(h), 1, fp) == 1 ? LBM_OK : fail(LBM_ERR_ARG, "short write");
    if (!rc) rc = DISPATCH(c, do_save<double>(c, fp), do_save<float>(c, fp));
    fclose(fp);
    return rc;
}

int lbm_load_state(lbm_ctx* c, const char* path) {
    if (!c || !c->initialised || !path) return fail(LBM_ERR_ARG, "lbm_load_state needs an initialised context");
    HIPCHK(hipSetDevice(c->device));
    // the state is replaced wholesale: nothing of the run so far may still be in flight on either stream
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->comm_stream));
    c->comm_issued = false;
    c->ext_split_pending = false;
    {
        const int big = INT_MAX;
        HIPCHK(hipMemcpyAsync(c->d_unstable, &big, sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(LBM_ERR_ARG, "cannot open %s", path);
    CkptHeader h{};
    int rc = LBM_OK;
    if (fread(&h, sizeof(h), 1, fp) != 1 || memcmp(h.magic, "LBMCKPT1", 8) != 0) rc = fail(LBM_ERR_ARG, "%s is not a checkpoint", path);
    else if (h.nx != c->nx || h.ny != c->p.ny || h.y_start != c->p.y_start || h.local_ny != c->nyl ||
             h.precision != c->p.precision || h.tau != c->p.tau || h.inlet_velocity != c->p.inlet_velocity ||
             h.cylinder_x != c->p.cylinder_x || h.cylinder_y != c->p.cylinder_y || h.cylinder_radius != c->p.cylinder_radius)
        rc = fail(LBM_ERR_ARG, "checkpoint %s was written for different parameters", path);
    if (!rc) rc = DISPATCH(c, do_load<double>(c, fp, h.steps_done), do_load<float>(c, fp, h.steps_done));
    fclose(fp);
    if (rc) return rc;
    c->log_count = 0;
    c->last_was_pair = false;
    c->mid_pair = false;
    if ((c->comm || c->loopback) && c->group_n <= 1)   // (a group: lbm_group_refresh_halos once every member is restored)
        rc = DISPATCH(c, exchange_rccl<double>(c, c->cur, c->stream), exchange_rccl<float>(c, c->cur, c->stream));
    return rc;
}

int lbm_set_option(lbm_ctx* c, const char* key, long value) {
    if (!c || !key) return fail(LBM_ERR_ARG, "null argument");
    const std::string k(key);
    if (c->initialised && (k == "variant" || k == "layout" || k == "nt" || k == "ntl" || k == "tune" || k == "pair" || k == "fuse" || k == "pair_ty" || k == "loopback" || k == "deep" || k == "arith"))
        return fail(LBM_ERR_ARG, "option %s must be set before lbm_initialise", key);
    if (k == "variant") c->variant = (int)value;
    else if (k == "timing") c->timing = (int)value;
    else if (k == "alternate") c->alternate = (int)value;
    else if (k == "layout") c->layout = (int)value ? 1 : 0;
    else if (k == "nt") c->use_nt = (int)value ? 1 : 0;
    else if (k == "ntl") c->use_ntl = (int)value ? 1 : 0;
    else if (k == "fuse") { if (value < 1 || value > 4) return fail(LBM_ERR_ARG, "fuse must be 1, 2, 3 or 4 (4: tile kernel, no strip faces)"); c->fuse = (int)value; c->deep = 0; }
    else if (k == "deep") {     // k_stepd_tile: 1: 6 iterations on 64x16 tiles, 2: 7 on 64x16, 3: 8 on 32x32 (1024 threads);
                                // k_stepc_col (registers): 6 / 7: 5 / 6 iterations on 64x32 regions. 2 and 3: whole-domain launches only.
        if (!deep_valid((int)value)) return fail(LBM_ERR_ARG, "deep must be 0..3, 6 or 7 (4 / 5, round 2's 32x16 LDS tiles, are retired)");
        c->deep = (int)value;
        if (c->deep) c->fuse = deep_depth(c->deep);
    }
    else if (k == "pair") c->fuse = (int)value ? 2 : 1;
    else if (k == "trailing_pair") c->trailing_pair = (int)value ? 1 : 0;
    else if (k == "xcd") c->xcd = (int)value ? 1 : 0;
    else if (k == "arith") { if (value != 0 && value != 1) return fail(LBM_ERR_ARG, "arith must be 0 (strict) or 1 (contracted)"); c->arith = (int)value; }
    else if (k == "deep_halo") { c->deep_halo = (int)value ? 1 : 0; c->deep_pinned = true; }
    else if (k == "skip_exchange") c->skip_exchange = (int)value ? 1 : 0;
    else if (k == "group_threads") c->group_threads = (int)value ? 1 : 0;
    else if (k == "graph") { if (value < 0 || value > 2) return fail(LBM_ERR_ARG, "graph must be 0, 1 or 2"); c->use_graph = (int)value; }
    else if (k == "loopback") c->loopback = (int)value;   // 0 off, 1 device copies, 2 RCCL self send/recv
    else if (k == "pair_ty") { if (value != 8 && value != 12) return fail(LBM_ERR_ARG, "pair_ty must be 8 or 12"); c->pair_ty = (int)value; }
    else if (k == "tune") c->tune = (int)value ? 1 : 0;
    else if (k == "overlap") { if (value < 0 || value > 2) return fail(LBM_ERR_ARG, "overlap must be 0, 1 or 2"); c->overlap = (int)value; c->overlap_pinned = true; }
    else return fail(LBM_ERR_ARG, "unknown option %s", key);
    return LBM_OK;
}

int lbm_last_step_kernel_ms(lbm_ctx* c, double* ms_per_launch) {
    if (!c || !ms_per_launch) return fail(LBM_ERR_ARG, "null argument");
    *ms_per_launch = 0.0;
    if (!c->timing || c->timed_launches <= 0) return LBM_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->ev_t1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
    *ms_per_launch = (double)ms / c->timed_launches;
    return LBM_OK;
}

int lbm_last_step_stats(lbm_ctx* c, double* ms_total, int* launches, int* iterations) {
    if (!c) return fail(LBM_ERR_ARG, "null argument");
    if (ms_total) *ms_total = 0.0;
    if (launches) *launches = c->timed_launches;
    if (iterations) *iterations = c->timed_steps;
    if (!c->timing || c->timed_launches <= 0) return LBM_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->ev_t1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
    if (ms_total) *ms_total = (double)ms;
    return LBM_OK;
}

long lbm_graph_replays(const lbm_ctx* c) { return c ? c->graph_replays : 0; }

const char* lbm_kernel_name(const lbm_ctx* c) {
    if (!c) return "";
    static thread_local char name[96];
    snprintf(name, sizeof(name), "%s", plan_kernel_name(c->fuse, c->deep, c->pair_ty, c->use_nt, c->arith, (int)c->esize, use_vec(c)).c_str());
    return name;
}

/* TEST HOOK (no device needed): the candidates lbm_initialise would time for a whole-domain context of this grid, one per line:
 * "name|lbm_set_option pairs|dominant kernel|iterations per launch". */
int lbm_debug_plan_candidates(int nx, int ny, int precision, int arith, int num_cus, char* out, int cap) {
    if (!out || cap < 1 || nx < 1 || ny < 1) return fail(LBM_ERR_ARG, "bad argument");
    PlanQuery q;
    q.nx = nx; q.nyl = ny; q.ny_glob = ny; q.esize = precision == LBM_PRECISION_F32 ? 4 : 8; q.num_cus = num_cus > 0 ? num_cus : 256;
    q.vec_ok = nx % (16 / q.esize) == 0;
    const Plan none{};
    std::string text;
    for (const Plan& pl : plan_candidates(q, none)) {
        const int fuse = pl.fuse > 0 ? pl.fuse : 1;
        text += pl.name + "|" + plan_option_string(pl.layout, pl.variant, pl.nt, pl.alternate, pl.ty, pl.xcd, fuse, pl.deep, pl.ntl) + "|" +
                plan_kernel_name(fuse, pl.deep, pl.ty ? pl.ty : 8, pl.nt, arith, q.esize, pl.variant == 0 && q.vec_ok) + "|" + std::to_string(pl.deep ? deep_depth(pl.deep) : fuse) + "\n";
    }
    if ((int)text.size() + 1 > cap) return fail(LBM_ERR_ARG, "buffer too small (%zu bytes needed)", text.size() + 1);
    memcpy(out, text.c_str(), text.size() + 1);
    return LBM_OK;
}

/* TEST HOOK (device 0): strict_div2 (the shared-reciprocal form of the collision's two divisions by rho, lbm_kernels.hpp) beside
 * the compiler's IEEE divisions on the same operands: q1,q2 = strict_div2(a1,a2,b); r1,r2 = a1/b, a2/b. Host arrays of n doubles. */
int lbm_debug_strict_div2(const double* a1, const double* a2, const double* b, int n, double* q1, double* q2, double* r1, double* r2) {
    if (!a1 || !a2 || !b || !q1 || !q2 || !r1 || !r2 || n < 1) return fail(LBM_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(0));
    double* d = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    HIPCHK(hipMalloc(&d, 7 * nb));
    int rc = LBM_OK;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == LBM_OK) rc = fail(LBM_ERR_HIP, "%s", hipGetErrorString(e)); };
    chk(hipMemcpy(d, a1, nb, hipMemcpyHostToDevice)); chk(hipMemcpy(d + n, a2, nb, hipMemcpyHostToDevice)); chk(hipMemcpy(d + 2 * (size_t)n, b, nb, hipMemcpyHostToDevice));
    if (rc == LBM_OK) {
        hipLaunchKernelGGL(k_debug_strict_div2<0>, dim3((n + 255) / 256), dim3(256), 0, nullptr, d, d + n, d + 2 * (size_t)n, n, d + 3 * (size_t)n, d + 4 * (size_t)n, d + 5 * (size_t)n, d + 6 * (size_t)n);
        chk(hipGetLastError());
        chk(hipDeviceSynchronize());
        chk(hipMemcpy(q1, d + 3 * (size_t)n, nb, hipMemcpyDeviceToHost)); chk(hipMemcpy(q2, d + 4 * (size_t)n, nb, hipMemcpyDeviceToHost));
        chk(hipMemcpy(r1, d + 5 * (size_t)n, nb, hipMemcpyDeviceToHost)); chk(hipMemcpy(r2, d + 6 * (size_t)n, nb, hipMemcpyDeviceToHost));
    }
    (void)hipFree(d);
    return rc;
}

/* TEST HOOK (no device needed): the decision tune_strip_schedule takes from per-rank pins. ranks x {tune-able, overlap_pinned,
 * overlap, deep_pinned, deep_halo}; returns LBM_OK and the agreed {go, overlap_pinned, overlap, deep_pinned, deep_halo} or LBM_ERR_ARG. */
int lbm_debug_strip_pins(const int* per_rank5, int nranks, int* agreed5) {
    if (!per_rank5 || !agreed5 || nranks < 1) return fail(LBM_ERR_ARG, "null argument");
    double m[5] = {1e30, 1e30, 1e30, 1e30, 1e30};
    for (int r = 0; r < nranks; ++r) {
        double v[5];
        const int* q = per_rank5 + 5 * r;
        strip_pins_pack(q[0] != 0, q[1] != 0, q[2], q[3] != 0, q[4], v);
        for (int k = 0; k < 5; ++k) m[k] = std::min(m[k], v[k]);        // what allreduce_doubles(..., MIN) returns on every rank
    }
    agreed5[2] = agreed5[4] = -1;
    if (!strip_pins_agree(m, &agreed5[0], &agreed5[1], &agreed5[2], &agreed5[3], &agreed5[4]))
        return fail(LBM_ERR_ARG, "the ranks pin different strip schedules");
    return LBM_OK;
}

const char* lbm_plan(const lbm_ctx* c) { return c ? c->plan_desc : ""; }
const char* lbm_plan_options(const lbm_ctx* c) { return c ? c->plan_opts : ""; }

#ifndef LBM_BUILD_ID_STR
#define LBM_BUILD_ID_STR "unversioned-----"
#endif
const char* lbm_build_id(void) {
    static const char tag[] = "LBM_BUILD_ID=" LBM_BUILD_ID_STR;   // the marker build.py looks for in the file
    return tag + 13;
}

}  // extern "C"
