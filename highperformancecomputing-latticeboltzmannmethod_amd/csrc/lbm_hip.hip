// csrc/lbm_hip.hip — host side of liblbm_hip.so: the C-ABI of include/lbm_hip.h over the gfx950 kernels in
// lbm_kernels.hpp. Plain HIP runtime + RCCL; no torch types, no CPU fallback.
#include "lbm_kernels.hpp"
#include "lbm_col_api.hpp"
#include "lbm_plan.hpp"
#include "../../include/lbm_hip.h"

#include <rccl/rccl.h>

#include <climits>
#include <cstdint>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

using namespace lbmk;

#include "lbm_ctx.hpp"

namespace {
#include "lbm_launch.inc.hpp"
#include "lbm_strips.inc.hpp"
#include "lbm_tune.inc.hpp"
#include "lbm_steps.inc.hpp"
#include "lbm_choreo.inc.hpp"
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
    HIPTRY(hipMalloc(&c->d_halo, 4 * HR1 * Q * sizeof(double) * (size_t)c->nx));
    HIPTRY(hipMalloc(&c->d_red, 64 * sizeof(double)));
#undef HIPTRY
    *out = c;
    return LBM_OK;
}

void lbm_destroy(lbm_ctx* c) {
    if (!c) return;
    lbm_trace("destroy", "ctx %p begin", (void*)c);
    (void)hipSetDevice(c->device);
    // A group that ran into LBM_ERR_TIMEOUT: its threads may still be inside a runtime call on this context's streams. Wait for them
    // (bounded); if they never come back, or the streams never drain, the context is LEAKED — freeing it under a thread that may still
    // wake up, or blocking for ever in a drain, would turn a reported error into a crash or a hang.
    bool leak = c->pool && c->pool->S->broken.load() && !c->pool->quiesce(std::max(5000L, c->pool->S->timeout_ms));
    if (!leak && c->stream && wait_stream(c, c->stream, "compute stream (lbm_destroy)") == LBM_ERR_TIMEOUT) leak = true;
    if (!leak && c->comm_stream && wait_stream(c, c->comm_stream, "exchange stream (lbm_destroy)") == LBM_ERR_TIMEOUT) leak = true;
    for (lbm_ctx* nb : {c->nb_south, c->nb_north}) {  // a destroyed member leaves its group
        if (!nb) continue;
        // the neighbour's exchange stream may still hold a pull (hipMemcpyPeerAsync) that READS this member's edge rows, and
        // freeing a buffer only waits for work of this member's own device: drain the neighbour's streams first
        (void)hipSetDevice(nb->device);
        if (!leak && nb->comm_stream && wait_stream(nb, nb->comm_stream, "a neighbour's exchange stream (lbm_destroy)") == LBM_ERR_TIMEOUT) leak = true;
        if (!leak && nb->stream && wait_stream(nb, nb->stream, "a neighbour's compute stream (lbm_destroy)") == LBM_ERR_TIMEOUT) leak = true;
        if (nb->nb_south == c) nb->nb_south = nullptr;
        if (nb->nb_north == c) nb->nb_north = nullptr;
    }
    (void)hipSetDevice(c->device);
    if (leak) {
        fprintf(stderr, "lbm_destroy: context %p left allocated (%s)\n", (void*)c, g_err);
        lbm_trace("destroy", "ctx %p LEAKED: %s", (void*)c, g_err);
        c->nb_south = c->nb_north = nullptr;
        return;
    }
    // (a graph that captured RCCL operations holds the communicator: ncclCommDestroy waits for it to go away first)
    if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    if (c->comm) { lbm_trace("destroy", "ctx %p ncclCommDestroy", (void*)c); ncclCommDestroy(c->comm); }
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
    lbm_trace("destroy", "ctx %p end", (void*)c);
    delete c;
}

int lbm_initialise(lbm_ctx* c, int* solid_count_out) {
    if (!c) return fail(LBM_ERR_ARG, "null context");
    lbm_trace("initialise", "ctx %p %dx%d rows %d..%d begin", (void*)c, c->nx, c->p.ny, c->p.y_start, c->p.y_start + c->nyl);
    HIPCHK(hipSetDevice(c->device));
    { int wr = wait_stream(c, c->stream, "compute stream (lbm_initialise)"); if (wr) return wr; }        // (re-)initialisation starts from quiet streams
    { int wr = wait_stream(c, c->comm_stream, "exchange stream (lbm_initialise)"); if (wr) return wr; }
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
    lbm_trace("initialise", "ctx %p end: %s", (void*)c, c->plan_desc);
    return LBM_OK;
}

int lbm_step(lbm_ctx* c, int nsteps, int output_frequency) {
    if (!c || !c->initialised) return fail(LBM_ERR_ARG, "context not initialised");
    if (nsteps < 0) return fail(LBM_ERR_ARG, "nsteps < 0");
    if (c->group_n > 1) return fail(LBM_ERR_ARG, "this context is a member of a group: use lbm_group_step");
    HIPCHK(hipSetDevice(c->device));
    lbm_trace("step", "ctx %p t=%d +%d begin", (void*)c, c->steps_done, nsteps);
    const int rc = DISPATCH(c, do_steps<double>(&c, 1, nsteps, output_frequency), do_steps<float>(&c, 1, nsteps, output_frequency));
    lbm_trace("step", "ctx %p t=%d issued rc=%d", (void*)c, c->steps_done, rc);
    return rc;
}

int lbm_sync(lbm_ctx* c) {
    if (!c) return fail(LBM_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    // bounded (a poll) where the caller has set "wait_timeout_ms"; else the blocking call, which a watchdog names if it stalls (lbm_ctx.hpp)
    auto wait = [&](hipStream_t s, const char* what) { return c->wait_timeout_ms > 0 ? wait_stream(c, s, what) : sync_stream_blocking(c, s, what); };
    int rc = wait(c->stream, "compute stream");
    if (!rc) rc = wait(c->comm_stream, "exchange stream");
    return rc;
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
    lbm_trace("comm_init", "ctx %p rank %d of %d begin", (void*)c, rank, nranks);
    NCCLCHK(ncclCommInitRank(&c->comm, nranks, id, rank));
    lbm_trace("comm_init", "ctx %p end", (void*)c);
    c->rank = rank;
    c->nranks = nranks;
    if (nranks > 1 && c->nyl < 2 * HR1) return fail(LBM_ERR_ARG, "a strip with neighbours needs at least %d rows", 2 * HR1);
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
    static thread_local char out[1024];
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
        if (n > 1 && c->nyl < 2 * HR1) return fail(LBM_ERR_ARG, "a strip with neighbours needs at least %d rows", 2 * HR1);
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
    lbm_trace("group_step", "%d strips t=%d +%d begin", n, cs[0]->steps_done, nsteps);
    rc = DISPATCH(cs[0], do_steps<double>(cs, n, nsteps, output_frequency), do_steps<float>(cs, n, nsteps, output_frequency));
    lbm_trace("group_step", "%d strips t=%d issued rc=%d", n, cs[0]->steps_done, rc);
    return rc;
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
    if (c->initialised && (k == "layout" || k == "nt" || k == "ntl" || k == "tune" || k == "pair" || k == "fuse" || k == "pair_ty" || k == "loopback" || k == "deep" || k == "arith"))
        return fail(LBM_ERR_ARG, "option %s must be set before lbm_initialise", key);
    if (k == "timing") c->timing = (int)value;
    else if (k == "alternate") c->alternate = (int)value;
    else if (k == "layout") c->layout = (int)value ? 1 : 0;
    else if (k == "nt") c->use_nt = (int)value ? 1 : 0;
    else if (k == "ntl") c->use_ntl = (int)value ? 1 : 0;
    else if (k == "fuse") { if (value < 1 || value > 4) return fail(LBM_ERR_ARG, "fuse must be 1, 2, 3 or 4 (4: tile kernel, no strip faces)"); c->fuse = (int)value; c->deep = 0; }
    else if (k == "deep") {     // k_stepd_tile: 1: 6 iterations on 64x16 tiles, 2: 7 on 64x16, 3: 8 on 32x32 (1024 threads);
                                // k_stepc_col (registers): 6 / 7 / 9: 5 / 6 / 7 iterations on 64x32 regions; 8 (fp32): 7 iterations on 64x48 regions.
        if (!deep_valid((int)value)) return fail(LBM_ERR_ARG, "deep must be 0..3 or 6..9 (4 / 5, round 2's 32x16 LDS tiles, are retired)");
        if (deep_is_tall((int)value) && c->esize != 4) return fail(LBM_ERR_ARG, "deep 8 (64x48 regions in registers, twelve waves x four rows) exists in fp32 only");
        c->deep = (int)value;
        if (c->deep) c->fuse = deep_depth(c->deep);
    }
    else if (k == "pair") c->fuse = (int)value ? 2 : 1;
    else if (k == "trailing_pair") c->trailing_pair = (int)value ? 1 : 0;
    else if (k == "xcd") c->xcd = (int)value ? 1 : 0;
    else if (k == "arith") { if (value != 0 && value != 1) return fail(LBM_ERR_ARG, "arith must be 0 (strict) or 1 (contracted)"); c->arith = (int)value; }
    else if (k == "deep_halo") { if (value < 0 || value > 2) return fail(LBM_ERR_ARG, "deep_halo must be 0, 1 or 2"); c->deep_halo = (int)value; c->deep_pinned = true; }
    else if (k == "halo_trim") { if (value != 0 && value != 1) return fail(LBM_ERR_ARG, "halo_trim must be 0 or 1"); c->halo_trim = (int)value; c->trim_pinned = true; }
    else if (k == "skip_exchange") c->skip_exchange = (int)value ? 1 : 0;
    else if (k == "group_threads") c->group_threads = (int)value ? 1 : 0;
    else if (k == "wait_timeout_ms") { if (value < 0) return fail(LBM_ERR_ARG, "wait_timeout_ms must be >= 0 (0: LBM_WAIT_TIMEOUT_MS or five minutes)"); c->wait_timeout_ms = value; }
    else if (k == "debug_fault_launch") c->debug_fault_launch = (int)value;      // TEST ONLY: see lbm_ctx
    else if (k == "debug_fault_point") { if (value < 0 || value > 2) return fail(LBM_ERR_ARG, "debug_fault_point must be 0, 1 or 2"); c->debug_fault_point = (int)value; }
    else if (k == "debug_fault_stall_ms") c->debug_fault_stall_ms = (int)value;
    else if (k == "debug_old_edge_band") c->debug_old_edge_band = (int)value ? 1 : 0;      // TEST ONLY: see lbm_ctx
    else if (k == "debug_skip_pull_wait") c->debug_skip_pull_wait = (int)value ? 1 : 0;    // TEST ONLY: see lbm_ctx
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
    // the kernel plan_launch really issues: a strip whose halos are staged through the host carries LBM_HALO_ROWS ghost
    // rows per exchange, so a deeper plan (seven / eight iterations per launch) falls back to the three-iteration tile kernel there
    const bool phys_face = face_south(c) || face_north(c);
    const bool usable = c->deep && (!phys_face || deep_depth(c->deep) <= (device_transport(c) ? GR : HR1));
    const int deep = usable ? c->deep : 0, fuse = usable || !c->deep ? c->fuse : std::min(c->fuse, 3);
    snprintf(name, sizeof(name), "%s", plan_kernel_name(fuse, deep, c->pair_ty, c->use_nt, c->arith, (int)c->esize).c_str());
    return name;
}

/* TEST HOOK (no device needed): the candidates lbm_initialise would time for a whole-domain context of this grid, one per line:
 * "name|lbm_set_option pairs|dominant kernel|iterations per launch". */
int lbm_debug_plan_candidates(int nx, int ny, int precision, int arith, int num_cus, char* out, int cap) {
    if (!out || cap < 1 || nx < 1 || ny < 1) return fail(LBM_ERR_ARG, "bad argument");
    PlanQuery q;
    q.nx = nx; q.nyl = ny; q.ny_glob = ny; q.esize = precision == LBM_PRECISION_F32 ? 4 : 8; q.num_cus = num_cus > 0 ? num_cus : 256;
    const Plan none{};
    std::string text;
    for (const Plan& pl : plan_candidates(q, none)) {
        const int fuse = pl.fuse > 0 ? pl.fuse : 1;
        text += pl.name + "|" + plan_option_string(pl.layout, pl.nt, pl.alternate, pl.ty, pl.xcd, fuse, pl.deep, pl.ntl) + "|" +
                plan_kernel_name(fuse, pl.deep, pl.ty ? pl.ty : 8, pl.nt, arith, q.esize) + "|" + std::to_string(pl.deep ? deep_depth(pl.deep) : fuse) + "\n";
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
int lbm_debug_strip_pins(const int* per_rank7, int nranks, int* agreed7) {
    if (!per_rank7 || !agreed7 || nranks < 1) return fail(LBM_ERR_ARG, "null argument");
    double m[7] = {1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30};
    for (int r = 0; r < nranks; ++r) {
        double v[7];
        const int* q = per_rank7 + 7 * r;
        strip_pins_pack(q[0] != 0, q[1] != 0, q[2], q[3] != 0, q[4], q[5] != 0, q[6], v);
        for (int k = 0; k < 7; ++k) m[k] = std::min(m[k], v[k]);        // what allreduce_doubles(..., MIN) returns on every rank
    }
    agreed7[2] = agreed7[4] = agreed7[6] = -1;
    if (!strip_pins_agree(m, &agreed7[0], &agreed7[1], &agreed7[2], &agreed7[3], &agreed7[4], &agreed7[5], &agreed7[6]))
        return fail(LBM_ERR_ARG, "the ranks pin different strip schedules");
    return LBM_OK;
}

/* TEST HOOK (no device needed): the runs of the halo message of a face of `hr` rows (csrc/lbm_strips.inc.hpp face_runs): up to five
 * {first sub-row, sub-rows} pairs relative to the block's first sub-row; south_block: the block lies below the strip it borders. */
int lbm_debug_face_runs(int hr, int trim, int south_block, int* runs10) {
    if (!runs10 || hr < 1) return fail(LBM_ERR_ARG, "bad argument");
    const FaceRuns r = face_runs(hr, trim != 0, south_block != 0);
    for (int k = 0; k < r.n; ++k) { runs10[2 * k] = r.off[k]; runs10[2 * k + 1] = r.cnt[k]; }
    return r.n;
}

/* TEST HOOK (no device needed): the host threads of a group (GroupPool, csrc/lbm_ctx.hpp) on a dummy job of `rounds` rounds with one
 * rendezvous each. Strip `fail_strip` reports an injected error in round `fail_round`; strip `stall_strip` sleeps `stall_ms` before
 * the rendezvous of round `stall_round` (-1: nobody). The job runs `repeat` times on the same pool. Returns what the LAST run returned
 * (LBM_OK / the injected LBM_ERR_HIP / LBM_ERR_TIMEOUT / "timed out earlier"); *rendezvous_out = rendezvous strip 0 passed in the last run. */
int lbm_debug_group_pool(int n, int rounds, int fail_strip, int fail_round, int stall_strip, int stall_round, int stall_ms, long timeout_ms, int repeat,
                         int* rendezvous_out) {
    if (n < 2 || n > 64 || rounds < 1 || repeat < 1) return fail(LBM_ERR_ARG, "bad argument");
    GroupPool pool(n);
    auto passed = std::make_shared<std::atomic<int>>(0);
    int rc = LBM_OK;
    for (int rep = 0; rep < repeat; ++rep) {
        passed->store(0);
        const bool last = rep + 1 == repeat;
        rc = pool.run([=](GroupPool::State& P, int i) {
            for (int r = 0; r < rounds; ++r) {
                if (last && i == fail_strip && r == fail_round) P.report(fail(LBM_ERR_HIP, "injected fault: strip %d, round %d", i, r), g_err);
                if (last && i == stall_strip && r == stall_round) std::this_thread::sleep_for(std::chrono::milliseconds(stall_ms));
                if (P.arrive(i, r, 1)) return;
                if (i == 0) passed->fetch_add(1);
            }
        }, timeout_ms);
    }
    if (rendezvous_out) *rendezvous_out = passed->load();
    return rc;        // (~GroupPool: waits, bounded, for a straggler and joins; a thread that never comes back is detached)
}

/* TEST HOOK (no device needed): dry run of the launch choreography of a strip run and its check (csrc/lbm_choreo.inc.hpp). */
int lbm_debug_choreography(int nx, int ny, const int* bounds2, int nstrips, int precision, int transport, const char* options,
                           const int* calls2, int ncalls, int dump, char* out, int cap) {
    if (!bounds2 || nstrips < 1 || !calls2 || ncalls < 1 || (out && cap < 1) || transport < 0 || transport > 3) return fail(LBM_ERR_ARG, "bad argument");
    if (transport >= 2 && nstrips != 1) return fail(LBM_ERR_ARG, "transports 2 (one rank of a multi-process run) and 3 (loopback) describe ONE strip");
    std::vector<lbm_ctx*> cs;
    Choreo rec;
    auto cleanup = [&]() { for (lbm_ctx* c : cs) delete c; };
    for (int k = 0; k < nstrips; ++k) {
        lbm_ctx* c = choreo_fake_ctx(nx, ny, bounds2[2 * k], bounds2[2 * k + 1], precision, k);
        cs.push_back(c);
        if (c->p.y_start < 0 || c->nyl < 1 || c->p.y_start + c->nyl > ny) { cleanup(); return fail(LBM_ERR_ARG, "strip %d outside the lattice", k); }
        for (const char* q = options ? options : ""; *q;) {        // "key=value key=value"
            while (*q == ' ') ++q;
            const char* eq = strchr(q, '=');
            if (!eq) break;
            const std::string key(q, eq);
            char* end = nullptr;
            const long v = strtol(eq + 1, &end, 10);
            const int rc = lbm_set_option(c, key.c_str(), v);
            if (rc) { cleanup(); return rc; }
            q = end;
        }
        c->layout = 1; configure_layout(c, 1);        // strips: row-interleaved
        if (c->deep) c->fuse = deep_depth(c->deep);
        if (transport == 2) {      // rank and size as the strip's faces say
            const bool s_face = c->p.y_start > 0, n_face = c->p.y_start + c->nyl < ny;
            c->comm = (ncclComm_t)(uintptr_t)0x1;
            c->nranks = 1 + (s_face ? 1 : 0) + (n_face ? 1 : 0);
            c->rank = s_face ? 1 : 0;
        } else if (transport == 3) c->loopback = 1;
        c->initialised = true;
        c->cur = 1;
    }
    if (transport < 2) {
        for (int k = 0; k < nstrips; ++k) {
            cs[(size_t)k]->nb_south = k > 0 ? cs[(size_t)k - 1] : nullptr;
            cs[(size_t)k]->nb_north = k + 1 < nstrips ? cs[(size_t)k + 1] : nullptr;
            cs[(size_t)k]->group_transport = transport; cs[(size_t)k]->group_n = nstrips; cs[(size_t)k]->group_threads = 0;
            const int expect = k == 0 ? 0 : cs[(size_t)k - 1]->p.y_start + cs[(size_t)k - 1]->nyl;
            if (cs[(size_t)k]->p.y_start != expect || (k + 1 == nstrips && expect + cs[(size_t)k]->nyl != ny)) { cleanup(); return fail(LBM_ERR_ARG, "the strips must cover the lattice bottom to top"); }
        }
    }
    ChoreoChecker chk;
    chk.init(cs.data(), nstrips);       // (before `rec` is set: the initial state is what lbm_initialise leaves)
    for (lbm_ctx* c : cs) c->rec = &rec;
    int rc = LBM_OK;
    for (int k = 0; k < ncalls && !rc; ++k)
        rc = DISPATCH(cs[0], do_steps<double>(cs.data(), nstrips, calls2[2 * k], calls2[2 * k + 1]), do_steps<float>(cs.data(), nstrips, calls2[2 * k], calls2[2 * k + 1]));
    if (rc) { cleanup(); return rc; }
    chk.run(rec.ops);
    std::string text;
    for (const ChoreoViolation& v : chk.bad) {
        char b[96];
        if (v.kind == 0) {
            snprintf(b, sizeof(b), "RACE strip %d buffer %d row %d:\n   ", v.strip, v.buf, v.row);
            text += b + choreo_op_text(rec.ops[(size_t)v.op_a], v.op_a) + "\n   " + choreo_op_text(rec.ops[(size_t)v.op_b], v.op_b) + "\n";
        } else {
            snprintf(b, sizeof(b), "STALE strip %d buffer %d row %d holds iteration %d, wanted %d:\n   ", v.strip, v.buf, v.row, v.have == INT_MIN ? -1 : v.have, v.want);
            text += b + choreo_op_text(rec.ops[(size_t)v.op_a], v.op_a) + "\n";
        }
        if (text.size() > 6000) break;
    }
    if (dump) for (int i = 0; i < (int)rec.ops.size(); ++i) text += choreo_op_text(rec.ops[(size_t)i], i) + "\n";
    if (out) { const size_t m = std::min(text.size(), (size_t)cap - 1); memcpy(out, text.data(), m); out[m] = 0; }
    const int nbad = (int)chk.bad.size();
    cleanup();
    return nbad;
}

/* TEST HOOK (no device needed): every rank of a multi-process strip run issues its launch groups DRY (as lbm_debug_choreography does for one
 * rank) and the transcripts of exchange_rccl's posting loops are paired: see include/lbm_hip.h. */
int lbm_debug_p2p_matching(int nx, int ny, const int* bounds2, int nranks, int precision, const char* options, const char* options_rank1,
                           const int* calls2, int ncalls, char* out, int cap) {
    if (!bounds2 || nranks < 2 || !calls2 || ncalls < 1 || (out && cap < 1)) return fail(LBM_ERR_ARG, "bad argument");
    std::vector<lbm_ctx*> cs;
    std::vector<Choreo> rec((size_t)nranks);
    auto cleanup = [&]() { for (lbm_ctx* c : cs) delete c; };
    auto apply = [&](lbm_ctx* c, const char* opts) -> int {
        for (const char* q = opts ? opts : ""; *q;) {
            while (*q == ' ') ++q;
            const char* eq = strchr(q, '=');
            if (!eq) break;
            const std::string key(q, eq);
            char* end = nullptr;
            const long v = strtol(eq + 1, &end, 10);
            const int rc = lbm_set_option(c, key.c_str(), v);
            if (rc) return rc;
            q = end;
        }
        return LBM_OK;
    };
    for (int k = 0; k < nranks; ++k) {
        lbm_ctx* c = choreo_fake_ctx(nx, ny, bounds2[2 * k], bounds2[2 * k + 1], precision, 0);
        cs.push_back(c);
        const int expect = k == 0 ? 0 : cs[(size_t)k - 1]->p.y_start + cs[(size_t)k - 1]->nyl;
        if (c->nyl < 1 || c->p.y_start != expect || (k + 1 == nranks && expect + c->nyl != ny)) { cleanup(); return fail(LBM_ERR_ARG, "the strips must cover the lattice bottom to top"); }
        int rc = apply(c, options);
        if (!rc && k == 1) rc = apply(c, options_rank1);
        if (rc) { cleanup(); return rc; }
        c->layout = 1; configure_layout(c, 1);
        if (c->deep) c->fuse = deep_depth(c->deep);
        c->comm = (ncclComm_t)(uintptr_t)0x1; c->rank = k; c->nranks = nranks;
        c->initialised = true; c->cur = 1;
        c->rec = &rec[(size_t)k];
        for (int q = 0; q < ncalls && !rc; ++q)
            rc = DISPATCH(c, do_steps<double>(&c, 1, calls2[2 * q], calls2[2 * q + 1]), do_steps<float>(&c, 1, calls2[2 * q], calls2[2 * q + 1]));
        if (rc) { cleanup(); return rc; }
    }
    std::string text;
    int bad = 0;
    auto pick = [&](int r, int kind, int peer) { std::vector<ChoreoP2P> v; for (const ChoreoP2P& p : rec[(size_t)r].p2p) if (p.kind == kind && p.peer == peer) v.push_back(p); return v; };
    auto groups = [&](int r) { int n = 0; for (const ChoreoP2P& p : rec[(size_t)r].p2p) n += p.kind == 2; return n; };
    auto pair_up = [&](int from, int to, long from_block, long to_block, const char* what) {
        const std::vector<ChoreoP2P> S = pick(from, 0, to), R = pick(to, 1, from);
        char b[256];
        if (S.size() != R.size()) { ++bad; snprintf(b, sizeof(b), "%s: rank %d posts %zu sends to rank %d, which posts %zu receives from it\n", what, from, S.size(), to, R.size()); text += b; return; }
        for (size_t k = 0; k < S.size(); ++k)
            if (S[k].cnt != R[k].cnt || S[k].off - from_block != R[k].off - to_block) {
                ++bad;
                if (text.size() < 3000) { snprintf(b, sizeof(b), "%s: message %zu: rank %d sends %ld elements from block offset %ld, rank %d receives %ld at block offset %ld\n", what, k, from,
                                                   S[k].cnt, S[k].off - from_block, to, R[k].cnt, R[k].off - to_block); text += b; }
            }
    };
    for (int r = 0; r + 1 < nranks; ++r) {
        const FaceSpans lo = face_spans(cs[(size_t)r]), hi = face_spans(cs[(size_t)r + 1]);
        pair_up(r, r + 1, lo.top_rows, hi.ghost_s, "northbound");
        pair_up(r + 1, r, hi.bot_rows, lo.ghost_n, "southbound");
        if (groups(r) != groups(r + 1)) { ++bad; char b[160]; snprintf(b, sizeof(b), "ranks %d and %d issue %d and %d exchanges\n", r, r + 1, groups(r), groups(r + 1)); text += b; }
    }
    {
        char b[160];
        snprintf(b, sizeof(b), "%d exchanges per rank, %zu messages posted by rank 0\n", groups(0), rec[0].p2p.size() - (size_t)groups(0));
        text += b;
    }
    if (out) { const size_t m = std::min(text.size(), (size_t)cap - 1); memcpy(out, text.data(), m); out[m] = 0; }
    cleanup();
    return bad;
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
