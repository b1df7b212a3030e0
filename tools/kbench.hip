// tools/kbench.hip — kernel-variant microbenchmark (tuning tool, not part of the product or the parity path).
// Times candidate formulations of the step kernel on the same SoA layout, interleaved in ONE process
// (cdna_hip_programming.md §5.4 rule 24), and prints algorithmic GB/s = 144 B (or 72 B) x cells / time.
// Variants that win are moved into csrc/lbm_kernels.hpp and must then pass tests/ -m gpu.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -o tools/kbench tools/kbench.hip
#include "../highperformancecomputing-latticeboltzmannmethod_amd/csrc/lbm_kernels.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

using namespace lbmk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- candidates ------------------------------------------------------------------------------------------
// copy ceiling: 9 planes in, 9 planes out, same grid as the site kernel
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_copy9(const KArgs<T> a) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * VEC;
    const int y = blockIdx.y;
    if (x >= a.nx) return;
    const long c = (long)(y + GR) * a.pitch + a.xoff + x;
    typedef T VT __attribute__((ext_vector_type(VEC)));
    VT v[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) v[i] = *reinterpret_cast<const VT*>(a.src + (long)i * a.plane + c);
#pragma unroll
    for (int i = 0; i < Q; ++i) *reinterpret_cast<VT*>(a.dst + (long)i * a.plane + c) = v[i];
}

// site kernel with traversal reversal on odd steps / nontemporal accesses / y from a 1-D grid
template <typename T, bool REV, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_site_opt(const KArgs<T> a) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    int y = blockIdx.y;
    if (REV && (a.t & 1)) y = a.ny_loc - 1 - y;
    if (x >= a.nx) return;
    const int yg = a.y_start + y;
    const long c = (long)(y + GR) * a.pitch + a.xoff + x;
    T f[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const T* p = a.src + (long)i * a.plane + c - (long)cy(i) * a.pitch - cx(i);
        f[i] = NTL ? __builtin_nontemporal_load(p) : *p;
    }
    const bool solid = is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);
    T rho_bc, u_out;
    if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
    if (any_unstable(f)) atomicMin(a.unstable_t, a.t);
    if (solid) return;
    bgk_collide(f, a.tau_inv);
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        T* p = a.dst + (long)i * a.plane + c;
        if (NTS) __builtin_nontemporal_store(f[i], p); else *p = f[i];
    }
}

// V sites per thread along x; vector loads (unaligned by one element for the cx = +-1 planes)
template <typename T, int V, bool REV, bool NTS>
__global__ void __launch_bounds__(256) k_site_vec(const KArgs<T> a) {
    typedef T VA __attribute__((ext_vector_type(V)));                       // naturally aligned
    typedef T VU __attribute__((ext_vector_type(V), aligned(sizeof(T))));   // element aligned
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * V;
    int y = blockIdx.y;
    if (REV && (a.t & 1)) y = a.ny_loc - 1 - y;
    if (x0 >= a.nx) return;   // (nx is a multiple of V in this tool)
    const int yg = a.y_start + y;
    const long c = (long)(y + GR) * a.pitch + a.xoff + x0;
    VA fv[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const T* p = a.src + (long)i * a.plane + c - (long)cy(i) * a.pitch - cx(i);
        if (cx(i) == 0) fv[i] = *reinterpret_cast<const VA*>(p);
        else { VU u = *reinterpret_cast<const VU*>(p); for (int k = 0; k < V; ++k) fv[i][k] = u[k]; }
    }
    bool bad = false;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int x = x0 + k;
        T f[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) f[i] = fv[i][k];
        const bool solid = is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);
        T rho_bc, u_out;
        if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
        bad |= any_unstable(f);
        if (!solid) bgk_collide(f, a.tau_inv);
        else { for (int i = 0; i < Q; ++i) f[i] = wgt<T>(i); }   // solid cells hold w_i: rewriting them is a no-op
#pragma unroll
        for (int i = 0; i < Q; ++i) fv[i][k] = f[i];
    }
    if (bad) atomicMin(a.unstable_t, a.t);
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        VA* p = reinterpret_cast<VA*>(a.dst + (long)i * a.plane + c);
        if (NTS) __builtin_nontemporal_store(fv[i], p); else *p = fv[i];
    }
}



static bool g_leak = false;
static bool g_check = false;
static int g_trials = 1;
// ---- harness ---------------------------------------------------------------------------------------------
template <typename T>
struct Bench {
    int nx, ny, pitch, xoff;
    long plane;
    T *A, *B;
    int* d_unst;
    hipStream_t s;
    hipEvent_t e0, e1;
    double u_in;
    int t = 0;

    int rowstride;   // elements between consecutive rows of one plane
    size_t total;    // elements per buffer
    Bench(int nx_, int ny_, long plane_pad_elems, bool rowil, int pitch_pad) : nx(nx_), ny(ny_) {
        const int per128 = 128 / sizeof(T);
        xoff = per128;
        const int pitch0 = (xoff + nx + 1 + per128 - 1) / per128 * per128 + pitch_pad;
        if (rowil) { plane = pitch0; rowstride = Q * pitch0 + (int)plane_pad_elems; total = (size_t)rowstride * (ny + 2 * GR); }
        else {
            rowstride = pitch0;
            long raw = (long)pitch0 * (ny + 2 * GR);
            if (plane_pad_elems < 0) {   // rule: plane stride = k*64 KiB + (-pad) bytes
                const long w = 65536 / sizeof(T);
                plane = (raw + w - 1) / w * w + (-plane_pad_elems) / (long)sizeof(T);
            } else plane = raw + plane_pad_elems;
            total = (size_t)Q * plane;
        }
        pitch = rowstride;
        // (The round-1 "--single --boff N" experiment — both buffers carved out of ONE allocation at a byte offset — is gone:
        // every run with N >= 1.5 MiB died with a GPU memory access fault about 264 MiB into the first buffer although the
        // pointer arithmetic stays inside the allocation; see profiles/r02/README.md. Two allocations, as the library does.)
        CK(hipMalloc(&A, (total + 64) * sizeof(T)));
        CK(hipMalloc(&B, (total + 64) * sizeof(T)));
        printf("A=%p B=%p\n", (void*)A, (void*)B);
        fflush(stdout);
        CK(hipMalloc(&d_unst, sizeof(int)));
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        u_in = 200.0 * (0.1 / 3.0) / (0.1 * ny);
    }
    ~Bench() { if (!g_leak) { hipFree(A); hipFree(B); } hipFree(d_unst); hipStreamDestroy(s); hipEventDestroy(e0); hipEventDestroy(e1); }

    KArgs<T> args(bool flip) {
        KArgs<T> a{};
        a.y_lo = 0; a.y_cnt = ny; a.reverse = 0;
        a.src = flip ? B : A; a.dst = flip ? A : B;
        a.plane = plane; a.pitch = pitch; a.xoff = xoff; a.nx = nx; a.ny_loc = ny; a.ny_glob = ny; a.y_start = 0;
        a.cyl_x = (int)(0.2 * nx); a.cyl_y = (int)(0.5 * ny);
        const int r = (int)(0.05 * ny); a.cyl_r2 = (double)(r * r);
        a.tau_inv = (T)(1.0 / 0.6); a.u_in = (T)u_in; a.unstable_t = d_unst; a.t = t;
        return a;
    }
    void init() {
        InitArgs<T> ia;
        ia.a = A; ia.b = B; ia.plane = plane; ia.pitch = pitch; ia.xoff = xoff; ia.nx = nx; ia.ny_loc = ny;
        ia.ny_glob = ny; ia.y_start = 0; ia.cyl_x = (int)(0.2 * nx); ia.cyl_y = (int)(0.5 * ny);
        const int r = (int)(0.05 * ny); ia.cyl_r2 = (double)(r * r);
        const double ux = u_in, usq = ux * ux, t3 = 1.5 * usq;
        ia.feq_in[0] = (T)(wgt<double>(0) * (1.0 - 1.5 * usq));
        for (int i = 1; i < Q; ++i) { const double cu = cx(i) * ux; ia.feq_in[i] = (T)(wgt<double>(i) * (((1.0 + 3.0 * cu) - t3) + 4.5 * cu * cu)); }
        ia.solid_count = d_unst;
        hipLaunchKernelGGL((k_init<T>), dim3((nx + 2 + 255) / 256, ny + 2 * GR), dim3(256), 0, s, ia);
        const int big = 0x7fffffff;
        CK(hipMemcpyAsync(d_unst, &big, sizeof(int), hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        t = 0;
    }
    // run `n` launches of `launch(args)`, ping-ponging A/B; returns mean ms per launch
    double run(const std::function<void(const KArgs<T>&)>& launch, int n) {
        CK(hipEventRecord(e0, s));
        for (int k = 0; k < n; ++k) { launch(args(t & 1)); ++t; }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / n;
    }
};

static std::string g_filter;
template <typename T>
void suite(int nx, int ny, long pad, int reps, int rounds, bool rowil, int pitch_pad) {
    Bench<T> b(nx, ny, pad, rowil, pitch_pad);
    const double bytes = (double)nx * ny * 2 * Q * sizeof(T);
    struct Var { std::string name; std::function<void(const KArgs<T>&)> fn; std::vector<double> ms; };
    std::vector<Var> vars;
    hipStream_t s = b.s;
    auto grid1 = dim3((nx + 255) / 256, ny);
    constexpr int V16 = 16 / sizeof(T);
    auto gridv = [&](int v) { return dim3((nx / v + 255) / 256, ny); };
    vars.push_back({"site (product baseline)", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step_site<T, MODE_STEP>), grid1, dim3(256), 0, s, a); }, {}});
    vars.push_back({"site rev", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_site_opt<T, true, false, false>), grid1, dim3(256), 0, s, a); }, {}});
    vars.push_back({"site nt-store", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_site_opt<T, false, false, true>), grid1, dim3(256), 0, s, a); }, {}});
    vars.push_back({"site nt-load+store", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_site_opt<T, false, true, true>), grid1, dim3(256), 0, s, a); }, {}});
    vars.push_back({"site rev nt-store", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_site_opt<T, true, false, true>), grid1, dim3(256), 0, s, a); }, {}});
    vars.push_back({"vec16B", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_site_vec<T, V16, false, false>), gridv(V16), dim3(256), 0, s, a); }, {}});
    vars.push_back({"vec16B rev", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_site_vec<T, V16, true, false>), gridv(V16), dim3(256), 0, s, a); }, {}});
    vars.push_back({"vec16B nt-store", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_site_vec<T, V16, false, true>), gridv(V16), dim3(256), 0, s, a); }, {}});
    vars.push_back({"vec16B rev nt-store", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_site_vec<T, V16, true, true>), gridv(V16), dim3(256), 0, s, a); }, {}});
    vars.push_back({"copy9 1/thread", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_copy9<T, 1>), grid1, dim3(256), 0, s, a); }, {}});
    vars.push_back({"copy9 16B/thread", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_copy9<T, V16>), gridv(V16), dim3(256), 0, s, a); }, {}});
    if (!g_filter.empty()) {
        std::vector<Var> keep;
        for (auto& v : vars) if (("," + g_filter + ",").find("," + v.name + ",") != std::string::npos) keep.push_back(v);
        vars = keep;
    }
    {
        K2Extra<T> ex;
        {
            T hv[Q];
            const double ux = b.u_in, usq = ux * ux, t3 = 1.5 * usq;
            hv[0] = (T)(wgt<double>(0) * (1.0 - 1.5 * usq));
            for (int i = 1; i < Q; ++i) { const double cu = cx(i) * ux; hv[i] = (T)(wgt<double>(i) * (((1.0 + 3.0 * cu) - t3) + 4.5 * cu * cu)); }
            T* dv; CK(hipMalloc(&dv, sizeof(hv))); CK(hipMemcpy(dv, hv, sizeof(hv), hipMemcpyHostToDevice));
            ex.feq_in = dv; ex.small = 1;
        }
        auto g2 = [&](int ty) { return dim3(nx / 64, (ny + ty - 1) / ty); };
        vars.push_back({"step2 TY8 512t", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 8, 512, false>), g2(8), dim3(512), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY8 704t", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 8, 704, false>), g2(8), dim3(704), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY8 512t nt", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 8, 512, true>), g2(8), dim3(512), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY4 384t", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 4, 384, false>), g2(4), dim3(384), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY12 768t", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 12, 768, false>), g2(12), dim3(768), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY16 1024t", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 16, 1024, false>), g2(16), dim3(1024), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY12 768t nt", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 12, 768, true>), g2(12), dim3(768), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY12 768t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 12, 768, true, true>), g2(12), dim3(768), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY8 512t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 8, 512, true, true>), g2(8), dim3(512), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY14 896t nt", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 14, 896, true>), g2(14), dim3(896), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY14 896t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 14, 896, true, true>), g2(14), dim3(896), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY14 448t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 14, 448, true, true>), g2(14), dim3(448), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY12 384t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 12, 384, true, true>), g2(12), dim3(384), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY12 768t nt", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 12, 768, true, false>), g2(12), dim3(768), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY12 768t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 12, 768, true, true>), g2(12), dim3(768), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY12 576t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 12, 576, true, true>), g2(12), dim3(576), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY12 640t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 12, 640, true, true>), g2(12), dim3(640), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY12 704t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 12, 704, true, true>), g2(12), dim3(704), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY12 832t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 12, 832, true, true>), g2(12), dim3(832), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY12 896t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 12, 896, true, true>), g2(12), dim3(896), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY12 960t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 12, 960, true, true>), g2(12), dim3(960), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY12 1024t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 12, 1024, true, true>), g2(12), dim3(1024), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY10 640t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 10, 640, true, true>), g2(10), dim3(640), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY10 704t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 10, 704, true, true>), g2(10), dim3(704), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY10 512t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 10, 512, true, true>), g2(10), dim3(512), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY11 704t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 11, 704, true, true>), g2(11), dim3(704), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY9 576t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 9, 576, true, true>), g2(9), dim3(576), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY8 512t nt", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 8, 512, true, false>), g2(8), dim3(512), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY8 512t nt xcd", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 8, 512, true, true>), g2(8), dim3(512), 0, s, a, ex); }, {}});
        vars.push_back({"step3 TY16 1024t nt", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step3_tile<T, 16, 1024, true, false>), g2(16), dim3(1024), 0, s, a, ex); }, {}});
        vars.push_back({"step2 TY16 1024t nt", [=](const KArgs<T>& a) { hipLaunchKernelGGL((k_step2_tile<T, 16, 1024, true>), g2(16), dim3(1024), 0, s, a, ex); }, {}});
    }
    if (g_check) {   // step2 variants: 6 launches must equal 12 single steps, bit for bit
        std::vector<T> ref(b.total), got(b.total);
        b.init();
        for (int k = 0; k < 12; ++k) { hipLaunchKernelGGL((k_step_site<T, MODE_STEP>), grid1, dim3(256), 0, s, b.args(b.t & 1)); ++b.t; }
        CK(hipMemcpyAsync(ref.data(), b.A, b.total * sizeof(T), hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        for (auto& v : vars) {
            const bool s2 = v.name.rfind("step2", 0) == 0, s3 = v.name.rfind("step3", 0) == 0;
            if (!s2 && !s3) continue;
            b.init();
            const int nl = s2 ? 6 : 4, per = s2 ? 2 : 3;      // 12 iterations either way, result lands in A
            for (int k = 0; k < nl; ++k) { KArgs<T> a = b.args(k & 1); a.t = per * k; v.fn(a); }
            CK(hipMemcpyAsync(got.data(), b.A, b.total * sizeof(T), hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            size_t diff = 0; for (size_t k = 0; k < b.total; ++k) diff += (memcmp(&ref[k], &got[k], sizeof(T)) != 0);
            printf("CHECK %-24s %s (%zu differing elements)\n", v.name.c_str(), diff ? "MISMATCH" : "bit-identical to 12 single steps", diff);
        }
    }
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vars) {
            b.init();
            b.run(v.fn, 20);   // warm
            v.ms.push_back(b.run(v.fn, reps));
        }
    printf("== %dx%d %s %s rowstride=%d plane=%ld elems (pad %ld, pitchpad %d)  %d launches x %d rounds\n", nx, ny,
           sizeof(T) == 8 ? "f64" : "f32", rowil ? "ROW-INTERLEAVED" : "PLANAR", b.pitch, b.plane, pad, pitch_pad, reps, rounds);
    for (auto& v : vars) {
        std::sort(v.ms.begin(), v.ms.end());
        const double lup = (v.name.rfind("step2", 0) == 0) ? 2.0 : (v.name.rfind("step3", 0) == 0) ? 3.0 : 1.0;
        for (auto& m : v.ms) m /= lup;   // per lattice update
        const double med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("  %-26s median %8.2f us  %7.1f GB/s (%.1f%% of 8 TB/s)   best %8.2f us %7.1f GB/s\n", v.name.c_str(), med * 1e3,
               bytes / med / 1e6, bytes / med / 1e6 / 80.0, mn * 1e3, bytes / mn / 1e6);
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    int nx = 4096, ny = 1024, reps = 200, rounds = 3;
    std::string prec = "f64";
    std::vector<long> pads = {0};
    std::vector<long> ppads = {0};
    bool rowil = false;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        if (k == "--nx") nx = atoi(argv[++i]);
        else if (k == "--ny") ny = atoi(argv[++i]);
        else if (k == "--reps") reps = atoi(argv[++i]);
        else if (k == "--rounds") rounds = atoi(argv[++i]);
        else if (k == "--prec") prec = argv[++i];
        else if (k == "--layout") rowil = (std::string(argv[++i]) == "rowil");
        else if (k == "--variants") g_filter = argv[++i];
        else if (k == "--leak") g_leak = true;
        else if (k == "--check") g_check = true;
        else if (k == "--trials") g_trials = atoi(argv[++i]);
        else if (k == "--pitchpads") { ppads.clear(); char* tok = strtok(argv[++i], ","); while (tok) { ppads.push_back(atol(tok)); tok = strtok(nullptr, ","); } }
        else if (k == "--pads") { pads.clear(); char* tok = strtok(argv[++i], ","); while (tok) { pads.push_back(atol(tok)); tok = strtok(nullptr, ","); } }
    }
    for (int tr = 0; tr < g_trials; ++tr)
    for (long pp : ppads)
        for (long pad : pads) {
            if (prec == "f64") suite<double>(nx, ny, pad, reps, rounds, rowil, (int)pp);
            else suite<float>(nx, ny, pad, reps, rounds, rowil, (int)pp);
        }
    return 0;
}
