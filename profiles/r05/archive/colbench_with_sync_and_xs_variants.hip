// tools/colbench.hip — development harness of the register-resident column kernel k_stepc_col (tuning tool, not part of
// the product or the parity path): every shape is first held bit for bit against D single-iteration launches of
// k_step_site on a small grid with every boundary and the cylinder inside, then timed on the headline grid next to the
// LDS-image kernel k_stepd_tile<32,16,5> (round 2's production shape, no longer built into the library).
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -o tools/colbench tools/colbench.hip
//        (+ -DLBM_COL_PROF -o tools/colbench_prof: in-kernel phase record, `--prof DIR --filter NAME`)
#include "../highperformancecomputing-latticeboltzmannmethod_amd/csrc/lbm_kernel_col.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

using namespace lbmk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static bool g_map = false;
static int g_ntl = 0;
static int g_pad = 0;       // --pad N: N x 128 B added to every sub-row (channel-mapping experiment)       // --ntl 1: non-temporal level-1 loads in the register kernel
template <typename T>
struct Lattice {
    int nx, ny, pitch, xoff;
    long plane;
    size_t total;
    T *A, *B, *d_feq;
    int* d_unst;
    int* d_zero;
    hipStream_t s;
    hipEvent_t e0, e1;
    double u_in;
    int cur = 0;      // buffer holding the current state (0 = A)
    int t = 0;
    // pipelined launches (pipe_variant): one stream per part of the lattice, one event per part and step parity
    hipStream_t ps[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pe[2][4];
    int pstep = 0;
    void pipe_init() {
        if (ps[0]) return;
        for (int p = 0; p < 4; ++p) { CK(hipStreamCreateWithFlags(&ps[p], hipStreamNonBlocking)); for (int k = 0; k < 2; ++k) CK(hipEventCreateWithFlags(&pe[k][p], hipEventDisableTiming)); }
    }
    Lattice(int nx_, int ny_) : nx(nx_), ny(ny_) {
        const int per128 = 128 / sizeof(T);
        xoff = per128;
        const int pitch0 = (xoff + nx + 1 + per128 - 1) / per128 * per128 + g_pad * per128;
        plane = pitch0; pitch = Q * pitch0; total = (size_t)pitch * (ny + 2 * GR);     // row-interleaved
        CK(hipMalloc(&A, (total + 64) * sizeof(T)));
        CK(hipMalloc(&B, (total + 64) * sizeof(T)));
        CK(hipMalloc(&d_unst, sizeof(int)));
        CK(hipMalloc(&d_zero, sizeof(int)));
        CK(hipMemset(d_zero, 0, sizeof(int)));
        CK(hipMalloc(&d_feq, Q * sizeof(T)));
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        u_in = 200.0 * (0.1 / 3.0) / (0.1 * ny);
        if (u_in > 0.1) u_in = 0.1;
    }
    ~Lattice() { hipFree(A); hipFree(B); hipFree(d_unst); hipFree(d_feq); hipStreamDestroy(s); hipEventDestroy(e0); hipEventDestroy(e1); }
    KArgs<T> args(int depth_t) {
        KArgs<T> a{};
        a.y_lo = 0; a.y_cnt = ny; a.reverse = 0;
        a.src = cur ? B : A; a.dst = cur ? A : B;
        a.plane = plane; a.pitch = pitch; a.xoff = xoff; a.nx = nx; a.ny_loc = ny; a.ny_glob = ny; a.y_start = 0;
        a.cyl_x = (int)(0.2 * nx); a.cyl_y = (int)(0.5 * ny);
        const int r = (int)(0.05 * ny); a.cyl_r2 = (double)(r * r);
        a.tau_inv = (T)(1.0 / 0.6); a.u_in = (T)u_in; a.unstable_t = d_unst; a.t = depth_t; a.t_base = d_zero;
        return a;
    }
    K2Extra<T> extra() { K2Extra<T> e; e.feq_in = d_feq; e.xcd = 1; e.nt = 1; e.ntl = g_ntl; e.small = (total * sizeof(T) + 4096 < (size_t(1) << 32)) ? 1 : 0; return e; }
    void init() {
        InitArgs<T> ia;
        ia.a = A; ia.b = B; ia.plane = plane; ia.pitch = pitch; ia.xoff = xoff; ia.nx = nx; ia.ny_loc = ny;
        ia.ny_glob = ny; ia.y_start = 0; ia.cyl_x = (int)(0.2 * nx); ia.cyl_y = (int)(0.5 * ny);
        const int r = (int)(0.05 * ny); ia.cyl_r2 = (double)(r * r);
        const double ux = u_in, usq = ux * ux, t3 = 1.5 * usq;
        ia.feq_in[0] = (T)(wgt<double>(0) * (1.0 - 1.5 * usq));
        for (int i = 1; i < Q; ++i) { const double cu = cx(i) * ux; ia.feq_in[i] = (T)(wgt<double>(i) * (((1.0 + 3.0 * cu) - t3) + 4.5 * cu * cu)); }
        CK(hipMemcpy(d_feq, ia.feq_in, sizeof(ia.feq_in), hipMemcpyHostToDevice));
        ia.solid_count = d_unst;
        hipLaunchKernelGGL((k_init<T>), dim3((nx + 2 + 255) / 256, ny + 2 * GR), dim3(256), 0, s, ia);
        const int big = 0x7fffffff;
        CK(hipMemcpyAsync(d_unst, &big, sizeof(int), hipMemcpyHostToDevice, s));
        cur = 0; t = 0; pstep = 0;
        // iteration 0's collision: A -> B
        KArgs<T> a = args(0);
        hipLaunchKernelGGL((k_step_site<T, MODE_COLLIDE_ONLY, false, AR_CONTRACTED>), dim3((nx + 255) / 256, ny), dim3(256), 0, s, a);
        cur = 1;
        CK(hipStreamSynchronize(s));
    }
    void flip(int depth) { cur ^= 1; t += depth; }
    std::vector<T> download() {
        std::vector<T> h(total);
        CK(hipMemcpyAsync(h.data(), cur ? B : A, total * sizeof(T), hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        return h;
    }
    int unstable() { int v; CK(hipMemcpyAsync(&v, d_unst, sizeof(int), hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); return v; }
};

template <typename T> struct Variant { std::string name; int depth; std::function<void(Lattice<T>&)> launch; int waves = 8; };

template <typename T, int R, int NW, int D, int AR>
Variant<T> col_variant(bool nt, bool alt = false, int persist = 0) {
    char nm[96];
    snprintf(nm, sizeof(nm), "col R=%d NW=%d D=%d %s %s p%d", R, NW, D, nt ? "nt" : "  ", alt ? "alt" : "", persist);
    return {nm, D, [=](Lattice<T>& L) {
        constexpr int OW = 64 - 2 * (D - 1), OH = R * NW - 2 * (D - 1);
        int nb = ((L.nx + OW - 1) / OW) * ((L.ny + OH - 1) / OH);
        (void)persist;
        dim3 grid((nb + 7) / 8 * 8);
        KArgs<T> a = L.args(L.t);
        a.reverse = alt ? (L.cur & 1) : 0;       // walk the tile rows top-down on every other launch
        if (nt) hipLaunchKernelGGL((k_stepc_col<T, R, NW, D, true, AR>), grid, dim3(NW * 64), 0, L.s, a, L.extra());
        else hipLaunchKernelGGL((k_stepc_col<T, R, NW, D, false, AR>), grid, dim3(NW * 64), 0, L.s, a, L.extra());
    }};
}
static int g_cus = 256;
// round 5: the waves of a block meet through neighbour flags instead of one barrier per level (k_stepc_col<.., SYNC = 1>)
template <typename T, int R, int NW, int D, int AR, int SYNC = 1>
Variant<T> colsync_variant(bool alt = false) {
    char nm[96];
    snprintf(nm, sizeof(nm), "colsync%d R=%d NW=%d D=%d %s", SYNC, R, NW, D, alt ? "alt" : "");
    return {nm, D, [=](Lattice<T>& L) {
        constexpr int OW = 64 - 2 * (D - 1), OH = R * NW - 2 * (D - 1);
        const int nb = ((L.nx + OW - 1) / OW) * ((L.ny + OH - 1) / OH);
        KArgs<T> a = L.args(L.t);
        a.reverse = alt ? (L.cur & 1) : 0;
        hipLaunchKernelGGL((k_stepc_col<T, R, NW, D, false, AR, SYNC>), dim3((nb + 7) / 8 * 8), dim3(NW * 64), 0, L.s, a, L.extra());
    }, NW};
}
// round 5: x-shifts through the LDS crossbar (ds_bpermute) instead of DPP moves (k_stepc_col<.., XS>)
template <typename T, int R, int NW, int D, int AR, int XS>
Variant<T> colxs_variant(bool alt = true) {
    char nm[96];
    snprintf(nm, sizeof(nm), "colxs%d R=%d NW=%d D=%d %s", XS, R, NW, D, alt ? "alt" : "");
    return {nm, D, [=](Lattice<T>& L) {
        constexpr int OW = 64 - 2 * (D - 1), OH = R * NW - 2 * (D - 1);
        const int nb = ((L.nx + OW - 1) / OW) * ((L.ny + OH - 1) / OH);
        KArgs<T> a = L.args(L.t);
        a.reverse = alt ? (L.cur & 1) : 0;
        hipLaunchKernelGGL((k_stepc_col<T, R, NW, D, false, AR, 0, XS>), dim3((nb + 7) / 8 * 8), dim3(NW * 64), 0, L.s, a, L.extra());
    }, NW};
}
template <typename T, int TX, int TY, int D, int AR>
Variant<T> tile_variant() {
    char nm[96];
    snprintf(nm, sizeof(nm), "lds tile %dx%d D=%d nt", TX, TY, D);
    return {nm, D, [=](Lattice<T>& L) {
        dim3 grid((L.nx + TX - 1) / TX, (L.ny + TY - 1) / TY);
        KArgs<T> a = L.args(L.t);
        hipLaunchKernelGGL((k_stepd_tile<T, TX, TY, D, AR>), grid, dim3(TX * TY), 0, L.s, a, L.extra());
    }, TX * TY / 64};
}

template <typename T, int AR>
std::vector<Variant<T>> variants() {
    std::vector<Variant<T>> v;
    v.push_back(tile_variant<T, 32, 16, 5, AR>());
    v.push_back(tile_variant<T, 32, 32, 8, AR>());
    v.push_back(tile_variant<T, 64, 16, 6, AR>());
    v.push_back(col_variant<T, 4, 8, 5, AR>(true));
    v.push_back(col_variant<T, 4, 8, 6, AR>(true));
    v.push_back(col_variant<T, 4, 8, 6, AR>(false));
    v.push_back(col_variant<T, 4, 8, 6, AR>(false, true));
    v.push_back(colsync_variant<T, 4, 8, 6, AR>(false));
    v.push_back(colsync_variant<T, 4, 8, 6, AR>(true));
    v.push_back(colsync_variant<T, 4, 8, 7, AR>(true));
    v.push_back(colsync_variant<T, 4, 8, 6, AR, 2>(true));
    v.push_back(colxs_variant<T, 4, 8, 6, AR, 1>());
    v.push_back(colxs_variant<T, 4, 8, 6, AR, 2>());
    v.push_back(colxs_variant<T, 4, 8, 6, AR, 3>());
    v.push_back(colxs_variant<T, 4, 8, 7, AR, 2>());
    v.push_back(col_variant<T, 4, 8, 7, AR>(false));
    v.push_back(col_variant<T, 4, 8, 7, AR>(false, true));
    v.push_back(col_variant<T, 3, 8, 5, AR>(true));
    v.push_back(col_variant<T, 3, 8, 6, AR>(true));
    v.push_back(col_variant<T, 3, 8, 6, AR>(false, true));
    if constexpr (sizeof(T) == 8) {        // two fp64 rows per thread on 12 waves: the 64x24 region of the strict shape with 24 waves per CU
        v.push_back(col_variant<T, 2, 12, 5, AR>(true));
        v.push_back(col_variant<T, 2, 12, 5, AR>(false, true));
        v.push_back(col_variant<T, 2, 12, 6, AR>(true));
        v.push_back(col_variant<T, 2, 12, 6, AR>(false, true));
    }
    if constexpr (sizeof(T) == 8) {        // fp64: 64x64 regions need 16 waves (one block per CU)
        v.push_back(col_variant<T, 4, 16, 6, AR>(false, true));
        v.push_back(col_variant<T, 4, 16, 7, AR>(false, true));
        v.push_back(col_variant<T, 4, 16, 8, AR>(false, true));
    }
    if constexpr (sizeof(T) == 4) {        // fp32: taller regions (state = 9 R registers per thread)
        v.push_back(col_variant<T, 6, 8, 6, AR>(false, true));
        v.push_back(col_variant<T, 6, 8, 7, AR>(false, true));
        v.push_back(col_variant<T, 8, 8, 6, AR>(false, true));
        v.push_back(col_variant<T, 8, 8, 7, AR>(false, true));
        v.push_back(col_variant<T, 8, 8, 8, AR>(false, true));
        v.push_back(col_variant<T, 8, 8, 8, AR>(true, true));
        // (trimmed exchange buffer: more waves per block at the standard register budget)
        v.push_back(col_variant<T, 4, 12, 6, AR>(false, true));
        v.push_back(col_variant<T, 4, 12, 7, AR>(false, true));
        v.push_back(col_variant<T, 4, 12, 6, AR>(true, false));
        v.push_back(col_variant<T, 3, 16, 6, AR>(false, true));
        v.push_back(col_variant<T, 3, 16, 7, AR>(false, true));
        v.push_back(col_variant<T, 2, 16, 6, AR>(false, true));
    }
    return v;
}

// bit-equality with single-iteration launches on a small grid: inlet, outlet, both walls, the cylinder, partial tiles
template <typename T, int AR>
int check(int nx, int ny, int launches) {
    int failures = 0;
    for (auto& v : variants<T, AR>()) {
        Lattice<T> ref(nx, ny), got(nx, ny);
        ref.init(); got.init();
        for (int k = 0; k < launches * v.depth; ++k) {
            KArgs<T> a = ref.args(ref.t);
            hipLaunchKernelGGL((k_step_site<T, MODE_STEP, false, AR>), dim3((nx + 255) / 256, ny), dim3(256), 0, ref.s, a);
            ref.flip(1);
        }
        for (int k = 0; k < launches; ++k) { v.launch(got); got.flip(v.depth); }
        CK(hipGetLastError());
        const std::vector<T> r = ref.download(), g = got.download();
        size_t diff = 0, first = 0;
        for (size_t k = 0; k < r.size(); ++k)
            if (memcmp(&r[k], &g[k], sizeof(T)) != 0) { if (!diff) first = k; ++diff; }
        const int ur = ref.unstable(), ug = got.unstable();
        printf("CHECK %dx%d %s ar=%d %-24s %s", nx, ny, sizeof(T) == 8 ? "f64" : "f32", AR, v.name.c_str(),
               diff ? "MISMATCH" : "bit-identical");
        if (diff) {
            const size_t row = first / ref.pitch, rem = first % ref.pitch;
            printf(" (%zu elements; first at gy=%zu plane=%zu col=%zu)", diff, row, rem / ref.plane, rem % ref.plane);
        }
        if (ur != ug) printf("  UNSTABLE-FLAG %d vs %d", ug, ur);
        printf("\n");
        if (diff && g_map) {   // which cells differ (any plane): one character per cell, rows top-down
            for (int y = ny - 1; y >= 0; --y) {
                std::string line;
                for (int x = 0; x < nx; ++x) {
                    int m = 0;
                    for (int i = 0; i < Q; ++i) {
                        const size_t k = (size_t)(y + GR) * ref.pitch + (size_t)i * ref.plane + ref.xoff + x;
                        if (memcmp(&r[k], &g[k], sizeof(T)) != 0) m |= 1 << i;
                    }
                    line += m == 0 ? '.' : (m == 511 ? '#' : (char)('a' + __builtin_popcount(m)));
                }
                printf("%3d %s\n", y, line.c_str());
            }
        }
        failures += (diff != 0) + (ur != ug);
    }
    return failures;
}

template <typename T, int AR>
void timeit(int nx, int ny, int reps, int rounds, const std::string& filter) {
    Lattice<T> L(nx, ny);
    auto vs = variants<T, AR>();
    std::vector<std::vector<double>> us(vs.size());
    for (int r = 0; r < rounds; ++r)
        for (size_t k = 0; k < vs.size(); ++k) {
            if (!filter.empty() && vs[k].name.find(filter) == std::string::npos) continue;
            L.init();
            for (int i = 0; i < 10; ++i) { vs[k].launch(L); L.flip(vs[k].depth); }
            CK(hipEventRecord(L.e0, L.s));
            for (int i = 0; i < reps; ++i) { vs[k].launch(L); L.flip(vs[k].depth); }
            CK(hipEventRecord(L.e1, L.s));
            CK(hipEventSynchronize(L.e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, L.e0, L.e1));
            us[k].push_back(ms * 1e3 / reps);
        }
    printf("== %dx%d %s ar=%d, %d launches x %d rounds\n", nx, ny, sizeof(T) == 8 ? "f64" : "f32", AR, reps, rounds);
    for (size_t k = 0; k < vs.size(); ++k) {
        if (us[k].empty()) continue;
        std::sort(us[k].begin(), us[k].end());
        const double med = us[k][us[k].size() / 2], best = us[k][0];
        printf("  %-26s %8.2f us/launch  %6.2f us/iteration  %8.1f GLUPS (median)   best %8.1f GLUPS\n", vs[k].name.c_str(), med,
               med / vs[k].depth, (double)nx * ny * vs[k].depth / med * 1e-3, (double)nx * ny * vs[k].depth / best * 1e-3);
    }
    fflush(stdout);
}

#ifdef LBM_COL_PROF
// one profiled launch of every variant whose name contains `filter`: the raw marks as CSV (tools/prof_phases.py reads them)
template <typename T, int AR>
void profile(int nx, int ny, const std::string& filter, const std::string& outdir) {
    Lattice<T> L(nx, ny);
    for (auto& v : variants<T, AR>()) {
        if (v.name.find(filter) == std::string::npos) continue;
        const size_t nblk = 8192, nw = (size_t)v.waves, n = nblk * nw * PROF_SLOTS;
        unsigned long long* d; CK(hipMalloc(&d, n * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(lbmk::lbm_prof_buf), &d, sizeof(d)));
        L.init();
        for (int i = 0; i < 12; ++i) { v.launch(L); L.flip(v.depth); }
        CK(hipStreamSynchronize(L.s));
        CK(hipMemset(d, 0, n * 8));
        CK(hipEventRecord(L.e0, L.s));
        v.launch(L); L.flip(v.depth);
        CK(hipEventRecord(L.e1, L.s));
        CK(hipEventSynchronize(L.e1));
        float ms; CK(hipEventElapsedTime(&ms, L.e0, L.e1));
        std::vector<unsigned long long> h(n);
        CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
        std::string nm = v.name; for (auto& c : nm) if (c == ' ' || c == '=') c = '_';
        const std::string path = outdir + "/prof_" + nm + ".csv";
        FILE* fp = fopen(path.c_str(), "w");
        fprintf(fp, "# %s %dx%d launch_us=%.2f\nblock,wave,xcc,hwid,realtime", v.name.c_str(), nx, ny, ms * 1e3);
        for (int k = 0; k < PROF_SLOTS - 2; ++k) fprintf(fp, ",t%d", k);
        fprintf(fp, "\n");
        for (size_t b = 0; b < nblk; ++b)
            for (size_t w = 0; w < nw; ++w) {
                const unsigned long long* r = &h[(b * nw + w) * PROF_SLOTS];
                if (!r[PROF_SLOTS - 2]) continue;
                fprintf(fp, "%zu,%zu,%llu,%llu,%llu", b, w, r[PROF_SLOTS - 1] >> 32, r[PROF_SLOTS - 1] & 0xffffffffull, r[PROF_SLOTS - 2]);
                for (int k = 0; k < PROF_SLOTS - 2; ++k) fprintf(fp, ",%llu", r[k]);
                fprintf(fp, "\n");
            }
        fclose(fp);
        printf("PROF %s: %.2f us/launch (with marks) -> %s\n", v.name.c_str(), ms * 1e3, path.c_str());
        CK(hipFree(d));
    }
}
#endif

int main(int argc, char** argv) {
    int nx = 4096, ny = 1024, reps = 100, rounds = 3;
    bool do_check = true, do_time = true, strict = false;
    std::string prec = "f64", filter, profdir;
    int mnx = 0, mny = 0, mlaunch = 1;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        if (k == "--nx") nx = atoi(argv[++i]);
        else if (k == "--ny") ny = atoi(argv[++i]);
        else if (k == "--reps") reps = atoi(argv[++i]);
        else if (k == "--rounds") rounds = atoi(argv[++i]);
        else if (k == "--prec") prec = argv[++i];
        else if (k == "--no-check") do_check = false;
        else if (k == "--no-time") do_time = false;
        else if (k == "--strict") strict = true;
        else if (k == "--filter") filter = argv[++i];
        else if (k == "--prof") profdir = argv[++i];
        else if (k == "--ntl") g_ntl = atoi(argv[++i]);
        else if (k == "--pad") g_pad = atoi(argv[++i]);
        else if (k == "--map") { g_map = true; mnx = atoi(argv[++i]); mny = atoi(argv[++i]); mlaunch = atoi(argv[++i]); }
    }
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, 0) == hipSuccess && pr.multiProcessorCount > 0) g_cus = pr.multiProcessorCount; }
#ifdef LBM_COL_PROF
    if (!profdir.empty()) { profile<double, AR_CONTRACTED>(nx, ny, filter, profdir); return 0; }
#endif
    int failures = 0;
    if (g_map) { if (prec == "f64") check<double, AR_CONTRACTED>(mnx, mny, mlaunch); else check<float, AR_CONTRACTED>(mnx, mny, mlaunch); return 0; }
    if (do_check) {
        if (prec == "f64") { failures += check<double, AR_CONTRACTED>(300, 170, 3); failures += check<double, AR_STRICT>(300, 170, 2); failures += check<double, AR_CONTRACTED>(1024, 256, 4); }
        else { failures += check<float, AR_CONTRACTED>(300, 170, 3); failures += check<float, AR_CONTRACTED>(1024, 256, 4); }
        printf("check: %d failure(s)\n", failures);
        fflush(stdout);
    }
    if (do_time && !failures) {
        if (prec == "f64") { timeit<double, AR_CONTRACTED>(nx, ny, reps, rounds, filter); if (strict) timeit<double, AR_STRICT>(nx, ny, reps, rounds, filter); }
        else timeit<float, AR_CONTRACTED>(nx, ny, reps, rounds, filter);
    }
    return failures ? 1 : 0;
}
