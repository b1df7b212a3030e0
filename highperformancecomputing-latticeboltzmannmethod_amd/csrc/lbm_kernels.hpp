// csrc/lbm_kernels.hpp — device code of the D2Q9-BGK timestep for gfx950 (CDNA4, wave64).
//
// One persistent state per strip: the POST-COLLISION populations P_t (the reference's f_next right after
// collision_step() of iteration t, LBMSolver.h:84-126), stored as 9 SoA planes with a one-cell ghost frame:
//
//     plane i, local row gy in [0, ny_loc+2*GR), column col:   base[i*plane + gy*pitch + col]
//     interior cell (x, y)  <->  gy = y+GR, col = xoff + x      (xoff*sizeof(T) is a multiple of 128 B)
// GR = 12 ghost rows on each side, of which a strip exchanges six (once per launch of up to six fused iterations, or once
// per TWO launches of up to three: the first launch of such a pair also updates three ghost rows per internal face,
// redundantly with the neighbour, so that the second one finds valid inputs); one ghost column on each side.
// The two strides describe either of two layouts chosen by the host (lbm_hip.hip, "plan"):
//     PLANAR          plane = rows*pitch0 (+pad), pitch = pitch0            nine separate planes
//     ROW-INTERLEAVED plane = pitch0,             pitch = 9*pitch0          [gy][i][col]: the nine sub-rows of a
//                     lattice row are adjacent, so a block's 18 streams stay inside two ~300 KB windows
// Every kernel below is layout-agnostic: it only uses (plane, pitch).
//
// Ghost cells hold, permanently and in BOTH A/B buffers, what the reference's halo logic leaves in them on one
// rank: E/W ghost columns of globally-interior rows = 0, physical N/S ghost rows and the
// four corner ghosts = the initial equilibrium; solid cells hold w_i for ever (N4). Strip-internal ghost rows
// are refreshed every step by the halo exchange. The step kernel therefore needs no boundary branches for
// its nine pulls and writes fluid interior cells only.
//
// Kernel families (within one arithmetic mode all evaluate the SAME per-cell operation sequence => bit-identical results):
//   k_step_site                  one iteration per launch, 144 B of HBM traffic per lattice update (fp64)
//   k_step2/3/4_tile             two / three / four iterations per launch over 64 x TY tiles, intermediate states in LDS
//                                (what a call's last iterations, short strips and the strict mode's measurement use)
//   k_stepd_tile                 six / seven / eight iterations on an LDS-filling tile: grids of a single round of blocks
//   k_stepc_col                  (lbm_kernel_col.hpp) five / six iterations with the lattice of a 64x32 region held in
//                                registers: the production kernel of large grids and tall strips
//   k_init, k_macros, k_forces, k_halo_pack/unpack   set-up and the output cadence
// Arithmetic modes of the collision (enum Arith): strict IEEE operation by operation (bit-identical to the CPU oracle)
// or FMA-contracted with one reciprocal (what the reference's -ffast-math -mfma build permits; <= 1e-10).
//
// Step kernel K_t (one iteration = one loop body of Solver::run, LBMSolver.h:49-60):
//     pull from P_t        == exchange_ghost_cells + streaming_step        (LBMGrid.h:249, LBMSolver.h:128-145)
//     wall / inlet / outlet== apply_boundary_conditions, sequential order  (LBMSolver.h:147-236)
//     stability test       == Grid::check_stability                        (LBMGrid.h:285-317)
//     moments + BGK        == collision_step of iteration t+1              (LBMSolver.h:84-126)
//     store P_{t+1}
// Algorithmic traffic: 9 loads + 9 stores per lattice update = 144 B (fp64) / 72 B (fp32). HBM-bound; no MFMA.
#pragma once
#include <utility>
#include <hip/hip_runtime.h>

namespace lbmk {

constexpr int Q = 9;
constexpr int GR = 12;  // ghost rows ALLOCATED below and above the strip (the frame every kernel addresses rows by: gy = y + GR)
constexpr int HR1 = 6;  // ghost rows a strip's halo exchange refreshes per face (LBM_HALO_ROWS): one launch of up to six fused
                        // iterations, or two of up to three (the first of the pair recomputes three of the neighbour's rows), between
                        // exchanges. With "deep_halo" 2 (round 4) a deep plan exchanges all GR = 12 rows once per TWO launches of up to six
                        // iterations each: the first launch of such a pair also updates the six ghost rows next to each internal face.
// LBMConfig.h:13-34 — direction numbering is observable through f_current(x,y,i), keep it.
__host__ __device__ constexpr int cx(int i) { constexpr int v[Q] = {0, 1, 0, -1, 0, 1, -1, -1, 1}; return v[i]; }
__host__ __device__ constexpr int cy(int i) { constexpr int v[Q] = {0, 0, 1, 0, -1, 1, 1, -1, -1}; return v[i]; }
__host__ __device__ constexpr int opp(int i) { constexpr int v[Q] = {0, 3, 4, 1, 2, 7, 8, 5, 6}; return v[i]; }
template <typename T> __host__ __device__ constexpr T wgt(int i) {
    return i == 0 ? T(4.0 / 9.0) : (i < 5 ? T(1.0 / 9.0) : T(1.0 / 36.0));
}

template <typename T>
struct KArgs {
    const T* src;      // plane 0 of the buffer read  (P_t)
    T* dst;            // plane 0 of the buffer written (P_{t+1})
    long plane;        // elements per plane
    int pitch;         // elements per row
    int xoff;          // column of interior x = 0
    int nx, ny_loc;    // interior size of this strip
    int ny_glob;       // global rows
    int y_start;       // global row of local y = 0
    int cyl_x, cyl_y;  // LBMConfig.h:61-63 (integer cells)
    double cyl_r2;     // (double)(r*r), LBMGrid.h:169
    T tau_inv;         // 1/tau, LBMSolver.h:85
    T u_in;            // inlet velocity
    int* unstable_t;   // device word: first unstable iteration (INT_MAX if none)
    int t;             // iteration this launch completes, RELATIVE to *t_base (stability bookkeeping only)
    const int* t_base; // device word: the iteration `t` counts from. A launch replayed from a hipGraph carries a fixed `t`; the
                       // word is advanced on the device between replays. Read on the (rare) unstable path only.
    int y_lo, y_cnt;   // local rows [y_lo, y_lo + y_cnt) covered by this launch
    int y_lo2, y_cnt2; // optional second range [y_lo2, y_lo2 + y_cnt2) of the same launch (both edge bands of a strip
                       // in one grid); y_cnt2 == 0: none
    int reverse;       // 1: blockIdx.y walks the rows top-down (alternated per launch by the host, see row_of_block)
};

// Row handled by blockIdx.y. Blocks are dispatched roughly in index order; walking the rows in the opposite
// direction on every other step makes a step start on the rows the previous step wrote last, which are the
// ones most likely still resident in the 256 MiB Infinity Cache (measured +0..8 % at 4096x1024 fp64).
template <typename T>
__device__ __forceinline__ int row_of_block(const KArgs<T>& a) {
    const int by = (int)blockIdx.y;            // grid.y == y_cnt + y_cnt2
    if (by >= a.y_cnt) return a.y_lo2 + (by - a.y_cnt);
    return a.y_lo + (a.reverse ? a.y_cnt - 1 - by : by);
}

// Tile band of a fused launch: bands of TY rows over the first range, then over the second (grid.y = both counts).
// Returns the first row of the band and, through y_end, the end of the range it belongs to.
template <typename T>
__device__ __forceinline__ int band_origin(const KArgs<T>& a, int by, int TY, int& y_end) {
    const int nb1 = (a.y_cnt + TY - 1) / TY;
    if (by >= nb1) { y_end = a.y_lo2 + a.y_cnt2; return a.y_lo2 + (by - nb1) * TY; }
    y_end = a.y_lo + a.y_cnt;
    return a.y_lo + by * TY;
}

// Grid::setup_geometry, LBMGrid.h:152-173, as a pure function of GLOBAL integer coordinates.
__device__ __forceinline__ bool is_solid_cell(int x, int yg, int cyl_x, int cyl_y, double cyl_r2) {
    const double dx = (double)(x - cyl_x), dy = (double)(yg - cyl_y);
    return dx * dx + dy * dy <= cyl_r2;
}

// Block-uniform test: can any cell of the tile [X0, X0+TX) x [Y0, Y0+TY) grown by `ring` cells be solid? (bounding
// boxes in integer cells; cyl_r2 = r*r exactly, so r is recovered by an exact sqrt of a perfect square).
template <typename T>
__device__ __forceinline__ bool tile_near_cylinder(const KArgs<T>& a, int X0, int Y0, int TX, int TY, int ring) {
    const int r = (int)sqrt(a.cyl_r2) + 1;
    const int yg0 = a.y_start + Y0;
    return X0 - ring <= a.cyl_x + r && X0 + TX - 1 + ring >= a.cyl_x - r &&
           yg0 - ring <= a.cyl_y + r && yg0 + TY - 1 + ring >= a.cyl_y - r;
}

// apply_boundary_conditions on the pulled populations of ONE cell, in the reference's sequential loop order
// bottom -> top -> inlet -> outlet (LBMSolver.h:152-236). Solid cells are skipped by every one
// of those loops. Returns nothing; rho_bc/u_out are exposed for the macro snapshot kernel.
template <typename T>
__device__ __forceinline__ void apply_bcs(T (&f)[Q], bool bottom, bool top, bool inlet, bool outlet, T u_in,
                                          T& rho_bc, T& u_out) {
    if (bottom) { f[2] = f[4]; f[5] = f[7]; f[6] = f[8]; }                    // :155-163
    if (top)    { f[4] = f[2]; f[7] = f[5]; f[8] = f[6]; }                    // :168-176
    if (inlet) {                                                              // :181-206 (Zou-He velocity)
        rho_bc = (f[0] + f[2] + f[4] + T(2.0) * (f[3] + f[6] + f[7])) / (T(1.0) - u_in);
        f[1] = f[3] + T(2.0 / 3.0) * rho_bc * u_in;
        f[5] = f[7] - T(0.5) * (f[2] - f[4]) + T(1.0 / 6.0) * rho_bc * u_in;
        f[8] = f[6] + T(0.5) * (f[2] - f[4]) + T(1.0 / 6.0) * rho_bc * u_in;
    }
    if (outlet) {                                                             // :212-235 (Zou-He pressure, rho=1)
        const T rho_out = T(1.0);
        u_out = T(-1.0) + (f[0] + f[2] + f[4] + T(2.0) * (f[1] + f[5] + f[8])) / rho_out;
        f[3] = f[1] - T(2.0 / 3.0) * rho_out * u_out;
        f[6] = f[8] - T(0.5) * (f[2] - f[4]) - T(1.0 / 6.0) * rho_out * u_out;
        f[7] = f[5] + T(0.5) * (f[2] - f[4]) - T(1.0 / 6.0) * rho_out * u_out;
    }
}

// Arithmetic of the collision (the only place the two modes differ):
//   AR_STRICT     the reference's expression tree evaluated operation by operation in IEEE arithmetic, no contraction,
//                 two IEEE divisions: bit-identical to the strict CPU oracle (library default, parity tests).
//   AR_CONTRACTED the same formulas as fused multiply-adds with ONE reciprocal of rho (two Newton steps on v_rcp): what the
//                 reference's own build flags permit its compiler to do (CMakeLists.txt:21-22: -ffast-math -mfma). 70
//                 instead of 150 floating-point instructions per cell; rho/u stay within 1e-10 of the reference (tests).
enum Arith { AR_STRICT = 0, AR_CONTRACTED = 1 };

__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double recip_t(double d) {      // 1/d to an ulp: v_rcp_f64 + two Newton steps
    double r = __builtin_amdgcn_rcp(d);
    r = fma_t(fma_t(-d, r, 1.0), r, r);
    r = fma_t(fma_t(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float recip_t(float d) {        // v_rcp_f32 (1 ulp) + one Newton step
    float r = __builtin_amdgcn_rcpf(d);
    return fma_t(fma_t(-d, r, 1.0f), r, r);
}

// Buffer-descriptor access for the block-uniform "lean" paths: address = descriptor base + SGPR offset + ONE 32-bit VGPR
// offset, so the nine populations of a cell cost nine scalar adds and no vector address arithmetic at all (a global_load
// needs a vector add per population: the plane / row displacements are far beyond its immediate-offset range).
typedef unsigned lbm_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_desc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffff, 0x00020000);   // raw, no range limit
}
// AUX: cache policy of the load (0 default; 2 = nt, non-temporal: the line is not kept for re-use — round 4: the register kernel's
// level-1 loads read every line once per launch, and with nt loads + plain stores it runs 2-4 % faster at 4096x1024 fp64, 164-167
// against 158-163 GLUPS; with nt loads AND nt stores 140: the plan measurement chooses)
template <typename T, int AUX = 0> __device__ __forceinline__ T buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
    else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
}
template <bool NT> __device__ __forceinline__ void buf_store(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(lbm_u32x2, v), r, voff, soff, NT ? 2 : 0);   // aux 2 = nt
}
template <bool NT> __device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, NT ? 2 : 0);
}

// ux /= rho; uy /= rho (LBMSolver.h:108-109) in strict mode: two correctly rounded IEEE divisions by the SAME denominator.
// hipcc expands an fp64 division into v_div_scale x2, v_rcp_f64, two Newton steps on the reciprocal (four FMAs), q = a*r,
// rem = fma(-b, q, a), v_div_fmas (= fma(rem, r, q) unless the operands were scaled) and v_div_fixup (special values) — twice,
// because the scaling instruction takes the numerator too. For operands in the normal range (rho ~ 1, |rho u| < 1: nothing is
// scaled, nothing is special) the reciprocal chain depends on the denominator only, so the two divisions share it: the same
// operations on the same values in the same order => the same bits as the compiler's two expansions and as the oracle's divsd
// (every strict parity test holds the populations to np.array_equal), for 13 instead of ~26 instructions and a dozen fewer live
// registers. (A run that blows up is flagged by |f| > 1e5 long before rho or rho*u leave the range where no scaling happens.)
// RANGE of the bit-identity (tests/test_gpu_thin_spots.py::test_strict_div2_equals_ieee_division_in_its_range holds it on 4 M random
// operands through lbm_debug_strict_div2): denominators in [2^-20, 2^20], numerators 0 or of magnitude in [2^-400, 2^400] —
// populations of a stable run live in [1e-3, 1e1]. Outside it the chain is NOT an IEEE division: a -0 numerator gives +0 (the sign
// of a zero velocity is erased by the equilibrium's bracket and compares equal in every accessor), denormal quotients and
// exponent gaps beyond the fp64 range are not rescaled, rho == 0 gives NaN where IEEE gives +-inf (both flagged unstable).
__device__ __forceinline__ void strict_div2(double& a1, double& a2, double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = a1 * r;
    a1 = __builtin_fma(__builtin_fma(-b, q, a1), r, q);
    q = a2 * r;
    a2 = __builtin_fma(__builtin_fma(-b, q, a2), r, q);
}
__device__ __forceinline__ void strict_div2(float& a1, float& a2, float b) { a1 /= b; a2 /= b; }
// test kernel (lbm_debug_strict_div2): strict_div2 beside the compiler's own IEEE divisions, element by element
template <int UNUSED = 0>      // (a template: the header is included by three translation units)
__global__ void k_debug_strict_div2(const double* a1, const double* a2, const double* b, int n, double* q1, double* q2, double* r1, double* r2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double x = a1[k], y = a2[k];
    strict_div2(x, y, b[k]);
    q1[k] = x; q2[k] = y;
    r1[k] = a1[k] / b[k]; r2[k] = a2[k] / b[k];
}   // (fp32 has no oracle to be bit-equal to: plain divisions)

// collision_step for one cell, LBMSolver.h:101-123 (moments i = 0..8 ascending from 0, N7).
template <typename T, int AR = AR_STRICT>
__device__ __forceinline__ void bgk_collide(T (&f)[Q], T tau_inv) {
    T rho = T(0), ux = T(0), uy = T(0);
    if (AR == AR_CONTRACTED) {
        // moments as short trees over opposite pairs (depth 4 / 3 instead of chains of 9 / 6 dependent additions)
        const T a13 = f[1] + f[3], a24 = f[2] + f[4], a57 = f[5] + f[7], a68 = f[6] + f[8];
        const T d13 = f[1] - f[3], d24 = f[2] - f[4], d57 = f[5] - f[7], d68 = f[6] - f[8];
        rho = ((f[0] + a13) + (a24 + a57)) + a68;
        ux = (d13 + d57) - d68;
        uy = (d24 + d57) + d68;
        // v = 3u. 1 + 3cu + 4.5cu^2 - 1.5u^2 = base + cv*(1 + 0.5cv) with cv = c.v and base = 1 - v^2/6: the inner bracket
        // is one FMA with INLINE constants (0.5, 1.0). gfx950 VALU instructions take one SGPR/literal operand at most, so
        // fma(4.5, cu, 3.0) cost two v_mov per direction to build the 3.0 — 16 of ~100 vector instructions per cell.
        const T inv3 = recip_t(rho) * T(3.0);
        const T vx = ux * inv3, vy = uy * inv3;
        const T base = fma_t(T(-1.0 / 6.0), fma_t(vx, vx, vy * vy), T(1.0));
        const T wr0 = wgt<T>(0) * rho, wr1 = wgt<T>(1) * rho, wr5 = wgt<T>(5) * rho;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            T cv;
            if (cx(i) == 0 && cy(i) == 0) cv = T(0);
            else if (cy(i) == 0) cv = T(cx(i)) * vx;
            else if (cx(i) == 0) cv = T(cy(i)) * vy;
            else cv = T(cx(i)) * vx + T(cy(i)) * vy;
            const T t = (i == 0) ? base : fma_t(cv, fma_t(cv, T(0.5), T(1.0)), base);
            const T wr = i == 0 ? wr0 : (i < 5 ? wr1 : wr5);
            f[i] = fma_t(tau_inv, fma_t(wr, t, -f[i]), f[i]);     // f - (f - feq)/tau = f + (feq - f)/tau
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        rho += f[i];
        if (cx(i) != 0) ux += T(cx(i)) * f[i];
        if (cy(i) != 0) uy += T(cy(i)) * f[i];
    }
    strict_div2(ux, uy, rho);
    const T usq = ux * ux + uy * uy;
    // feq_i = w_i*rho*(1.0 + 3.0*cu + 4.5*cu*cu - 1.5*usq), cu = c_ix*ux + c_iy*uy with integer c (LBMSolver.h:119-121), evaluated
    // operation by operation in the reference's order: ((1.0 + 3.0*cu) + (4.5*cu)*cu) - 1.5*usq. Shared WITHOUT changing a bit:
    //  * where a component of c_i is 0 its product is an exact signed zero and the sum equals the other term bit for bit, except
    //    for the sign of a zero result — which the bracket erases (1 + 3*(+-0) = 1, 4.5*(+-0)^2 = +0);
    //  * opposite directions have cu of equal magnitude and opposite sign EXACTLY (negation is exact and rounding is symmetric:
    //    (-1)*ux + (-1)*uy == -(ux + uy)), hence 3.0*cu flips its sign exactly and (4.5*cu)*cu is the same number: one cu, one
    //    3*cu and one 4.5*cu*cu per PAIR, and 1.0 + (-t) is the subtraction 1.0 - t;
    //  * 1.5*usq and the three w*rho are common to all directions.
    // 112 instead of ~137 fp64 operations per cell (round 4); the populations stay np.array_equal to the oracle in every test.
    const T c15 = T(1.5) * usq;
    const T wr0 = wgt<T>(0) * rho, wr1 = wgt<T>(1) * rho, wr5 = wgt<T>(5) * rho;
    auto relax = [&](T& fi, T wr, T bracket) {
        const T feq = wr * bracket;
        fi = fi - tau_inv * (fi - feq);
    };
    relax(f[0], wr0, (T(1.0) + T(0.0)) - c15);      // cu = 0: 1.0 + 3.0*0 + 4.5*0*0 = 1.0 exactly
    auto pair = [&](int i, int ib, T cu, T wr) {     // i: the direction whose c.u is `cu`; ib: its opposite
        const T t3 = T(3.0) * cu, t45 = (T(4.5) * cu) * cu;
        relax(f[i], wr, ((T(1.0) + t3) + t45) - c15);
        relax(f[ib], wr, ((T(1.0) - t3) + t45) - c15);
    };
    pair(1, 3, ux, wr1);              // c1 = (1,0),  c3 = (-1,0)
    pair(2, 4, uy, wr1);              // c2 = (0,1),  c4 = (0,-1)
    pair(5, 7, ux + uy, wr5);         // c5 = (1,1),  c7 = (-1,-1)
    pair(8, 6, ux - uy, wr5);         // c8 = (1,-1), c6 = (-1,1): 1*ux + (-1)*uy == ux - uy
}

// Grid::check_stability on one cell's populations: NaN, Inf, > 1e5, < -1e5 (LBMGrid.h:296-307); 1e5 is exact in fp32 too,
// so the test is done in T. Filter first: bit 30 of an IEEE word (the top exponent bit) is set iff |f| >= 2 or f is
// Inf/NaN, so the OR of the nine (high) words has it clear iff every |f_i| < 2 — four v_or3_b32 and a test instead of nine
// compares; the exact test runs only for a wave that holds such a value (never, in a healthy flow). Same verdict, bit for bit.
__device__ __forceinline__ unsigned exp_word(double v) { return (unsigned)(__builtin_bit_cast(unsigned long long, v) >> 32); }
__device__ __forceinline__ unsigned exp_word(float v) { return __builtin_bit_cast(unsigned, v); }
template <typename T>
__device__ __forceinline__ bool any_unstable(const T (&f)[Q]) {
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < Q; ++i) o |= exp_word(f[i]);
    bool bad = false;
    if (o & 0x40000000u) {
#pragma unroll
        for (int i = 0; i < Q; ++i) bad |= !(fabs(f[i]) <= T(1e5));
    }
    return bad;
}

enum StepMode { MODE_STEP = 0, MODE_COLLIDE_ONLY = 1, MODE_STREAM_ONLY = 2 };

// One iteration per launch: one thread per lattice site, block = 256 sites of one row (4 waves), 8-byte (fp64) /
// 4-byte (fp32) coalesced plane accesses; the x±1 pulls are the same coalesced stream shifted by one element.
//   MODE_STEP         : K_t as described at the top of this file.
//   MODE_COLLIDE_ONLY : collision_step of iteration 0 on the initial state (no pull, no BC, no stability test).
//   MODE_STREAM_ONLY  : f_current snapshot for the accessor: pull + BCs + cylinder reversal
//                       (LBMSolver.h:240-257), every interior cell written, no collision.
template <typename T, int MODE, bool NT = false, int AR = AR_STRICT>
__global__ void __launch_bounds__(256) k_step_site(const KArgs<T> a) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = row_of_block(a);
    if (x >= a.nx) return;
    const int yg = a.y_start + y;
    const long c = (long)(y + GR) * a.pitch + a.xoff + x;
    T f[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const long off = (MODE == MODE_COLLIDE_ONLY) ? 0 : (long)cy(i) * a.pitch + cx(i);
        f[i] = a.src[(long)i * a.plane + c - off];
    }
    const bool solid = is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);
    if (MODE != MODE_COLLIDE_ONLY) {
        T rho_bc, u_out;
        if (!solid)
            apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
    }
    if (MODE == MODE_STEP) {
        if (any_unstable(f)) atomicMin(a.unstable_t, *a.t_base + a.t);
    }
    if (MODE == MODE_STREAM_ONLY) {
        if (solid) {
            T r[Q];
#pragma unroll
            for (int i = 0; i < Q; ++i) r[i] = f[opp(i)];
#pragma unroll
            for (int i = 0; i < Q; ++i) f[i] = r[i];
        }
    } else {
        if (solid) return;   // collision skips solid cells: they keep w_i for ever (LBMSolver.h:92, N4)
        bgk_collide<T, AR>(f, a.tau_inv);
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        T* p = a.dst + (long)i * a.plane + c;
        if (NT) __builtin_nontemporal_store(f[i], p);   // streaming store: do not keep the line in L2
        else *p = f[i];
    }
}

// (Rounds 1-3 also had k_step_vec, the same step with 16 bytes per lane — two fp64 / four fp32 sites per thread: 110 us per iteration
// at 4096x1024 fp64 against 112 for this kernel and 100 for this kernel with non-temporal stores; no measured plan ever took it. Retired
// in round 4 with its 14 instantiations; round 5 removed the option "variant" that used to select it.)

// Two timesteps per launch: temporal blocking through LDS. A block owns a TX x TY tile of outputs at iteration
// t+1. Phase 1 computes P_{t+1} on the (TX+2) x (TY+2) region around it from global P_t — the step kernel's
// per-cell sequence, with cells outside the domain taking their permanent ghost constants (N1/N2) and solid cells
// w_i — into LDS; after one barrier phase 2 pulls from LDS, applies the BCs of iteration t+1, collides and stores
// P_{t+2}. P_{t+1} never touches HBM: traffic per lattice update drops from 144 B to ~(1 + (TX+2)(TY+2)/(TX TY))*36 B
// (82 B at 64x8, less when the tile halo is still in L2 / Infinity Cache). Same arithmetic per cell => results
// bit-identical to two k_step_site launches (tests). Any nx (partial tiles at the right edge); rows of neighbouring strips must be
// present (at least) two deep. LDS: 9*(TY+2)*(TX+4)*sizeof(T) (47.9 KB at TY=8, fp64: three blocks per CU).
// feq_in: the nine initial-equilibrium values (permanent content of physical N/S ghost rows and corner ghosts), in
// device memory: they are needed by the few cells of a region that lie outside the domain only, and passing them by
// value would pin 18 scalar registers for the whole kernel (the fused kernels are SGPR-bound).
template <typename T> struct K2Extra {
    const T* feq_in;
    int small;   // the buffer is below 4 GiB: the lean path of k_step3_tile may address it with 32-bit byte offsets
    int xcd;     // remap the blocks so that every XCD walks one contiguous run of tiles (run-time: scalar index arithmetic only)
    int nt;      // non-temporal stores (run-time in the fused tile kernels: one block-uniform branch around the nine stores)
    int ntl;     // non-temporal LOADS at level 1 of the register kernel (run-time: one block-uniform branch around its 36 loads)
};

// the nine stores of one cell, plain or non-temporal (block-uniform choice)
template <typename T>
__device__ __forceinline__ void store_pops(T* base, long plane, const T (&f)[Q], bool nt) {
    if (nt) {
#pragma unroll
        for (int i = 0; i < Q; ++i) __builtin_nontemporal_store(f[i], base + (long)i * plane);
    } else {
#pragma unroll
        for (int i = 0; i < Q; ++i) base[(long)i * plane] = f[i];
    }
}
template <typename T>
__device__ __forceinline__ void buf_store_pops(const T (&f)[Q], __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned planeB, bool nt) {
    if (nt) {
#pragma unroll
        for (int i = 0; i < Q; ++i) buf_store<true>(f[i], r, voff, soff + (unsigned)i * planeB);
    } else {
#pragma unroll
        for (int i = 0; i < Q; ++i) buf_store<false>(f[i], r, voff, soff + (unsigned)i * planeB);
    }
}

template <typename T, int TY, int NTH, int AR = AR_STRICT>
__global__ void __launch_bounds__(NTH) k_step2_tile(const KArgs<T> a, const K2Extra<T> e) {
    constexpr int TX = 64, RW = TX + 2, RH = TY + 2, LP = RW + 2;
    __shared__ T lds[Q][RH][LP];
    // Tile of this block. Workgroups are dealt round-robin over the 8 XCDs (each with its own L2); with XCD the
    // linear block id is remapped so that every XCD walks one contiguous run of tiles: horizontally adjacent tiles,
    // which share the cache lines at their common edge, then run on the same L2 at the same time.
    int bx = blockIdx.x, by = blockIdx.y;
    if (e.xcd) {
        const int nb = gridDim.x * gridDim.y;
        int b = by * gridDim.x + bx;
        if (nb % 8 == 0) b = (b % 8) * (nb / 8) + b / 8;
        by = b / gridDim.x; bx = b - by * gridDim.x;
    }
    if (a.reverse) by = (int)gridDim.y - 1 - by;               // (the host never combines reverse with a second range)
    const int X0 = bx * TX;
    int y_end;                                                 // rows >= y_end belong to another band / launch
    const int Y0 = band_origin(a, by, TY, y_end);
    const bool near_cyl = tile_near_cylinder(a, X0, Y0, TX, TY, 1);   // block-uniform: most tiles skip the mask math
    bool bad = false;
    for (int r = threadIdx.x; r < RW * RH; r += NTH) {         // phase 1: iteration t on the region
        const int ry = r / RW, rx = r - ry * RW;
        const int x = X0 + rx - 1, y = Y0 + ry - 1;
        if (y > y_end) continue;                               // partial last tile: not needed by any output
        const int yg = a.y_start + y;
        const bool row_in = (yg >= 0 && yg < a.ny_glob), col_in = (x >= 0 && x < a.nx);
        T f[Q];
        if (!(row_in && col_in)) {
#pragma unroll
            for (int i = 0; i < Q; ++i) f[i] = (row_in && !col_in) ? T(0) : e.feq_in[i];
        } else {
            const long c = (long)(y + GR) * a.pitch + a.xoff + x;
#pragma unroll
            for (int i = 0; i < Q; ++i) f[i] = a.src[(long)i * a.plane + c - (long)cy(i) * a.pitch - cx(i)];
            bool solid = false;
            if (near_cyl) solid = is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);   // block-uniform branch
            T rho_bc, u_out;
            if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
            bad |= any_unstable(f);
            bgk_collide<T, AR>(f, a.tau_inv);
            if (near_cyl) {                    // solid cells keep w_i (the collision result of such a cell is discarded)
#pragma unroll
                for (int i = 0; i < Q; ++i) f[i] = solid ? wgt<T>(i) : f[i];
            }
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) lds[i][ry][rx] = f[i];
    }
    if (bad) atomicMin(a.unstable_t, *a.t_base + a.t);
    __syncthreads();
    bad = false;
    for (int o = threadIdx.x; o < TX * TY; o += NTH) {         // phase 2: iteration t+1 on the tile
        const int ly = o / TX, lx = o - ly * TX;
        const int x = X0 + lx, y = Y0 + ly;
        if (y >= y_end || x >= a.nx) continue;               // partial tiles at the right / top edge
        const int yg = a.y_start + y;
        T f[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) f[i] = lds[i][ly + 1 - cy(i)][lx + 1 - cx(i)];
        const bool solid = near_cyl && is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);
        T rho_bc, u_out;
        if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
        bad |= any_unstable(f);
        if (solid) continue;
        bgk_collide<T, AR>(f, a.tau_inv);
        const long c = (long)(y + GR) * a.pitch + a.xoff + x;
        store_pops(a.dst + c, a.plane, f, e.nt != 0);
    }
    if (bad) atomicMin(a.unstable_t, *a.t_base + a.t + 1);
}

// Three iterations per launch. Same idea one level deeper; the LDS image is reused in place:
//   phase 1  region 1 = tile + 2 rings: P_{t+1} from global P_t into LDS (9*(TY+4)*(TX+4)*sizeof(T): 78 KB at 64x12
//            fp64, two blocks of 1024 threads per CU = all 32 wave slots; the lean path needs 48 VGPRs);
//   phase 2  region 2 = tile + 1 ring: every thread first pulls its (<= 2) cells' nine values of P_{t+1} from LDS into
//            registers, barrier, then computes P_{t+2} and writes it IN PLACE (no second LDS image);
//   phase 3  the tile: pull P_{t+2} from LDS, BCs, collide, store P_{t+3}.
// HBM traffic per update ~ (1 + (TX+4)(TY+4)/(TX TY)) * 24 B (58 B at 64x12); redundant collisions 1.21x. Bit-identical to
// three single launches (tests). Rows of neighbouring strips must be present three deep beyond the rows written.
template <typename T, int TY, int NTH, int AR = AR_STRICT>
__global__ void __launch_bounds__(NTH, (2 * NTH / 256)) k_step3_tile(const KArgs<T> a, const K2Extra<T> e) {
    constexpr int TX = 64, R1W = TX + 4, R1H = TY + 4, R2W = TX + 2, R2H = TY + 2, LP = R1W;
    static_assert(R2W * R2H <= 2 * NTH, "two region-2 cells per thread at most");
    __shared__ T lds[Q][R1H][LP];
    int bx = blockIdx.x, by = blockIdx.y;
    if (e.xcd) {
        const int nb = gridDim.x * gridDim.y;
        int b = by * gridDim.x + bx;
        if (nb % 8 == 0) b = (b % 8) * (nb / 8) + b / 8;
        by = b / gridDim.x; bx = b - by * gridDim.x;
    }
    if (a.reverse) by = (int)gridDim.y - 1 - by;
    const int X0 = bx * TX;
    int y_end;
    const int Y0 = band_origin(a, by, TY, y_end);
    const bool near_cyl = tile_near_cylinder(a, X0, Y0, TX, TY, 2);
    auto outside_value = [&](bool row_in, bool col_in, int i) -> T { return (row_in && !col_in) ? T(0) : e.feq_in[i]; };
    // LEAN (block-uniform): the tile and its two rings lie strictly inside the domain, the tile is full, nothing is near
    // the cylinder — every cell of all three regions is a plain fluid cell: no boundary, ghost, solid or validity logic.
    const int yg0 = a.y_start + Y0;
    const bool lean = !near_cyl && X0 >= 3 && X0 + TX + 2 <= a.nx - 1 && yg0 >= 3 && yg0 + TY + 2 <= a.ny_glob - 1 &&
                      Y0 + TY <= y_end && e.small;
    // (lean path) one uniform base per buffer + a 32-bit byte offset per access: a vector add per plane instead of a 64-bit one
    const unsigned pitchB = (unsigned)a.pitch * (unsigned)sizeof(T), planeB = (unsigned)a.plane * (unsigned)sizeof(T);
    const unsigned KB = pitchB + (unsigned)sizeof(T);   // the source descriptor starts KB bytes early: every scalar offset stays >= 0
    const __amdgpu_buffer_rsrc_t rsrc = buf_desc(reinterpret_cast<const char*>(a.src) - KB), rdst = buf_desc(a.dst);
    auto run = [&]<bool LEAN>() {
        bool bad = false;
#pragma unroll
        for (int r = threadIdx.x; r < R1W * R1H; r += NTH) {                 // phase 1: iteration t
            const int ry = r / R1W, rx = r - ry * R1W;
            const int x = X0 + rx - 2, y = Y0 + ry - 2;
            const int yg = a.y_start + y;
            T f[Q];
            bool inside = true;
            if (!LEAN) {
                const bool row_in = (yg >= 0 && yg < a.ny_glob), col_in = (x >= 0 && x < a.nx);
                inside = row_in && col_in && y <= y_end + 1;
                if (!inside) {
#pragma unroll
                    for (int i = 0; i < Q; ++i) f[i] = outside_value(row_in, col_in, i);
                }
            }
            if (inside) {
                if (LEAN) {
                    const unsigned ub = (unsigned)(Y0 - 2 + GR) * pitchB + (unsigned)(a.xoff + X0 - 2) * (unsigned)sizeof(T) + KB;   // block-uniform
                    const unsigned voff = (unsigned)ry * pitchB + (unsigned)rx * (unsigned)sizeof(T);
#pragma unroll
                    for (int i = 0; i < Q; ++i)
                        f[i] = buf_load<T>(rsrc, voff, ub + (unsigned)i * planeB - (unsigned)cy(i) * pitchB - (unsigned)(cx(i) * (int)sizeof(T)));
                } else {
                    const long c = (long)(y + GR) * a.pitch + a.xoff + x;
#pragma unroll
                    for (int i = 0; i < Q; ++i) f[i] = a.src[(long)i * a.plane + c - (long)cy(i) * a.pitch - cx(i)];
                }
                if (LEAN) {
                    bad |= any_unstable(f);
                    bgk_collide<T, AR>(f, a.tau_inv);
                } else {
                    bool solid = false;
                    if (near_cyl) solid = is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);   // block-uniform branch
                    T rho_bc, u_out;
                    if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
                    bad |= any_unstable(f);
                    bgk_collide<T, AR>(f, a.tau_inv);
                    if (near_cyl) {            // solid cells keep w_i (the collision result of such a cell is discarded)
#pragma unroll
                        for (int i = 0; i < Q; ++i) f[i] = solid ? wgt<T>(i) : f[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < Q; ++i) lds[i][ry][rx] = f[i];
        }
        if (bad) atomicMin(a.unstable_t, *a.t_base + a.t);
        __syncthreads();
        // phase 2: iteration t+1 on region 2, in place
        T g[2][Q];
        int cell[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = threadIdx.x + k * NTH;
            cell[k] = (r < R2W * R2H) ? r : -1;
            if (cell[k] >= 0) {
                const int ry = r / R2W + 1, rx = r - (r / R2W) * R2W + 1;    // LDS coordinates of the cell
#pragma unroll
                for (int i = 0; i < Q; ++i) g[k][i] = lds[i][ry - cy(i)][rx - cx(i)];
            }
        }
        __syncthreads();
        bad = false;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (cell[k] < 0) continue;
            const int r = cell[k];
            const int ry = r / R2W + 1, rx = r - (r / R2W) * R2W + 1;
            const int x = X0 + rx - 2, y = Y0 + ry - 2;
            const int yg = a.y_start + y;
            T f[Q];
            bool inside = true;
            if (!LEAN) {
                const bool row_in = (yg >= 0 && yg < a.ny_glob), col_in = (x >= 0 && x < a.nx);
                inside = row_in && col_in;
                if (!inside) {
#pragma unroll
                    for (int i = 0; i < Q; ++i) f[i] = outside_value(row_in, col_in, i);
                }
            }
            if (inside) {
#pragma unroll
                for (int i = 0; i < Q; ++i) f[i] = g[k][i];
                if (LEAN) {
                    bad |= any_unstable(f);
                    bgk_collide<T, AR>(f, a.tau_inv);
                } else {
                    bool solid = false;
                    if (near_cyl) solid = is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);   // block-uniform branch
                    T rho_bc, u_out;
                    if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
                    if (y <= y_end) bad |= any_unstable(f);
                    bgk_collide<T, AR>(f, a.tau_inv);
                    if (near_cyl) {            // solid cells keep w_i (the collision result of such a cell is discarded)
#pragma unroll
                        for (int i = 0; i < Q; ++i) f[i] = solid ? wgt<T>(i) : f[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < Q; ++i) lds[i][ry][rx] = f[i];
        }
        if (bad) atomicMin(a.unstable_t, *a.t_base + a.t + 1);
        __syncthreads();
        bad = false;
        for (int o = threadIdx.x; o < TX * TY; o += NTH) {                    // phase 3: iteration t+2 on the tile
            const int ly = o / TX, lx = o - ly * TX;
            const int x = X0 + lx, y = Y0 + ly;
            if (!LEAN && (y >= y_end || x >= a.nx)) continue;
            const int yg = a.y_start + y;
            T f[Q];
#pragma unroll
            for (int i = 0; i < Q; ++i) f[i] = lds[i][ly + 2 - cy(i)][lx + 2 - cx(i)];
            if (LEAN) {
                bad |= any_unstable(f);
            } else {
                const bool solid = near_cyl && is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);
                T rho_bc, u_out;
                if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
                bad |= any_unstable(f);
                if (solid) continue;
            }
            bgk_collide<T, AR>(f, a.tau_inv);
            if (LEAN) {
                const unsigned ub = (unsigned)(Y0 + GR) * pitchB + (unsigned)(a.xoff + X0) * (unsigned)sizeof(T);
                const unsigned voff = (unsigned)ly * pitchB + (unsigned)lx * (unsigned)sizeof(T);
                buf_store_pops(f, rdst, voff, ub, planeB, e.nt != 0);
            } else {
                const long c = (long)(y + GR) * a.pitch + a.xoff + x;
                store_pops(a.dst + c, a.plane, f, e.nt != 0);
            }
        }
        if (bad) atomicMin(a.unstable_t, *a.t_base + a.t + 2);
    };
    if (lean) run.template operator()<true>();
    else run.template operator()<false>();
}

// Four iterations per launch: k_step3_tile one level deeper (two in-place levels). Region 1 = tile + 3 rings from HBM
// into LDS (9*(TY+6)*(TX+6)*sizeof(T): 70.5 KB at 64x8 fp64, 35 KB fp32), regions 2 and 3 in place, then the tile.
// HBM traffic per update ~ (1 + (TX+6)(TY+6)/(TX TY)) * 18 B; redundant collisions 1.45x at 64x8 — worth it where the
// three-iteration kernel is close to the memory roof (fp32). Needs four valid rows beyond the rows written, so strips
// (six rows per exchange = 2 x 3) never use it; the plan measurement decides elsewhere. Bit-identical to four single launches (tests).
template <typename T, int TY, int NTH, int AR = AR_STRICT>
__global__ void __launch_bounds__(NTH, (2 * NTH / 256)) k_step4_tile(const KArgs<T> a, const K2Extra<T> e) {
    constexpr int TX = 64, HW = 3, R1W = TX + 2 * HW, R1H = TY + 2 * HW, LP = R1W;
    static_assert((R1W - 2) * (R1H - 2) <= 2 * NTH, "two cells per thread at most in the in-place levels");
    __shared__ T lds[Q][R1H][LP];
    int bx = blockIdx.x, by = blockIdx.y;
    if (e.xcd) {
        const int nb = gridDim.x * gridDim.y;
        int b = by * gridDim.x + bx;
        if (nb % 8 == 0) b = (b % 8) * (nb / 8) + b / 8;
        by = b / gridDim.x; bx = b - by * gridDim.x;
    }
    if (a.reverse) by = (int)gridDim.y - 1 - by;
    const int X0 = bx * TX;
    int y_end;
    const int Y0 = band_origin(a, by, TY, y_end);
    const bool near_cyl = tile_near_cylinder(a, X0, Y0, TX, TY, HW);
    auto outside_value = [&](bool row_in, bool col_in, int i) -> T { return (row_in && !col_in) ? T(0) : e.feq_in[i]; };
    // LEAN (block-uniform): tile + three rings strictly inside the domain, full tile, nothing near the cylinder
    const int yg0 = a.y_start + Y0;
    const bool lean = !near_cyl && X0 >= HW + 1 && X0 + TX + HW <= a.nx - 1 && yg0 >= HW + 1 && yg0 + TY + HW <= a.ny_glob - 1 &&
                      Y0 + TY <= y_end && e.small;
    // one cell: BCs, stability, collision (solid cells keep w_i); `count` = the cell's instability is reported
    // (lean path) buffer descriptors: nine scalar offsets + one vector offset per cell (see buf_load)
    const unsigned pitchB = (unsigned)a.pitch * (unsigned)sizeof(T), planeB = (unsigned)a.plane * (unsigned)sizeof(T);
    const unsigned KB = pitchB + (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsrc = buf_desc(reinterpret_cast<const char*>(a.src) - KB), rdst = buf_desc(a.dst);
    auto update = [&](T (&f)[Q], int x, int yg, bool count, bool& bad) {
        if (lean) {                                 // block-uniform
            bad |= any_unstable(f);
            bgk_collide<T, AR>(f, a.tau_inv);
            return;
        }
        bool solid = false;
        if (near_cyl) solid = is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);   // block-uniform branch
        T rho_bc, u_out;
        if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
        if (count) bad |= any_unstable(f);
        bgk_collide<T, AR>(f, a.tau_inv);
        if (near_cyl) {
#pragma unroll
            for (int i = 0; i < Q; ++i) f[i] = solid ? wgt<T>(i) : f[i];
        }
    };
    bool bad = false;
#pragma unroll
    for (int r = threadIdx.x; r < R1W * R1H; r += NTH) {                 // level 1 on region 1: iteration t
        const int ry = r / R1W, rx = r - ry * R1W;
        const int x = X0 + rx - HW, y = Y0 + ry - HW;
        const int yg = a.y_start + y;
        const bool row_in = (yg >= 0 && yg < a.ny_glob), col_in = (x >= 0 && x < a.nx);
        T f[Q];
        if (lean) {                                                       // block-uniform
            const unsigned ub = (unsigned)(Y0 - HW + GR) * pitchB + (unsigned)(a.xoff + X0 - HW) * (unsigned)sizeof(T) + KB;
            const unsigned voff = (unsigned)ry * pitchB + (unsigned)rx * (unsigned)sizeof(T);
#pragma unroll
            for (int i = 0; i < Q; ++i)
                f[i] = buf_load<T>(rsrc, voff, ub + (unsigned)i * planeB - (unsigned)cy(i) * pitchB - (unsigned)(cx(i) * (int)sizeof(T)));
            update(f, x, yg, true, bad);
        } else if (!(row_in && col_in) || y > y_end + HW - 1) {
#pragma unroll
            for (int i = 0; i < Q; ++i) f[i] = outside_value(row_in, col_in, i);
        } else {
            const long c = (long)(y + GR) * a.pitch + a.xoff + x;
#pragma unroll
            for (int i = 0; i < Q; ++i) f[i] = a.src[(long)i * a.plane + c - (long)cy(i) * a.pitch - cx(i)];
            update(f, x, yg, true, bad);
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) lds[i][ry][rx] = f[i];
    }
    if (bad) atomicMin(a.unstable_t, *a.t_base + a.t);
    __syncthreads();
    // levels 2 and 3 on regions 2 and 3, in place: pull into registers, barrier, compute and overwrite, barrier
    auto in_place = [&]<int L>() {
        constexpr int O = L - 1, RW = R1W - 2 * O, RH = R1H - 2 * O;      // region L = region 1 shrunk by L-1 rings
        T g[2][Q];
        int cell[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = (int)threadIdx.x + k * NTH;
            cell[k] = (r < RW * RH) ? r : -1;
            if (cell[k] >= 0) {
                const int ry = r / RW + O, rx = r - (r / RW) * RW + O;    // LDS coordinates of the cell
#pragma unroll
                for (int i = 0; i < Q; ++i) g[k][i] = lds[i][ry - cy(i)][rx - cx(i)];
            }
        }
        __syncthreads();
        bool badl = false;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (cell[k] < 0) continue;
            const int r = cell[k];
            const int ry = r / RW + O, rx = r - (r / RW) * RW + O;
            const int x = X0 + rx - HW, y = Y0 + ry - HW;
            const int yg = a.y_start + y;
            const bool row_in = (yg >= 0 && yg < a.ny_glob), col_in = (x >= 0 && x < a.nx);
            T f[Q];
            if (!lean && !(row_in && col_in)) {
#pragma unroll
                for (int i = 0; i < Q; ++i) f[i] = outside_value(row_in, col_in, i);
            } else {
#pragma unroll
                for (int i = 0; i < Q; ++i) f[i] = g[k][i];
                update(f, x, yg, y <= y_end + HW - L, badl);
            }
#pragma unroll
            for (int i = 0; i < Q; ++i) lds[i][ry][rx] = f[i];
        }
        if (badl) atomicMin(a.unstable_t, *a.t_base + a.t + L - 1);
        __syncthreads();
    };
    in_place.template operator()<2>();
    in_place.template operator()<3>();
    bad = false;
    for (int o = threadIdx.x; o < TX * TY; o += NTH) {                    // level 4 on the tile: iteration t+3
        const int ly = o / TX, lx = o - ly * TX;
        const int x = X0 + lx, y = Y0 + ly;
        if (y >= y_end || x >= a.nx) continue;
        const int yg = a.y_start + y;
        T f[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) f[i] = lds[i][ly + HW - cy(i)][lx + HW - cx(i)];
        if (lean) {                                                       // block-uniform
            bad |= any_unstable(f);
            bgk_collide<T, AR>(f, a.tau_inv);
            const unsigned ub = (unsigned)(Y0 + GR) * pitchB + (unsigned)(a.xoff + X0) * (unsigned)sizeof(T);
            const unsigned voff = (unsigned)ly * pitchB + (unsigned)lx * (unsigned)sizeof(T);
            buf_store_pops(f, rdst, voff, ub, planeB, e.nt != 0);
            continue;
        }
        const bool solid = near_cyl && is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);
        T rho_bc, u_out;
        if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
        bad |= any_unstable(f);
        if (solid) continue;
        bgk_collide<T, AR>(f, a.tau_inv);
        const long c = (long)(y + GR) * a.pitch + a.xoff + x;
        store_pops(a.dst + c, a.plane, f, e.nt != 0);
    }
    if (bad) atomicMin(a.unstable_t, *a.t_base + a.t + 3);
}

// In-kernel phase record (tools/colbench -DLBM_COL_PROF only; never in the library): lane 0 of every wave writes the 100 MHz
// real-time counter at the marks below, plus HW_ID / XCC_ID, so that the host can lay the blocks of one CU side by side (profiles/r04).
#ifdef LBM_COL_PROF
constexpr int PROF_SLOTS = 96;
__device__ unsigned long long* lbm_prof_buf;     // [block][wave][PROF_SLOTS]
__device__ __forceinline__ void prof_mark(int blk, int nw, int w, int slot) {
    asm volatile("" ::: "memory");
    if (((int)threadIdx.x & 63) == 0 && slot < PROF_SLOTS) {
        unsigned long long t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");    // 100 MHz, chip-wide (s_memtime: per-CU offsets)
        lbm_prof_buf[((size_t)blk * nw + w) * PROF_SLOTS + slot] = t;
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void prof_ids(int blk, int nw, int w) {
    if (((int)threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
        unsigned long long rt;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt) :: "memory");
        lbm_prof_buf[((size_t)blk * nw + w) * PROF_SLOTS + PROF_SLOTS - 1] = ((unsigned long long)xcc << 32) | hw;
        lbm_prof_buf[((size_t)blk * nw + w) * PROF_SLOTS + PROF_SLOTS - 2] = rt;
    }
}
#define LBM_PROF(blk, nw, w, slot) prof_mark(blk, nw, w, slot)
#define LBM_PROF_IDS(blk, nw, w) prof_ids(blk, nw, w)
#else
#define LBM_PROF(blk, nw, w, slot) do {} while (0)
#define LBM_PROF_IDS(blk, nw, w) do {} while (0)
#endif

// D iterations per launch (D = 6..8) on a TX x TY tile: k_step4_tile generalised — level 1 from HBM into an LDS image of the
// (TX + 2(D-1)) x (TY + 2(D-1)) region, levels 2..D-1 in place (pull into registers, barrier, compute, overwrite, barrier),
// level D the tile -> HBM, on 64x16 / 32x32 tiles of 1024 threads: for grids so small that one launch is a single round of
// blocks. There the chip runs load -> compute -> store in lockstep, so the two memory phases are paid per LAUNCH: fusing more
// iterations divides them, at the price of ~1.5x redundant collisions that such a grid has VALU time to spare for.
// (Round 2's production shape — 32x16 tiles, 512 threads, D = 5/6, two blocks per CU: 125-130 GLUPS at 4096x1024 fp64 — is
// superseded by k_stepc_col, which keeps the same region in registers instead of LDS, and is no longer built.)
// Whole-domain launches only (rows outside the domain hold the permanent ghost values; a strip is refreshed six rows deep per single launch).
// waves per SIMD the register allocation may assume: as many blocks as the LDS image allows on a CU (at most 32 waves)
template <typename T, int TX, int TY, int D>
constexpr int deep_waves_per_simd() {
    const int lds = (int)sizeof(T) * Q * (TX + 2 * (D - 1)) * (TY + 2 * (D - 1)), waves = TX * TY / 64;
    int blocks = 160 * 1024 / lds;
    if (blocks * waves > 32) blocks = 32 / waves;
    return (blocks * waves) / 4 > 0 ? (blocks * waves) / 4 : 1;
}
template <typename T, int TX, int TY, int D, int AR = AR_STRICT>
__global__ void __launch_bounds__(TX * TY, (deep_waves_per_simd<T, TX, TY, D>())) k_stepd_tile(const KArgs<T> a, const K2Extra<T> e) {
    constexpr int NTH = TX * TY, HW = D - 1, R1W = TX + 2 * HW, R1H = TY + 2 * HW, LP = R1W;
    static_assert(D >= 3 && (NTH == 1024 || NTH == 512 || NTH == 256), "one tile cell per thread at the last level");
    static_assert((TX & (TX - 1)) == 0, "cell_xy splits an index with a mask");
    static_assert((size_t)Q * R1H * LP * sizeof(T) <= 160 * 1024, "level-1 region must fit the CU's LDS");
    __shared__ T lds[Q][R1H][LP];
    // Cell r of the region at ring offset O (rows and columns [O, R1 - O) of the LDS image) -> its LDS coordinates. The TX
    // columns above the tile come first, row by row: a half-wave then covers one aligned run of 32 cells (no LDS bank
    // conflict, whole cache lines at level 1) instead of a region row that wraps somewhere inside it; the 2 x (HW - O) halo
    // columns of every row follow.
    auto cell_xy = [&]<int O>(int r, int& ry, int& rx) {
        constexpr int RH = R1H - 2 * O, HALO = HW - O, NC = RH * TX;
        if (HALO == 0 || r < NC) { ry = O + r / TX; rx = HW + (r & (TX - 1)); }
        else {
            const int h = r - NC, q = h / (2 * HALO), c = h - q * (2 * HALO);
            ry = O + q;
            rx = c < HALO ? O + c : HW + TX + (c - HALO);
        }
    };
    int bx = blockIdx.x, by = blockIdx.y;
    if (e.xcd) {
        const int nb = gridDim.x * gridDim.y;
        int b = by * gridDim.x + bx;
        if (nb % 8 == 0) b = (b % 8) * (nb / 8) + b / 8;
        by = b / gridDim.x; bx = b - by * gridDim.x;
    }
    if (a.reverse) by = (int)gridDim.y - 1 - by;
    const int X0 = bx * TX;
    int y_end;
    const int Y0 = band_origin(a, by, TY, y_end);
    const bool near_cyl = tile_near_cylinder(a, X0, Y0, TX, TY, HW);
    auto outside_value = [&](bool row_in, bool col_in, int i) -> T { return (row_in && !col_in) ? T(0) : e.feq_in[i]; };
    const int yg0 = a.y_start + Y0;
    const bool lean = !near_cyl && X0 >= HW + 1 && X0 + TX + HW <= a.nx - 1 && yg0 >= HW + 1 && yg0 + TY + HW <= a.ny_glob - 1 &&
                      Y0 + TY <= y_end && e.small;
    // (lean path) buffer descriptors: nine scalar offsets + one vector offset per cell (see buf_load)
    const unsigned pitchB = (unsigned)a.pitch * (unsigned)sizeof(T), planeB = (unsigned)a.plane * (unsigned)sizeof(T);
    const unsigned KB = pitchB + (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsrc = buf_desc(reinterpret_cast<const char*>(a.src) - KB), rdst = buf_desc(a.dst);
    // one general cell: BCs, stability, collision (solid cells keep w_i); `count` = the cell's instability is reported
    auto update = [&](T (&f)[Q], int x, int yg, bool count, bool& bad) {
        bool solid = false;
        if (near_cyl) solid = is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);   // block-uniform branch
        T rho_bc, u_out;
        if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
        if (count) bad |= any_unstable(f);
        bgk_collide<T, AR>(f, a.tau_inv);
        if (near_cyl) {
#pragma unroll
            for (int i = 0; i < Q; ++i) f[i] = solid ? wgt<T>(i) : f[i];
        }
    };
    // The whole tile twice: LEAN = every cell of every level is a plain fluid cell (no boundary, ghost, solid or validity
    // logic, buffer addressing); a block takes one of the two (block-uniform), so neither pays for merging with the other.
    auto run = [&]<bool LEAN>() {
        bool bad = false;
        [[maybe_unused]] const int pb = (int)(blockIdx.y * gridDim.x + blockIdx.x), pw = (int)threadIdx.x >> 6;
        LBM_PROF_IDS(pb, NTH / 64, pw);
        LBM_PROF(pb, NTH / 64, pw, 0);
#pragma unroll
        for (int r = threadIdx.x; r < R1W * R1H; r += NTH) {                 // level 1 on region 1: iteration t
            int ry, rx;
            cell_xy.template operator()<0>(r, ry, rx);
            T f[Q];
            if (LEAN) {
                const unsigned ub = (unsigned)(Y0 - HW + GR) * pitchB + (unsigned)(a.xoff + X0 - HW) * (unsigned)sizeof(T) + KB;
                const unsigned voff = (unsigned)ry * pitchB + (unsigned)rx * (unsigned)sizeof(T);
#pragma unroll
                for (int i = 0; i < Q; ++i)
                    f[i] = buf_load<T>(rsrc, voff, ub + (unsigned)i * planeB - (unsigned)cy(i) * pitchB - (unsigned)(cx(i) * (int)sizeof(T)));
                bad |= any_unstable(f);
                bgk_collide<T, AR>(f, a.tau_inv);
            } else {
                const int x = X0 + rx - HW, y = Y0 + ry - HW;
                const int yg = a.y_start + y;
                const bool row_in = (yg >= 0 && yg < a.ny_glob), col_in = (x >= 0 && x < a.nx);
                if (!(row_in && col_in) || y > y_end + HW - 1) {
#pragma unroll
                    for (int i = 0; i < Q; ++i) f[i] = outside_value(row_in, col_in, i);
                } else {
                    const long c = (long)(y + GR) * a.pitch + a.xoff + x;
#pragma unroll
                    for (int i = 0; i < Q; ++i) f[i] = a.src[(long)i * a.plane + c - (long)cy(i) * a.pitch - cx(i)];
                    update(f, x, yg, true, bad);
                }
            }
#pragma unroll
            for (int i = 0; i < Q; ++i) lds[i][ry][rx] = f[i];
        }
        if (bad) atomicMin(a.unstable_t, *a.t_base + a.t);
        LBM_PROF(pb, NTH / 64, pw, 1);
        __syncthreads();
        auto in_place = [&]<int L>() {                                        // level L on region L = region 1 shrunk by L-1 rings
            constexpr int O = L - 1, RW = R1W - 2 * O, RH = R1H - 2 * O, CPT = (RW * RH + NTH - 1) / NTH;
            T g[CPT][Q];
            int cell[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int r = (int)threadIdx.x + k * NTH;
                cell[k] = (r < RW * RH) ? r : -1;
                if (cell[k] >= 0) {
                    int ry, rx;
                    cell_xy.template operator()<O>(r, ry, rx);
#pragma unroll
                    for (int i = 0; i < Q; ++i) g[k][i] = lds[i][ry - cy(i)][rx - cx(i)];
                }
            }
            __syncthreads();
            bool badl = false;
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                if (cell[k] < 0) continue;
                const int r = cell[k];
                int ry, rx;
                cell_xy.template operator()<O>(r, ry, rx);
                if (LEAN) {
                    badl |= any_unstable(g[k]);
                    bgk_collide<T, AR>(g[k], a.tau_inv);
                } else {
                    const int x = X0 + rx - HW, y = Y0 + ry - HW;
                    const int yg = a.y_start + y;
                    const bool row_in = (yg >= 0 && yg < a.ny_glob), col_in = (x >= 0 && x < a.nx);
                    if (!(row_in && col_in)) {
#pragma unroll
                        for (int i = 0; i < Q; ++i) g[k][i] = outside_value(row_in, col_in, i);
                    } else update(g[k], x, yg, y <= y_end + HW - L, badl);
                }
#pragma unroll
                for (int i = 0; i < Q; ++i) lds[i][ry][rx] = g[k][i];
            }
            if (badl) atomicMin(a.unstable_t, *a.t_base + a.t + L - 1);
            __syncthreads();
            LBM_PROF(pb, NTH / 64, pw, L);
        };
        [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (in_place.template operator()<Ls + 2>(), ...); }(std::make_integer_sequence<int, D - 2>{});
        bad = false;
        {                                                                     // level D on the tile: iteration t+D-1
            const int o = threadIdx.x;
            const int ly = o / TX, lx = o - ly * TX;
            const int x = X0 + lx, y = Y0 + ly;
            if (LEAN || (y < y_end && x < a.nx)) {
                const int yg = a.y_start + y;
                T f[Q];
#pragma unroll
                for (int i = 0; i < Q; ++i) f[i] = lds[i][ly + HW - cy(i)][lx + HW - cx(i)];
                if (LEAN) {
                    bad |= any_unstable(f);
                    bgk_collide<T, AR>(f, a.tau_inv);
                    const unsigned ub = (unsigned)(Y0 + GR) * pitchB + (unsigned)(a.xoff + X0) * (unsigned)sizeof(T);
                    const unsigned voff = (unsigned)ly * pitchB + (unsigned)lx * (unsigned)sizeof(T);
                    buf_store_pops(f, rdst, voff, ub, planeB, e.nt != 0);
                } else {
                    const bool solid = near_cyl && is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);
                    T rho_bc, u_out;
                    if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
                    bad |= any_unstable(f);
                    if (!solid) {
                        bgk_collide<T, AR>(f, a.tau_inv);
                        const long c = (long)(y + GR) * a.pitch + a.xoff + x;
                        store_pops(a.dst + c, a.plane, f, e.nt != 0);
                    }
                }
            }
        }
        if (bad) atomicMin(a.unstable_t, *a.t_base + a.t + D - 1);
        LBM_PROF(pb, NTH / 64, pw, D);                  // stores issued
#ifdef LBM_COL_PROF
        __builtin_amdgcn_s_waitcnt(0x0f70);             // vmcnt(0): ... and drained
        LBM_PROF(pb, NTH / 64, pw, D + 1);
#endif
    };
    if (lean) run.template operator()<true>();
    else run.template operator()<false>();
}

// ---------------------------------------------------------------------------------------------------------
// Initialisation: Grid::initialise (LBMGrid.h:185-246) written into BOTH buffers, plus the permanent ghost
// values of N1/N2 (see top of file). feq_in = f_eq(1,(u_in,0)) evaluated on the host in double with the
// reference's bracket order (LBMUtils.h:9-12,46); solid cells get f_eq(1,0,0) = w_i.
template <typename T>
struct InitArgs {
    T* a; T* b;
    long plane; int pitch, xoff, nx, ny_loc, ny_glob, y_start;
    int cyl_x, cyl_y; double cyl_r2;
    T feq_in[Q];
    int* solid_count;
};

template <typename T>
__global__ void __launch_bounds__(256) k_init(const InitArgs<T> p) {
    const int gx = blockIdx.x * 256 + threadIdx.x;   // 0 .. nx+1  (ghost-inclusive column)
    const int gy = blockIdx.y;                       // 0 .. ny_loc+2*GR-1
    if (gx > p.nx + 1) return;
    const int x = gx - 1, yg = p.y_start + gy - GR;  // global coordinates (may lie outside the domain)
    const bool row_interior = (yg >= 0 && yg < p.ny_glob);
    const bool col_interior = (x >= 0 && x < p.nx);
    const bool own_row = (gy >= GR && gy < p.ny_loc + GR);
    bool solid = false;
    if (row_interior && col_interior) solid = is_solid_cell(x, yg, p.cyl_x, p.cyl_y, p.cyl_r2);
    if (solid && own_row) atomicAdd(p.solid_count, 1);
    const long c = (long)gy * p.pitch + p.xoff + x;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        T v = p.feq_in[i];
        if (solid) v = wgt<T>(i);
        if (row_interior && !col_interior) v = T(0);   // N1: E/W ghost column of a globally-interior row
        p.a[(long)i * p.plane + c] = v;
        p.b[(long)i * p.plane + c] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Macroscopic snapshot from the buffer `old` = P_t that the last step kernel READ:
//   fluid interior : moments of P_t at the cell. collision conserves rho and rho*u, so these equal the
//                    pre-collision moments the reference stored at LBMSolver.h:112-114 to round-off (~1e-16).
//   inlet/outlet   : (rho_bc, u_in, 0) / (1, u_out, 0) of apply_boundary_conditions of iteration t,
//                    recomputed exactly as the step kernel did (pull from P_t + wall BC + Zou-He).
//   solid          : (1, 0, 0) (rho never rewritten after init, u zeroed at LBMSolver.h:260-261).
template <typename T>
struct MacroArgs {
    const T* old; long plane; int pitch, xoff, nx, ny_loc, ny_glob, y_start;
    int cyl_x, cyl_y; double cyl_r2; T u_in;
    int initial;        // steps_done == 0: analytic initial macros (LBMGrid.h:219-228)
    double* rho; double* ux; double* uy;   // [ny_loc][nx]
    unsigned long long* max_usq_bits;      // optional running max of ux^2+uy^2 (bit pattern of a double >= 0)
};

template <typename T>
__global__ void __launch_bounds__(256) k_macros(const MacroArgs<T> p) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    double usq = 0.0;
    if (x < p.nx) {
        const int yg = p.y_start + y;
        const long c = (long)(y + GR) * p.pitch + p.xoff + x;
        const bool solid = is_solid_cell(x, yg, p.cyl_x, p.cyl_y, p.cyl_r2);
        double r, vx, vy;
        if (solid) { r = 1.0; vx = 0.0; vy = 0.0; }
        else if (p.initial) { r = 1.0; vx = (double)p.u_in; vy = 0.0; }
        else if (x == 0 || x == p.nx - 1) {
            T f[Q];
#pragma unroll
            for (int i = 0; i < Q; ++i) f[i] = p.old[(long)i * p.plane + c - (long)cy(i) * p.pitch - cx(i)];
            T rho_bc = T(1), u_out = T(0);
            apply_bcs(f, yg == 0, yg == p.ny_glob - 1, x == 0, x == p.nx - 1, p.u_in, rho_bc, u_out);
            // the outlet loop runs after the inlet loop (only matters for nx == 1)
            if (x == p.nx - 1) { r = 1.0; vx = (double)u_out; vy = 0.0; }
            else { r = (double)rho_bc; vx = (double)p.u_in; vy = 0.0; }
        } else {
            T rr = T(0), sx = T(0), sy = T(0);
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                const T v = p.old[(long)i * p.plane + c];
                rr += v;
                if (cx(i) != 0) sx += T(cx(i)) * v;
                if (cy(i) != 0) sy += T(cy(i)) * v;
            }
            sx /= rr; sy /= rr;
            r = (double)rr; vx = (double)sx; vy = (double)sy;
        }
        const long m = (long)y * p.nx + x;
        p.rho[m] = r; p.ux[m] = vx; p.uy[m] = vy;
        usq = vx * vx + vy * vy;
    }
    if (p.max_usq_bits) {
        // wave max, then one atomic per wave; non-negative doubles order like their bit patterns
        for (int o = 32; o > 0; o >>= 1) usq = fmax(usq, __shfl_xor(usq, o));
        if ((threadIdx.x & 63) == 0) atomicMax(p.max_usq_bits, (unsigned long long)__double_as_longlong(usq));
    }
}

// ---------------------------------------------------------------------------------------------------------
// IOManager::record_forces, LBMIO.h:133-160: momentum exchange over solid->fluid links, evaluated on the
// post-collision populations P_t. One block scans the cylinder's bounding box (+1 cell) restricted to the rows
// this strip owns, visiting FLUID cells and their solid neighbours (mask from global coordinates), so strip
// partial sums add up to the one-rank value. Deterministic tree reduction in LDS.
template <typename T>
struct ForceArgs {
    const T* cur; long plane; int pitch, xoff, nx, ny_loc, ny_glob, y_start;
    int cyl_x, cyl_y, cyl_r; double cyl_r2;
    int x0, x1, y0, y1;     // inclusive box in (x, local y)
    double* out;            // out[0] = t (as double), out[1] = fx, out[2] = fy
    int t;
};

template <typename T>
__global__ void __launch_bounds__(1024) k_forces(const ForceArgs<T> p) {
    __shared__ double sfx[1024];
    __shared__ double sfy[1024];
    double fx = 0.0, fy = 0.0;
    const int bw = p.x1 - p.x0 + 1, bh = p.y1 - p.y0 + 1;
    const long ncell = (bw > 0 && bh > 0) ? (long)bw * bh : 0;
    for (long k = threadIdx.x; k < ncell; k += 1024) {
        const int x = p.x0 + (int)(k % bw), y = p.y0 + (int)(k / bw);
        const int yg = p.y_start + y;
        if (is_solid_cell(x, yg, p.cyl_x, p.cyl_y, p.cyl_r2)) continue;
        const long c = (long)(y + GR) * p.pitch + p.xoff + x;
#pragma unroll
        for (int i = 1; i < Q; ++i) {
            const int sx = x + cx(i), sy = yg + cy(i);
            if (sx < 0 || sx >= p.nx || sy < 0 || sy >= p.ny_glob) continue;
            if (!is_solid_cell(sx, sy, p.cyl_x, p.cyl_y, p.cyl_r2)) continue;
            const double fi = (double)p.cur[(long)i * p.plane + c];
            fx += 2.0 * cx(i) * fi;
            fy += 2.0 * cy(i) * fi;
        }
    }
    sfx[threadIdx.x] = fx; sfy[threadIdx.x] = fy;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sfx[threadIdx.x] += sfx[threadIdx.x + s]; sfy[threadIdx.x] += sfy[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { p.out[0] = (double)p.t; p.out[1] = sfx[0]; p.out[2] = sfy[0]; }
}

// ---------------------------------------------------------------------------------------------------------
// Host-staged halo rows (lbm_halo_export / lbm_halo_import): GR rows x 9 planes x nx interior columns per face,
// as double, [GR][9][nx]. `row0` = first local gy of the GR consecutive rows.
template <typename T>
__global__ void __launch_bounds__(256) k_halo_pack(const T* buf, long plane, int pitch, int xoff, int nx, int row0,
                                                   double* out) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;          // 0 .. GR*Q-1  = row-major (row, plane)
    if (x >= nx) return;
    const int r = k / Q, i = k - r * Q;
    out[(long)k * nx + x] = (double)buf[(long)i * plane + (long)(row0 + r) * pitch + xoff + x];
}
template <typename T>
__global__ void __launch_bounds__(256) k_halo_unpack(T* buf, long plane, int pitch, int xoff, int nx, int row0,
                                                     const double* in) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (x >= nx) return;
    const int r = k / Q, i = k - r * Q;
    buf[(long)i * plane + (long)(row0 + r) * pitch + xoff + x] = (T)in[(long)k * nx + x];
}

}  // namespace lbmk
