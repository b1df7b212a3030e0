// csrc/lbm_kernel_col.hpp — D iterations per launch with the lattice held in REGISTERS (round 3's production kernel).
//
// The LDS-image kernels (k_stepd_tile) move every population of every cell through LDS at every level: nine ds_write and
// nine ds_read per cell and level, with a thread <-> cell map that changes from level to level, two barriers per level,
// and an LDS image that caps the tile at ~1000 cells per block. On gfx950 that LDS traffic costs about as much as the
// collision arithmetic and the two never overlap inside a block (profiles/r02). This kernel keeps the map FIXED instead:
//
//   block  = NW waves; it owns a region of 64 columns x H = R*NW rows of the lattice for all D levels
//   wave w = rows [w*R, (w+1)*R) of the region, lane = column: every thread holds R vertically adjacent cells, i.e.
//            R x 9 populations, in VGPRs from the first load to the last store
//   level 1  pulls P_t from HBM for all cells of the region (nine displaced coalesced row loads per row)
//   level l  needs, per cell, the level l-1 populations of its eight neighbours:
//              same thread, row above/below  -> a register (free)
//              x -/+ 1                        -> DPP wave shift of the neighbour lane's register (v_mov_b32_dpp wave_shr/shl:1)
//              row above/below the thread's R -> the neighbouring WAVE: three populations per face through LDS
//            so per wave and level 6 ds_write + 6 ds_read in all (1.5 + 1.5 per cell at R = 4, against 9 + 9) and ONE
//            barrier (the exchange buffer is double-buffered); the rest population never leaves its register.
//   level D  stores the (64 - 2(D-1)) x (H - 2(D-1)) cells in the middle of the region: P_{t+D}
// Cells at distance < l-1 from the region's edge hold garbage at level l (their neighbours were never loaded); nothing
// valid ever reads them — the garbage moves inward one ring per level, exactly as the valid region shrinks — and the
// stability verdict is masked to the valid cells of each level. Same per-cell operation sequence as every other step
// kernel => bit-identical results (tests).
// Redundant collisions: x 64/56, y (H + .. + H-8)/(5 (H-8)) at D = 5: 1.33x at H = 32. HBM traffic per update at D = 5,
// H = 32: (2048 + 1344) * 72 B / (1344 * 5) = 36 B before L2 absorbs the overlap, like the 32x16 LDS tile.
#pragma once
#include "lbm_kernels.hpp"

namespace lbmk {

// lane i <- lane i-1 (lane 0 reads 0) / lane i <- lane i+1 (lane 63 reads 0): full-wave DPP shifts. bound_ctrl:0 with every
// row and bank enabled leaves no lane that keeps an "old" value, so the instruction has no tied input and the compiler
// needs no copy in front of it (with old = src it emitted one v_mov per shifted dword: 38 of ~390 vector instructions a level).
__device__ __forceinline__ unsigned dpp_shr1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned dpp_shl1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ double from_left(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = dpp_shr1((unsigned)u), hi = dpp_shr1((unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double from_right(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = dpp_shl1((unsigned)u), hi = dpp_shl1((unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float from_left(float v) { return __builtin_bit_cast(float, dpp_shr1(__builtin_bit_cast(unsigned, v))); }
__device__ __forceinline__ float from_right(float v) { return __builtin_bit_cast(float, dpp_shl1(__builtin_bit_cast(unsigned, v))); }

// (in-kernel phase record, tools/colbench -DLBM_COL_PROF only: LBM_PROF / LBM_PROF_IDS, lbm_kernels.hpp)
// waves per SIMD the register allocation may assume: two blocks per CU when they fit 32 waves (8 waves: four per SIMD, 128 VGPRs;
// 12 waves: six per SIMD, 80 VGPRs — the two-row fp64 strict shape), else one
template <int NW> constexpr int col_waves_per_simd() { return NW == 12 ? 6 : NW <= 8 ? (2 * NW) / 4 : NW / 4; }

// stability verdict of one cell, counted only where `valid` (garbage cells may hold anything, NaN included)
template <typename T>
__device__ __forceinline__ bool unstable_if(const T (&f)[Q], bool valid) {
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < Q; ++i) o |= exp_word(f[i]);
    o = valid ? o : 0u;
    bool bad = false;
    if (o & 0x40000000u) {
#pragma unroll
        for (int i = 0; i < Q; ++i) bad |= !(fabs(f[i]) <= T(1e5));
        bad = bad && valid;
    }
    return bad;
}

// Tile walk: blocks are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2) in index order; here every XCD walks
// one contiguous run of the row-major tile order instead, so that x-neighbours — which share the lines at their common
// edge — meet in the same L2 at the same time and every XCD streams whole lattice rows (149.6 -> 161.7 GLUPS at 4096x1024
// fp64). A band-major walk (8..37 tile columns per band, so that y-neighbours meet in L2 too) was measured and is gone:
// 141-147 GLUPS — a tile then touches 300 sub-rows that its XCD's other tiles do not share, and the TLB / DRAM-page
// locality of the row-interleaved layout is lost; walking 2..4 tile rows together column by column (y-neighbours adjacent in
// the order) gains nothing at 4096x1024 and loses 12-25 % at 8192x2048 (profiles/r03/README.md).
// The grid is one-dimensional: cdiv(nx, OW) * (bands of both row ranges) blocks, rounded up to a multiple of 8 — one block per
// tile. PERSISTENT blocks (two per CU walking the tiles of their XCD's run in a loop, so that a wave which has stored its last
// rows issues its next tile's loads at once) were built and measured: 123 GLUPS against 142 for the same binary launched
// one block per tile — blocks that start together and take equally long stay phase-locked, and the two blocks of a CU then
// load together and compute together instead of filling each other's gaps, which the dispatcher's staggered hand-out gives
// for free; the loop also costs registers (every loop-invariant scalar offset wants an SGPR for the whole kernel: 106 SGPRs,
// > 100 spilled, 164 -> 142 GLUPS even when launched one block per tile). Gone.
// Round 5 measured two more forms of this kernel and dropped both (profiles/r05/README.md §2-3; the code is archived as a patch in
// profiles/r05/archive/): the waves of a block meeting through NEIGHBOUR FLAGS in LDS instead of one barrier per level (bit-exact; a
// wave's first wait shrinks, but a CU's slot frees only with a block's last wave: residency 1.63 -> 1.47 blocks per CU, -5 %), and the
// x-shifts through the LDS CROSSBAR (ds_bpermute_b32) instead of DPP moves (bit-exact; -1..3 %: the crossbar's latency sits in every
// row's dependency chain and vector issue is not the binding resource at the margin).
template <typename T, int R, int NW, int D, bool NT, int AR = AR_STRICT>
__global__ void __launch_bounds__(NW * 64, (col_waves_per_simd<NW>())) k_stepc_col(const KArgs<T> a, const K2Extra<T> e) {
    constexpr int H = R * NW, HW = D - 1, OW = 64 - 2 * HW, OH = H - 2 * HW, LW = 64 + 2;
    static_assert(D >= 2 && OH >= 1 && R >= 2, "(D <= GR on a strip: its ghost rows go GR deep — the host's business)");
    // exchange buffer: per wave the three north-going populations of its top row and the three south-going ones of its
    // bottom row; the first / last wave, which have no neighbour below / above, read their OWN slot instead (garbage for cells
    // that are garbage anyway: round 4 dropped the two phantom slots, 63 -> 51 KB at 8 fp64 waves, so that 12 waves fit two
    // blocks per CU); one pad column on each side for the diagonal reads at lane -/+ 1
    __shared__ T xbuf[2][NW][6][LW];
    const int lane = (int)threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nbx = (a.nx + OW - 1) / OW, nby = (a.y_cnt + OH - 1) / OH + (a.y_cnt2 + OH - 1) / OH, nb = nbx * nby;
    int b = blockIdx.x;
    { const int per = (int)gridDim.x >> 3; b = (b & 7) * per + (b >> 3); }                   // gridDim.x is a multiple of 8
    if (b >= nb) return;                                                                      // (whole block: no barrier is left waiting)
    int by = b / nbx;
    const int bx = b - by * nbx;
    if (a.reverse) by = nby - 1 - by;
    int y_end;
    const int Yo = band_origin(a, by, OH, y_end);          // first output row / column of this block
    const int Xo = bx * OW;
    const int X0 = Xo - HW, Yr = Yo - HW;                  // region origin
    const int ry0 = w * R;                                 // this wave's first region row
    const bool near_cyl = tile_near_cylinder(a, Xo, Yo, OW, OH, HW);
    auto outside_value = [&](bool row_in, bool col_in, int i) -> T { return (row_in && !col_in) ? T(0) : e.feq_in[i]; };
    const int yg0 = a.y_start + Yo;
    // LEAN (block-uniform): every cell of the region is a plain fluid cell strictly inside the domain and the tile is full
    const bool lean = !near_cyl && Xo >= HW + 1 && Xo + OW + HW <= a.nx - 1 && yg0 >= HW + 1 && yg0 + OH + HW <= a.ny_glob - 1 &&
                      Yo + OH <= y_end && e.small;
    const unsigned pitchB = (unsigned)a.pitch * (unsigned)sizeof(T), planeB = (unsigned)a.plane * (unsigned)sizeof(T);
    const unsigned KB = pitchB + (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsrc = buf_desc(reinterpret_cast<const char*>(a.src) - KB), rdst = buf_desc(a.dst);
    const int x = X0 + lane;
    // one general cell: BCs, stability, collision (solid cells keep w_i); `count` = the cell's instability is reported
    auto update = [&](T (&f)[Q], int yg, bool count, bool& bad) {
        bool solid = false;
        if (near_cyl) solid = is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2);   // block-uniform branch
        T rho_bc, u_out;
        if (!solid) apply_bcs(f, yg == 0, yg == a.ny_glob - 1, x == 0, x == a.nx - 1, a.u_in, rho_bc, u_out);
        bad |= unstable_if(f, count);
        bgk_collide<T, AR>(f, a.tau_inv);
        if (near_cyl) {
#pragma unroll
            for (int i = 0; i < Q; ++i) f[i] = solid ? wgt<T>(i) : f[i];
        }
    };
    auto run = [&]<bool LEAN>() {
        T g[R][Q];
        bool bad = false;
        const bool col_in = (x >= 0 && x < a.nx);
        LBM_PROF_IDS(b, NW, w);
        LBM_PROF(b, NW, w, 0);
        // ---- level 1: iteration t on the whole region, from HBM
        if (LEAN) {
            const unsigned ub = (unsigned)(Yr + ry0 + GR) * pitchB + (unsigned)(a.xoff + X0) * (unsigned)sizeof(T) + KB;   // wave-uniform
            const unsigned voff = (unsigned)lane * (unsigned)sizeof(T);
            auto load_all = [&]<int AUX>() {
#pragma unroll
                for (int j = 0; j < R; ++j)
#pragma unroll
                    for (int i = 0; i < Q; ++i)
                        g[j][i] = buf_load<T, AUX>(rsrc, voff, ub + (unsigned)j * pitchB + (unsigned)i * planeB - (unsigned)cy(i) * pitchB - (unsigned)(cx(i) * (int)sizeof(T)));
            };
            if (e.ntl) load_all.template operator()<2>(); else load_all.template operator()<0>();      // (block-uniform)
#pragma unroll
            for (int j = 0; j < R; ++j) {
                bad |= any_unstable(g[j]);
                bgk_collide<T, AR>(g[j], a.tau_inv);
            }
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int y = Yr + ry0 + j, yg = a.y_start + y;
                const bool row_in = (yg >= 0 && yg < a.ny_glob);
                if (!(row_in && col_in) || y > y_end + HW - 1) {
#pragma unroll
                    for (int i = 0; i < Q; ++i) g[j][i] = outside_value(row_in, col_in, i);
                } else {
                    const long c = (long)(y + GR) * a.pitch + a.xoff + x;
#pragma unroll
                    for (int i = 0; i < Q; ++i) g[j][i] = a.src[(long)i * a.plane + c - (long)cy(i) * a.pitch - cx(i)];
                    update(g[j], yg, true, bad);
                }
            }
        }
        if (bad) atomicMin(a.unstable_t, *a.t_base + a.t);
        LBM_PROF(b, NW, w, 1);
        // ---- levels 2..D: exchange with the neighbouring waves / lanes, then update in place (level D: store)
        auto level = [&]<int L>() {
            T (*xb)[6][LW] = xbuf[L & 1];
            const int wb = w > 0 ? w - 1 : 0, wa = w + 1 < NW ? w + 1 : NW - 1;
            xb[w][0][1 + lane] = g[R - 1][2]; xb[w][1][1 + lane] = g[R - 1][5]; xb[w][2][1 + lane] = g[R - 1][6];
            xb[w][3][1 + lane] = g[0][4];     xb[w][4][1 + lane] = g[0][7];     xb[w][5][1 + lane] = g[0][8];
            __syncthreads();
            // from the wave below (its top row): f2 at x, f5 at x-1, f6 at x+1; from the wave above (its bottom row): f4, f7 at x+1, f8 at x-1
            T p2 = xb[wb][0][1 + lane], p5 = xb[wb][1][lane], p6 = xb[wb][2][2 + lane];
            bool badl = false;
            const bool lane_ok = lane >= L - 1 && lane <= 64 - L;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int ry = ry0 + j;
                const T n2 = g[j][2], n5 = g[j][5], n6 = g[j][6];       // this row's north-going values, for row j+1
                if (ry >= L - 1 && ry <= H - L) {                       // (wave-uniform) rows outside hold garbage from here on
                T f[Q];
                f[0] = g[j][0]; f[1] = from_left(g[j][1]); f[3] = from_right(g[j][3]);
                f[2] = p2; f[5] = p5; f[6] = p6;
                if (j < R - 1) { f[4] = g[j + 1][4]; f[7] = from_right(g[j + 1][7]); f[8] = from_left(g[j + 1][8]); }
                else { f[4] = xb[wa][3][1 + lane]; f[7] = xb[wa][4][2 + lane]; f[8] = xb[wa][5][lane]; }   // (read here, not after the barrier: six registers fewer are live across the rows below; the buffer is not rewritten before this wave has passed the next barrier)
                const bool valid = lane_ok;
                const int y = Yr + ry, yg = a.y_start + y;
                bool store = L == D && lane >= HW && lane < 64 - HW && ry >= HW && ry < H - HW;
                if (LEAN) {
                    badl |= unstable_if(f, valid);
                    bgk_collide<T, AR>(f, a.tau_inv);
                } else {
                    const bool row_in = (yg >= 0 && yg < a.ny_glob);
                    if (!(row_in && col_in)) {
#pragma unroll
                        for (int i = 0; i < Q; ++i) f[i] = outside_value(row_in, col_in, i);
                        store = false;
                    } else {
                        update(f, yg, valid && y <= y_end + HW - L, badl);
                        store = store && y < y_end && !(near_cyl && is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2));
                    }
                }
                if (L < D) {
#pragma unroll
                    for (int i = 0; i < Q; ++i) g[j][i] = f[i];
                } else if (store) {
                    if (LEAN) {
                        const unsigned ub = (unsigned)(y + GR) * pitchB + (unsigned)(a.xoff + X0) * (unsigned)sizeof(T);
                        const unsigned voff = (unsigned)lane * (unsigned)sizeof(T);
#pragma unroll
                        for (int i = 0; i < Q; ++i) buf_store<NT>(f[i], rdst, voff, ub + (unsigned)i * planeB);
                    } else {
                        const long c = (long)(y + GR) * a.pitch + a.xoff + x;
#pragma unroll
                        for (int i = 0; i < Q; ++i) {
                            T* p = a.dst + (long)i * a.plane + c;
                            if (NT) __builtin_nontemporal_store(f[i], p); else *p = f[i];
                        }
                    }
                }
                }
                if (j < R - 1) { p2 = n2; p5 = from_left(n5); p6 = from_right(n6); }   // row j+1 pulls them from this row
            }
            if (badl) atomicMin(a.unstable_t, *a.t_base + a.t + L - 1);
            LBM_PROF(b, NW, w, L);
        };
        [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (level.template operator()<Ls + 2>(), ...); }(std::make_integer_sequence<int, D - 1>{});
    };
    if (lean) run.template operator()<true>();
    else run.template operator()<false>();
}

}  // namespace lbmk
