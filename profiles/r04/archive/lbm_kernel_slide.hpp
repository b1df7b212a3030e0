// tools/experimental/lbm_kernel_slide.hpp (EXPERIMENT of round 4, not part of the library: measured slower than k_stepc_col, profiles/r04/README.md) — D iterations per launch with the lattice in registers AND no overlap between tiles in y
// (round 4): the sliding, time-skewed form of k_stepc_col (lbm_kernel_col.hpp).
//
// k_stepc_col gives every block a 64 x 32 region and lets the valid part shrink by one ring per level: a launch of six
// iterations stores 54 x 22 of the 64 x 32 cells it loaded, i.e. it reads the lattice 1.48 x and collides 1.45 x as many cells
// as it updates — all of it y-overlap (the x-overlap of neighbouring regions meets in L2). Here a block owns a COLUMN STRIP of
// 64 lanes (54 output columns at D = 6) and walks UP a segment of it tile by tile (H = R*NW rows per tile), with the levels
// skewed by one row each:
//
//     tile k, level l (l = 1..D) covers the rows  [B_k - (l-1), B_k - (l-1) + H),   B_k = Ya + (D-1) + k*H
//
// so level l+1 of row y needs level l of rows y-1, y, y+1 = tile-local rows ry-2, ry-1, ry of the SAME tile: dependencies
// point downwards only. The top rows of a tile need nothing from above; its two bottom rows need the two top rows of the
// PREVIOUS tile at the same level — nine values per lane and level, left behind in LDS (`carry`). Every row of the segment is
// loaded once, collided once per level and stored once: the only redundancy left is the x-halo (64/54) and a warm-up triangle
// of 2(D-l) rows per level below the first tile of a segment (tile k = -1: its top rows only, nothing stored).
//
//   thread = lane (column) x R rows, wave w = tile rows [w*R, (w+1)*R), as in k_stepc_col; a thread's rows move DOWN by one
//            lattice row per level (it "follows the skew"), so that
//   new row j  <-  old row j   : f4, f7 (x+1), f8 (x-1)
//                  old row j-1 : f0, f1 (x-1), f3 (x+1)
//                  old row j-2 : f2, f5 (x-1), f6 (x+1)          x -/+ 1 by DPP wave shifts
//   rows are updated in place in DESCENDING order (new row j overwrites old row j, which only rows j+1, j+2 — already done —
//   still needed): no copies. Rows j >= 2 need the thread's own registers only; rows 1 and 0 need the two top rows of the wave
//   below (wave 0: of the previous tile): published through LDS at the start of a step, read after the step's first barrier.
//   Two barriers per step (the exchange buffer is single: 160 KB of LDS hold two blocks' 17 slots of 9 x 64 fp64 values), ten
//   per tile; every wave of a block has the same work at every level (no rows fall out of a tile), so they arrive together.
//   Level 1 pulls P_t from HBM with the nine displaced row loads of every other step kernel (rows [B_k, B_k + H)); level D
//   stores rows [B_k - (D-1), ...) — the tile's outputs trail its inputs by D-1 rows.
// Blocks are PERSISTENT (one segment each, all resident at once: two per CU): the slowest block is the launch. So the work is
// dealt by the host (SlideSeg table, slide_plan in lbm_col_api.hpp): LEAN segments — interior column strips away from the
// cylinder; walls are a wave-uniform swap there and rows outside the domain are simply not computed — run a loop of their own
// whose tiles prefetch: during the last step of a tile, as soon as a row's registers are dead (its level-D values are on
// their way to HBM), the next tile's level-1 loads of that row go out. GENERAL segments (the inlet and outlet strips, the rows
// of the strips that cross the cylinder's box) run the per-cell boundary logic of every other fused kernel and get shorter
// pieces. A lean block that meets the cylinder or a domain edge column (a host-side planning error) poisons the stability word
// (LBM_SLIDE_PLAN_ERROR) instead of computing garbage.
// Same per-cell operation sequence as every other step kernel => bit-identical results (tools/colbench, tests).
#pragma once
#include "../../highperformancecomputing-latticeboltzmannmethod_amd/csrc/lbm_kernel_col.hpp"

namespace lbmk {

struct SlideSeg { int bx, ya, yb, general; };   // column strip, local rows [ya, yb), path
struct SlideArgs {
    const SlideSeg* segs;   // one per block (device memory), in the order the XCD-aware walk should meet them
    int nblocks;
    int* cu_ticket;         // 2048 words, zeroed once: a never-reset counter per CU (XCC, SE, SH, CU id) — see `stagger`
    int stagger;            // > 0: of the two blocks a CU holds, the one that drew the odd ticket starts `stagger` x 64 x 127 cycles late
};
constexpr int LBM_SLIDE_PLAN_ERROR = -0x7ffffff0;   // what *unstable_t reads after a planning error (no iteration number is negative)

template <typename T, int R, int NW, int D, bool NT, int AR = AR_STRICT>
__global__ void __launch_bounds__(NW * 64, (col_waves_per_simd<NW>())) k_steps_col(const KArgs<T> a, const K2Extra<T> e, const SlideArgs sa) {
    constexpr int H = R * NW, HW = D - 1, OW = 64 - 2 * HW;
    static_assert(D >= 2 && R >= 3 && H >= 2 * (D - 1) && OW >= 1, "rows j >= 2 read the thread's own rows; the warm-up triangle fits one tile");
    // slot = nine values per lane: the top row's f0,f1,f3,f2,f5,f6 and the row below it's f2,f5,f6
    __shared__ T xb[NW - 1][9][64];          // wave w -> wave w+1, within a step
    __shared__ T carry[2][D - 1][9][64];     // top wave of tile k -> wave 0 of tile k+1, per level (parity of k: tile k writes its
                                             // level-l slot half a step BEFORE it reads tile k-1's)
    const int lane = (int)threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int b = blockIdx.x;
    { const int per = (int)gridDim.x >> 3; b = (b & 7) * per + (b >> 3); }                   // gridDim.x is a multiple of 8
    if (b >= sa.nblocks) return;
    const SlideSeg sg = sa.segs[b];
    const int Ya = sg.ya, Yb = sg.yb;
    if (Ya >= Yb) return;
    const int nt = (Yb - Ya + H - 1) / H;
    const int Xo = sg.bx * OW, X0 = Xo - HW;
    const int x = X0 + lane;
    const int ry0 = w * R;
    const unsigned pitchB = (unsigned)a.pitch * (unsigned)sizeof(T), planeB = (unsigned)a.plane * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsrc = buf_desc(reinterpret_cast<const char*>(a.src) - (pitchB + (unsigned)sizeof(T))), rdst = buf_desc(a.dst);
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(T);
    LBM_PROF_IDS(b, NW, w);
    int mark = 0;
    (void)mark;
    // Blocks that start together and do the same work stay in phase: the two blocks of a CU then load together, compute
    // together and store together, and the store phase (issue-bound per CU) takes each of them twice as long — for the whole
    // launch (profiles/r04: a third of the blocks took 220 us where the median took 160). So one of every CU's two blocks
    // starts half a tile late: they draw consecutive tickets from their CU's counter, the odd one sleeps.
    if (sa.stagger > 0) {
        __shared__ int late;
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
            const unsigned key = ((((xcc & 7u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u + ((hw >> 8) & 15u)) & 2047u;
            late = atomicAdd(sa.cu_ticket + key, 1) & 1;
        }
        __syncthreads();
        if (late) for (int i = 0; i < sa.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // the LDS side of a step, shared by both paths. publish: the level-L values the wave above (the next tile) pulls from this
    // wave's two top rows; pulled rows: see the header
    auto publish = [&](const T (&g)[R][Q], int k, int L) {
        T (*slot)[64] = (w == NW - 1) ? (carry[k & 1][0] + (L - 1) * 9) : xb[w];
        slot[0][lane] = g[R - 1][0]; slot[1][lane] = g[R - 1][1]; slot[2][lane] = g[R - 1][3];
        slot[3][lane] = g[R - 1][2]; slot[4][lane] = g[R - 1][5]; slot[5][lane] = g[R - 1][6];
        slot[6][lane] = g[R - 2][2]; slot[7][lane] = g[R - 2][5]; slot[8][lane] = g[R - 2][6];
    };
    auto pull_own = [&](T (&f)[Q], const T (&g)[R][Q], int j) {          // j >= 2
        f[4] = g[j][4]; f[7] = from_right(g[j][7]); f[8] = from_left(g[j][8]);
        f[0] = g[j - 1][0]; f[1] = from_left(g[j - 1][1]); f[3] = from_right(g[j - 1][3]);
        f[2] = g[j - 2][2]; f[5] = from_left(g[j - 2][5]); f[6] = from_right(g[j - 2][6]);
    };
    auto pull_row1 = [&](T (&f)[Q], const T (&g)[R][Q], const T (*from)[64]) {
        f[4] = g[1][4]; f[7] = from_right(g[1][7]); f[8] = from_left(g[1][8]);
        f[0] = g[0][0]; f[1] = from_left(g[0][1]); f[3] = from_right(g[0][3]);
        f[2] = from[3][lane]; f[5] = from_left(from[4][lane]); f[6] = from_right(from[5][lane]);
    };
    auto pull_row0 = [&](T (&f)[Q], const T (&g)[R][Q], const T (*from)[64]) {
        f[4] = g[0][4]; f[7] = from_right(g[0][7]); f[8] = from_left(g[0][8]);
        f[0] = from[0][lane]; f[1] = from_left(from[1][lane]); f[3] = from_right(from[2][lane]);
        f[2] = from[6][lane]; f[5] = from_left(from[7][lane]); f[6] = from_right(from[8][lane]);
    };
    auto below = [&](int k, int L) -> const T (*)[64] { return (w == 0) ? (carry[(k & 1) ^ 1][0] + (L - 1) * 9) : xb[w - 1]; };
    // (wave-uniform) is level l of tile row ry0 + j of tile k needed? warm-up tile (k = -1): its top 2(D-l) rows only; above the
    // segment's cone: no; LEAN: rows outside the domain are never computed (a wall row's pulls from them are overwritten by its swap)
    auto needed = [&]<bool LEAN>(int k, int l, int j) {
        const int ry = ry0 + j, y = Ya + HW + k * H - (l - 1) + ry, yg = a.y_start + y;
        return !(k < 0 && ry < H - 2 * (D - l)) && y < Yb + (D - l) && (!LEAN || (yg >= 0 && yg < a.ny_glob));
    };

    if (!sg.general) {
        // ================= LEAN segment: plain fluid cells and wall rows of an interior column strip, 32-bit offsets =================
        if (!(Xo >= HW + 1 && Xo + OW + HW <= a.nx - 1 && e.small)) { if (threadIdx.x == 0) *a.unstable_t = LBM_SLIDE_PLAN_ERROR; return; }
        T g[R][Q];
        auto zero_row = [&](int j) {     // rows that are not needed hold zeros (any DEFINED value)
#pragma unroll
            for (int i = 0; i < Q; ++i) g[j][i] = T(0);
        };
        // the nine displaced row loads of tile row ry0 + j of tile k. (The strides pass through an empty asm at every use: the
        // 36 + 9 scalar offsets would otherwise be hoisted out of the loop as invariants and live — i.e. spill — for the whole kernel.)
        auto load_row = [&](int k, int j) {
            unsigned pB = pitchB, plB = planeB;
            asm volatile("" : "+s"(pB), "+s"(plB));
            const unsigned ub = (unsigned)(Ya + HW + k * H + ry0 + GR) * pB + (unsigned)(a.xoff + X0) * (unsigned)sizeof(T) + pB + (unsigned)sizeof(T);   // wave-uniform
#pragma unroll
            for (int i = 0; i < Q; ++i)
                g[j][i] = buf_load<T>(rsrc, voff, ub + (unsigned)j * pB + (unsigned)i * plB - (unsigned)cy(i) * pB - (unsigned)(cx(i) * (int)sizeof(T)));
        };
        auto walls = [&](T (&f)[Q], int yg) {      // LBMSolver.h:153-176 on an interior column: wave-uniform
            if (yg == 0) { f[2] = f[4]; f[5] = f[7]; f[6] = f[8]; }
            if (yg == a.ny_glob - 1) { f[4] = f[2]; f[7] = f[5]; f[8] = f[6]; }
        };
#pragma unroll
        for (int j = R - 1; j >= 0; --j) { if (needed.template operator()<true>(-1, 1, j)) load_row(-1, j); else zero_row(j); }
        for (int k = -1; k < nt; ++k) {
            const int B = Ya + HW + k * H;                       // first level-1 row of the tile
            if (tile_near_cylinder(a, Xo, B, OW, H, HW)) { if (threadIdx.x == 0) *a.unstable_t = LBM_SLIDE_PLAN_ERROR; }
            bool bad = false;
            LBM_PROF(b, NW, w, mark++);
            // ---- level 1: iteration t on rows [B, B + H), loaded during the previous tile's last step
#pragma unroll
            for (int j = R - 1; j >= 0; --j) {
                if (!needed.template operator()<true>(k, 1, j)) continue;
                walls(g[j], a.y_start + B + ry0 + j);
                bad |= any_unstable(g[j]);
                bgk_collide<T, AR>(g[j], a.tau_inv);
            }
            if (bad) atomicMin(a.unstable_t, *a.t_base + a.t);
            // every load of this tile has landed on the path that used it; say so for the paths that skipped rows too, or the
            // compiler's wait-count bookkeeping, merged over the loop, drains the memory pipe (vmcnt(0)) in front of every
            // prefetch of the last step, stores included. What may still be in flight here are the previous tile's last stores.
            __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0) only (gfx9 encoding: expcnt 7, lgkmcnt 15)
            LBM_PROF(b, NW, w, mark++);
            // ---- steps 1..D-1: step L computes level L+1 (iteration t+L) of rows [B - L, B - L + H) from level L
            auto step = [&]<int L>() {
                publish(g, k, L);
                bool badl = false;
                const bool lane_ok = lane >= L && lane <= 63 - L;
                auto finish = [&](T (&f)[Q], int j) {
                    const int y = B - L + ry0 + j;
                    walls(f, a.y_start + y);
                    badl |= unstable_if(f, lane_ok);
                    bgk_collide<T, AR>(f, a.tau_inv);
                    if (L + 1 < D) {
#pragma unroll
                        for (int i = 0; i < Q; ++i) g[j][i] = f[i];
                    } else if (lane >= HW && lane < 64 - HW && y < Yb && k >= 0) {
                        unsigned pB = pitchB, plB = planeB;
                        asm volatile("" : "+s"(pB), "+s"(plB));
                        const unsigned ub = (unsigned)(y + GR) * pB + (unsigned)(a.xoff + X0) * (unsigned)sizeof(T);
#pragma unroll
                        for (int i = 0; i < Q; ++i) {
#if defined(LBM_SLIDE_EXP) && LBM_SLIDE_EXP == 1
                            if (i & 1) continue;         // TIMING EXPERIMENT (wrong results): half the store instructions
#elif defined(LBM_SLIDE_EXP) && LBM_SLIDE_EXP == 2
                            if (i > 0) continue;         // TIMING EXPERIMENT: one store instruction per row
#endif
                            buf_store<NT>(f[i], rdst, voff, ub + (unsigned)i * plB);
                        }
                    }
                };
                // last step: row j's registers are dead once rows j, j+1, j+2 are done — the next tile's level-1 loads of that row
                // go out right behind this row's stores
                auto prefetch = [&](int j) {
                    if (L + 1 == D) { if (k + 1 < nt && needed.template operator()<true>(k + 1, 1, j)) load_row(k + 1, j); else zero_row(j); }
                };
#pragma unroll
                for (int j = R - 1; j >= 2; --j) {
                    if (needed.template operator()<true>(k, L + 1, j)) { T f[Q]; pull_own(f, g, j); finish(f, j); }
                    prefetch(j);
                }
                __syncthreads();
                const T (*from)[64] = below(k, L);
                if (needed.template operator()<true>(k, L + 1, 1)) { T f[Q]; pull_row1(f, g, from); finish(f, 1); }
                prefetch(1);
                if (needed.template operator()<true>(k, L + 1, 0)) { T f[Q]; pull_row0(f, g, from); finish(f, 0); }
                prefetch(0);
                if (badl) atomicMin(a.unstable_t, *a.t_base + a.t + L);
                __syncthreads();
                LBM_PROF(b, NW, w, mark++);
            };
            [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (step.template operator()<Ls + 1>(), ...); }(std::make_integer_sequence<int, D - 1>{});
        }
        return;
    }

    // ================= GENERAL segment: every boundary of the lattice, cell by cell (as in k_stepc_col's general path) =================
    const bool col_in = (x >= 0 && x < a.nx);
    auto outside_value = [&](bool row_in, bool cin, int i) -> T { return (row_in && !cin) ? T(0) : e.feq_in[i]; };
    for (int k = -1; k < nt; ++k) {
        const int B = Ya + HW + k * H;
        const bool near_cyl = tile_near_cylinder(a, Xo, B, OW, H, HW);
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
        T g[R][Q];
        bool bad = false;
        LBM_PROF(b, NW, w, mark++);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int y = B + ry0 + j, yg = a.y_start + y;
            if (!needed.template operator()<false>(k, 1, j)) {
#pragma unroll
                for (int i = 0; i < Q; ++i) g[j][i] = T(0);
                continue;
            }
            const bool row_in = (yg >= 0 && yg < a.ny_glob);
            if (!(row_in && col_in)) {
#pragma unroll
                for (int i = 0; i < Q; ++i) g[j][i] = outside_value(row_in, col_in, i);
            } else {
                const long c = (long)(y + GR) * a.pitch + a.xoff + x;
#pragma unroll
                for (int i = 0; i < Q; ++i) g[j][i] = a.src[(long)i * a.plane + c - (long)cy(i) * a.pitch - cx(i)];
                update(g[j], yg, true, bad);
            }
        }
        if (bad) atomicMin(a.unstable_t, *a.t_base + a.t);
        LBM_PROF(b, NW, w, mark++);
        auto step = [&]<int L>() {
            publish(g, k, L);
            bool badl = false;
            const bool lane_ok = lane >= L && lane <= 63 - L;
            auto finish = [&](T (&f)[Q], int j) {
                const int y = B - L + ry0 + j, yg = a.y_start + y;
                bool store = (L + 1 == D) && lane >= HW && lane < 64 - HW && y < Yb && k >= 0;
                const bool row_in = (yg >= 0 && yg < a.ny_glob);
                if (!(row_in && col_in)) {
#pragma unroll
                    for (int i = 0; i < Q; ++i) f[i] = outside_value(row_in, col_in, i);
                    store = false;
                } else {
                    update(f, yg, lane_ok, badl);
                    store = store && !(near_cyl && is_solid_cell(x, yg, a.cyl_x, a.cyl_y, a.cyl_r2));
                }
                if (L + 1 < D) {
#pragma unroll
                    for (int i = 0; i < Q; ++i) g[j][i] = f[i];
                } else if (store) {
                    const long c = (long)(y + GR) * a.pitch + a.xoff + x;
#pragma unroll
                    for (int i = 0; i < Q; ++i) {
                        T* p = a.dst + (long)i * a.plane + c;
                        if (NT) __builtin_nontemporal_store(f[i], p); else *p = f[i];
                    }
                }
            };
#pragma unroll
            for (int j = R - 1; j >= 2; --j)
                if (needed.template operator()<false>(k, L + 1, j)) { T f[Q]; pull_own(f, g, j); finish(f, j); }
            __syncthreads();
            const T (*from)[64] = below(k, L);
            if (needed.template operator()<false>(k, L + 1, 1)) { T f[Q]; pull_row1(f, g, from); finish(f, 1); }
            if (needed.template operator()<false>(k, L + 1, 0)) { T f[Q]; pull_row0(f, g, from); finish(f, 0); }
            if (badl) atomicMin(a.unstable_t, *a.t_base + a.t + L);
            __syncthreads();
            LBM_PROF(b, NW, w, mark++);
        };
        [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (step.template operator()<Ls + 1>(), ...); }(std::make_integer_sequence<int, D - 1>{});
    }
}

}  // namespace lbmk
