// csrc/lbm_col_api.hpp — what the host translation unit (lbm_hip.hip) sees of the register-resident column kernel: its region
// shape and one launcher per element type. The kernel itself (lbm_kernel_col.hpp) is compiled in its own translation unit
// (lbm_col.hip) so that the two halves of the device code build side by side (build.py).
#pragma once
#include "lbm_kernels.hpp"

namespace lbmk {

// Waves per block and rows per thread (the region is 64 columns x waves * rows): 8 waves x 4 rows = 64 x 32, two blocks per CU —
// except fp64 STRICT arithmetic, whose collision needs a dozen more live registers than 128 VGPRs leave beside 4 x 9 fp64
// populations. Round 4 first ran it on 8 waves x 3 rows (64 x 24, 122 VGPRs, no scratch: 106-114 -> 127-130 GLUPS at 4096x1024; round
// 3's 4-row kernels spilled 12 VGPRs), then on the same 64 x 24 region as 12 waves x 2 rows: 70 VGPRs = six waves per SIMD = two blocks
// of 12 waves = 24 waves per CU instead of 16: 135.7 against 127.3 GLUPS (tools/colbench --strict). Contracted arithmetic is fastest
// on 8 x 4 (64 x 24 on 24 waves: 157-159 against 158-162; 8 x 3: 146-150).
constexpr int col_waves(int esize, bool strict, bool tall = false) {
    return (esize == 8 && strict && !tall) || (esize == 4 && tall && !strict) ? 12 : 8;
}
// TALL (fp32 only, round 4): a 64 x 48 region — 12 waves x 4 rows in contracted arithmetic (76 VGPRs = six waves per SIMD = two
// blocks = 24 waves per CU, as many as three standard blocks), 8 waves x 6 rows in strict arithmetic (112-116 VGPRs). At seven
// iterations it stores 52 x 36 of its 64 x 48 cells where the 64 x 32 region stores 54 x 22 at six: 1.33 x instead of 1.45 x the
// lattice read per launch. 16384x4096 fp32 (tools/colbench): 320 GLUPS at seven iterations, 315 at six, against 289-298 on 64 x 32;
// 4096x1024 262 against 260; a 16384x512 strip 259 against 246-260. The first tall shape of the round, 8 waves x 8 rows = 64 x 64
// (121-127 VGPRs, 12 B of scratch), read 310 at 16384x4096 and lost elsewhere (216 at 4096x1024, 209 on the strip): replaced.
constexpr int col_rows_per_thread(int esize, bool strict, bool tall = false) {
    return tall && esize == 4 ? (strict ? 6 : 4) : (esize == 8 && strict) ? 2 : 4;
}
// output tile of a launch of `depth` iterations
constexpr int col_tile_w(int depth) { return 64 - 2 * (depth - 1); }
constexpr int col_tile_h(int depth, int rows_per_thread, int waves) { return rows_per_thread * waves - 2 * (depth - 1); }
constexpr int col_tile_h(int depth, int esize, bool strict, bool tall) {
    return col_tile_h(depth, col_rows_per_thread(esize, strict, tall), col_waves(esize, strict, tall));
}

// k_stepc_col<T, rows per thread, waves, depth, nt, arith> over the rows a.y_lo.. / a.y_lo2.. of the launch: depth 5, 6 or 7 on
// 64 x 32 regions (one object file per element type: lbm_col.hip -DLBM_COL_T=double / float)
template <typename T>
void launch_col(const KArgs<T>& a, const K2Extra<T>& e, int depth, bool nt, bool contracted, hipStream_t s);
// ... and on the tall fp32 regions, depth 6, 7 or 8, plain stores only (non-temporal ones cost 14 % there); one object file per
// arithmetic mode (lbm_col.hip -DLBM_COL_TALL=1 contracted / 0 strict: eight unrolled rows x up to eight levels compile slowly)
void launch_col_tall_contracted(const KArgs<float>& a, const K2Extra<float>& e, int depth, hipStream_t s);
void launch_col_tall_strict(const KArgs<float>& a, const K2Extra<float>& e, int depth, hipStream_t s);

}  // namespace lbmk
