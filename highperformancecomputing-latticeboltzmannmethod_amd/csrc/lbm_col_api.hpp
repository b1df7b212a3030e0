// csrc/lbm_col_api.hpp — what the host translation unit (lbm_hip.hip) sees of the register-resident column kernel: its region
// shape and one launcher per element type. The kernel itself (lbm_kernel_col.hpp) is compiled in its own translation unit
// (lbm_col.hip) so that the two halves of the device code build side by side (build.py).
#pragma once
#include "lbm_kernels.hpp"

namespace lbmk {

constexpr int COL_R = 4, COL_NW = 8;      // rows per thread x waves per block: a 64 x 32 region, two blocks per CU
// output tile of a launch of `depth` iterations
constexpr int col_tile_w(int depth) { return 64 - 2 * (depth - 1); }
constexpr int col_tile_h(int depth) { return COL_R * COL_NW - 2 * (depth - 1); }

// k_stepc_col<T, COL_R, COL_NW, depth, nt, arith> over the rows a.y_lo.. / a.y_lo2.. of the launch (depth 5, 6 or — whole
// domains only: a strip's ghost rows go six deep — 7)
template <typename T>
void launch_col(const KArgs<T>& a, const K2Extra<T>& e, int depth, bool nt, bool contracted, hipStream_t s);

}  // namespace lbmk
