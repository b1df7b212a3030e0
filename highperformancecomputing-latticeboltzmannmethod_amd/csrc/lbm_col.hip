// csrc/lbm_col.hip — the translation unit(s) of k_stepc_col (lbm_kernel_col.hpp): its 30 instantiations (five / six / seven
// iterations x store policy x arithmetic x element type, twelve per element type and object file, plus the six tall fp32 ones) and the launcher lbm_hip.hip calls (lbm_col_api.hpp).
#include "lbm_kernel_col.hpp"
#include "lbm_col_api.hpp"

namespace lbmk {

template <typename T>
void launch_col(const KArgs<T>& a, const K2Extra<T>& e, int depth, bool nt, bool contracted, bool tall, hipStream_t s) {
#define LBM_KC(D_, R_, NT_, AR_) do { \
        constexpr int OW_ = col_tile_w(D_), OH_ = col_tile_h(D_, R_); \
        const int nb_ = ((a.nx + OW_ - 1) / OW_) * ((a.y_cnt + OH_ - 1) / OH_ + (a.y_cnt2 + OH_ - 1) / OH_); \
        const dim3 gridc((unsigned)((nb_ + 7) / 8 * 8)), blockc(COL_NW * 64); \
        hipLaunchKernelGGL((k_stepc_col<T, R_, COL_NW, D_, NT_, AR_>), gridc, blockc, 0, s, a, e); } while (0)
#define LBM_KD(D_) do { \
        if (contracted) { if (nt) LBM_KC(D_, RC, true, AR_CONTRACTED); else LBM_KC(D_, RC, false, AR_CONTRACTED); } \
        else { if (nt) LBM_KC(D_, RS, true, AR_STRICT); else LBM_KC(D_, RS, false, AR_STRICT); } } while (0)
    constexpr int RC = col_rows_per_thread((int)sizeof(T), false), RS = col_rows_per_thread((int)sizeof(T), true);
    if constexpr (sizeof(T) == 4) {
        if (tall) {     // 64 x 64 (strict: 64 x 48) regions, plain stores
            constexpr int TC = col_rows_per_thread(4, false, true), TS = col_rows_per_thread(4, true, true);
#define LBM_KT(D_) do { if (contracted) LBM_KC(D_, TC, false, AR_CONTRACTED); else LBM_KC(D_, TS, false, AR_STRICT); } while (0)
            if (depth == 6) LBM_KT(6); else if (depth == 8) LBM_KT(8); else LBM_KT(7);
#undef LBM_KT
            return;
        }
    }
    if (depth == 5) LBM_KD(5); else if (depth == 7) LBM_KD(7); else LBM_KD(6);
#undef LBM_KD
#undef LBM_KC
}

// compiled once per element type (build.py: -DLBM_COL_T=double / float), the two halves side by side
#ifndef LBM_COL_T
#error "compile with -DLBM_COL_T=double or -DLBM_COL_T=float"
#endif
template void launch_col<LBM_COL_T>(const KArgs<LBM_COL_T>&, const K2Extra<LBM_COL_T>&, int, bool, bool, bool, hipStream_t);

}  // namespace lbmk
