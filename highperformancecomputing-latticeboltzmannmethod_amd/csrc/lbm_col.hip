// csrc/lbm_col.hip — the translation unit(s) of k_stepc_col (lbm_kernel_col.hpp): its 26 instantiations (five / six iterations x
// store policy + seven iterations, x arithmetic x element type: ten per element type and object file, plus the tall fp32 ones, three per
// arithmetic mode and object file) and the launchers lbm_hip.hip calls (lbm_col_api.hpp). Four objects from this one source (build.py).
#include "lbm_kernel_col.hpp"
#include "lbm_col_api.hpp"

namespace lbmk {

#define LBM_KC(T_, D_, R_, NW_, NT_, AR_) do { \
        constexpr int OW_ = col_tile_w(D_), OH_ = col_tile_h(D_, R_, NW_); \
        const int nb_ = ((a.nx + OW_ - 1) / OW_) * ((a.y_cnt + OH_ - 1) / OH_ + (a.y_cnt2 + OH_ - 1) / OH_); \
        const dim3 gridc((unsigned)((nb_ + 7) / 8 * 8)), blockc(NW_ * 64); \
        hipLaunchKernelGGL((k_stepc_col<T_, R_, NW_, D_, NT_, AR_>), gridc, blockc, 0, s, a, e); } while (0)

#if defined(LBM_COL_T)       // the 64 x 32 (fp64 strict: 64 x 24 on twelve waves) regions of one element type
template <typename T>
void launch_col(const KArgs<T>& a, const K2Extra<T>& e, int depth, bool nt, bool contracted, hipStream_t s) {
#define LBM_KD(D_) do { \
        if (contracted) { if (nt) LBM_KC(T, D_, RC, WC, true, AR_CONTRACTED); else LBM_KC(T, D_, RC, WC, false, AR_CONTRACTED); } \
        else { if (nt) LBM_KC(T, D_, RS, WS, true, AR_STRICT); else LBM_KC(T, D_, RS, WS, false, AR_STRICT); } } while (0)
    constexpr int RC = col_rows_per_thread((int)sizeof(T), false), RS = col_rows_per_thread((int)sizeof(T), true);
    constexpr int WC = col_waves((int)sizeof(T), false), WS = col_waves((int)sizeof(T), true);
    // (seven iterations: plain stores only — the depth of a call's remainders and of the "deep" 9 plans, whose candidates all store plainly)
    if (depth == 5) LBM_KD(5);
    else if (depth == 7) { if (contracted) LBM_KC(T, 7, RC, WC, false, AR_CONTRACTED); else LBM_KC(T, 7, RS, WS, false, AR_STRICT); }
    else LBM_KD(6);
#undef LBM_KD
}
template void launch_col<LBM_COL_T>(const KArgs<LBM_COL_T>&, const K2Extra<LBM_COL_T>&, int, bool, bool, hipStream_t);
#elif defined(LBM_COL_TALL)  // the tall fp32 regions (64 x 48) of one arithmetic mode: contracted 12 waves x 4 rows (1), strict 8 x 6 (0)
#if LBM_COL_TALL
void launch_col_tall_contracted(const KArgs<float>& a, const K2Extra<float>& e, int depth, hipStream_t s) {
    constexpr int R = col_rows_per_thread(4, false, true), W = col_waves(4, false, true), AR = AR_CONTRACTED;
#else
void launch_col_tall_strict(const KArgs<float>& a, const K2Extra<float>& e, int depth, hipStream_t s) {
    constexpr int R = col_rows_per_thread(4, true, true), W = col_waves(4, true, true), AR = AR_STRICT;
#endif
    if (depth == 6) LBM_KC(float, 6, R, W, false, AR); else if (depth == 8) LBM_KC(float, 8, R, W, false, AR); else LBM_KC(float, 7, R, W, false, AR);
}
#else
#error "compile with -DLBM_COL_T=double, -DLBM_COL_T=float, -DLBM_COL_TALL=1 or -DLBM_COL_TALL=0"
#endif
#undef LBM_KC

}  // namespace lbmk
