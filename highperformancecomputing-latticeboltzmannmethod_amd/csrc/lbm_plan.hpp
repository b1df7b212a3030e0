// csrc/lbm_plan.hpp — which formulations of the step a context may run: the candidate list of the plan measurement
// (choose_plan, lbm_hip.hip), the strip rule, and the names / option strings of a plan. Pure host C++ (no HIP, no device): the
// library exports it through lbm_debug_plan_candidates so that the bench's bookkeeping can be checked on the CPU
// (tests/test_bench_cpu.py: every fused candidate of every BASELINE.json grid has a committed counter pass).
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "lbm_col_api.hpp"

namespace lbmk {

// layout 0 planar / 1 row-interleaved; nt: non-temporal stores;
// alternate: walk direction alternates per launch; fuse: iterations per launch of the tile kernels (1..4) or of the deep shape;
// ty: tile height of the two- / three-iteration tile kernels (8 or 12); xcd: XCD-aware tile walk; deep: 0, or the deep shape
// (1..3: k_stepd_tile six / seven / eight iterations; 6 / 7: k_stepc_col five / six iterations in registers on 64x32 regions;
// 9: the same kernel with SEVEN iterations as the plan's depth — the largest grids; 8: fp32 only, k_stepc_col seven iterations
// on TALL 64x48 regions — lbm_col_api.hpp); ntl: the register kernel's level-1 loads
// are non-temporal
struct Plan { int layout, nt, alternate, fuse, ty, xcd; std::string name; int deep = 0; int ntl = 0; };

inline bool deep_is_col(int id) { return id >= 6 && id <= 9; }
inline bool deep_is_tall(int id) { return id == 8; }
inline bool deep_valid(int id) { return id == 0 || (id >= 1 && id <= 3) || deep_is_col(id); }
inline int deep_depth(int id) {
    static const int d[10] = {0, 6, 7, 8, 0, 0, 5, 6, 7, 7};
    return id >= 0 && id <= 9 ? d[id] : 0;
}
inline const char* deep_tile(int id) {
    static const char* t[4] = {"", "64,16", "64,16", "32,32"};
    return id >= 0 && id <= 3 ? t[id] : "";
}

struct PlanQuery {
    int nx = 0, nyl = 0, ny_glob = 0;   // columns, rows of this strip, rows of the lattice
    int esize = 8;                      // bytes per element
    int num_cus = 256;
    int nstrips = 1;                    // strips of the run (ranks or group members)
    bool strips = false;                // this context has (or simulates) strip faces
    bool faces = false;                 // ... and at least one of them is an internal face
    bool tune = true, can_tune = true;  // option "tune"; the grid is neither too small nor too large to measure
};

// The strip rule: 0: three iterations on 64x12 LDS tiles in pairs between exchanges; 1: six iterations on 64x16 LDS tiles of
// 1024 threads (one cell per thread: the shortest launch, and on strips this short the chain edge band -> exchange -> edge band IS
// the time step); 7: six iterations on 64x32 regions held in registers (four cells per thread: the highest throughput).
// One GPU, one rank of N exchanging with itself through RCCL, 4096 columns (tools/strip_proxy.py, profiles/r03): 128 rows
// 7.6 us per iteration on the LDS tiles against 8.5 in registers; 256 rows 13.2 against 11.6. A function of the GLOBAL grid and
// the number of strips only: every rank — measuring or not — issues the same launch depths.
inline int strip_rule_deep(bool strips, int ny_glob, int nstrips) {
    const int strip_rows = ny_glob / (nstrips > 0 ? nstrips : 1);
    return !strips ? 0 : strip_rows >= 192 ? 7 : strip_rows >= 64 ? 1 : 0;
}

// Every candidate lbm_initialise times for this context (one entry: nothing is measured). `fixed` = the plan the options pin.
inline std::vector<Plan> plan_candidates(const PlanQuery& q, const Plan& fixed) {
    std::vector<Plan> cand;
    // "small": 1024-cell tiles make at most two rounds of one block per CU (fp32: two blocks per CU)
    const bool small_grid = (size_t)q.nx * q.nyl <= (size_t)2048 * q.num_cus * (q.esize == 4 ? 2 : 1);
    const int strip_deep = strip_rule_deep(q.strips, q.ny_glob, q.nstrips);
    const std::string deep_name = strip_deep == 7 ? "row-interleaved/6-step 64x32 in registers" : "row-interleaved/6-step 64x16";
    if (!q.tune) { cand.push_back(fixed); return cand; }
    if (!q.can_tune) {
        if (strip_deep) cand.push_back({1, 1, 0, 6, 12, 1, deep_name + " (default, not measured)", strip_deep});
        else if (q.strips) cand.push_back({1, 1, 0, 3, 12, 1, "row-interleaved (default, not measured)"});
        else cand.push_back({0, 1, 0, 3, 12, 0, "planar (default, not measured)"});
        return cand;
    }
    // Strips exchange GR rows x 9 sub-rows as one contiguous run: row-interleaved only. Every rank must issue the same
    // sequence of launches (one exchange per launch), so the fusion depth and tile shape of a strip run are fixed by rule;
    // only rank-local choices (the store policy) are measured.
    if (strip_deep) {
        cand.push_back({1, 1, 0, 6, 12, 1, deep_name + "/nt-store/xcd", strip_deep});
        cand.push_back({1, 0, 0, 6, 12, 1, deep_name + "/xcd", strip_deep});
        if (strip_deep == 7) cand.push_back({1, 0, 0, 6, 12, 1, deep_name + "/nt-load/xcd", strip_deep, 1});
        return cand;
    }
    if (q.strips) {
        cand.push_back({1, 1, 0, 3, 12, 1, "row-interleaved/3-step 64x12/nt-store/xcd"});
        cand.push_back({1, 1, 0, 3, 12, 0, "row-interleaved/3-step 64x12/nt-store"});
        cand.push_back({1, 0, 1, 3, 12, 1, "row-interleaved/3-step 64x12/alternate/xcd"});
        return cand;
    }
    cand.push_back({1, 1, 0, 4, 8, 1, "row-interleaved/4-step 64x8/nt-store/xcd"});
    cand.push_back({1, 1, 0, 6, 12, 1, "row-interleaved/6-step 64x32 in registers/nt-store/xcd", 7});   // k_stepc_col
    cand.push_back({1, 1, 0, 5, 12, 1, "row-interleaved/5-step 64x32 in registers/nt-store/xcd", 6});
    // (non-temporal stores pay where most of the lattice fits the 256 MiB Infinity Cache — 4096x1024 fp64: +1 % — and
    // cost 3-12 % on the large grids: 8192x2048 fp64 168 -> 173 GLUPS, 16384x4096 fp32 259 -> 290 without them)
    cand.push_back({1, 0, 0, 6, 12, 1, "row-interleaved/6-step 64x32 in registers/xcd", 7});
    cand.push_back({1, 0, 0, 5, 12, 1, "row-interleaved/5-step 64x32 in registers/xcd", 6});
    cand.push_back({1, 0, 1, 6, 12, 1, "row-interleaved/6-step 64x32 in registers/alternate/xcd", 7});
    // (round 4: non-temporal level-1 LOADS with plain stores: 164-167 against 158-163 GLUPS at 4096x1024 fp64)
    cand.push_back({1, 0, 0, 6, 12, 1, "row-interleaved/6-step 64x32 in registers/nt-load/xcd", 7, 1});
    cand.push_back({1, 0, 1, 6, 12, 1, "row-interleaved/6-step 64x32 in registers/nt-load/alternate/xcd", 7, 1});
    cand.push_back({1, 0, 0, 5, 12, 1, "row-interleaved/5-step 64x32 in registers/nt-load/xcd", 6, 1});
    // (round 4: seven iterations as the plan's own depth pay on the largest grids — 8192x2048 fp64 175.9 GLUPS against 169.9 for six;
    // at 4096x1024 they lose, 157-160 against 160-166)
    if (!small_grid) {
        cand.push_back({1, 0, 0, 7, 12, 1, "row-interleaved/7-step 64x32 in registers/xcd", 9});
        cand.push_back({1, 0, 1, 7, 12, 1, "row-interleaved/7-step 64x32 in registers/alternate/xcd", 9});
    }
    // (round 4, fp32: 64x48 regions on twelve waves — 16384x4096 320 GLUPS against 289-298 on 64x32; 4096x1024 262 against 260)
    if (q.esize == 4 && !small_grid) {
        cand.push_back({1, 0, 0, 7, 12, 1, "row-interleaved/7-step 64x48 in registers/xcd", 8});
        cand.push_back({1, 0, 1, 7, 12, 1, "row-interleaved/7-step 64x48 in registers/alternate/xcd", 8});
    }
    cand.push_back({1, 1, 0, 6, 12, 1, "row-interleaved/6-step 64x16/nt-store/xcd", 1});
    if (small_grid && !q.faces) {   // one round of LDS-filling tiles: a launch's load and store phases are paid once per 7-8 iterations
        cand.push_back({1, 1, 0, 7, 12, 1, "row-interleaved/7-step 64x16/nt-store/xcd", 2});
        cand.push_back({1, 1, 0, 8, 12, 1, "row-interleaved/8-step 32x32/nt-store/xcd", 3});
    }
    cand.push_back({1, 1, 0, 3, 12, 1, "row-interleaved/3-step 64x12/nt-store/xcd"});
    cand.push_back({1, 1, 0, 3, 8, 1, "row-interleaved/3-step 64x8/nt-store/xcd"});
    cand.push_back({1, 1, 0, 2, 12, 1, "row-interleaved/2-step 64x12/nt-store/xcd"});
    cand.push_back({1, 1, 0, 2, 8, 1, "row-interleaved/2-step 64x8/nt-store/xcd"});
    cand.push_back({1, 1, 0, 1, 0, 0, "row-interleaved/site/nt-store"});
    cand.push_back({1, 0, 1, 1, 0, 0, "row-interleaved/site/alternate"});
    cand.push_back({0, 1, 0, 4, 8, 1, "planar/4-step 64x8/nt-store/xcd"});
    cand.push_back({0, 1, 0, 6, 12, 1, "planar/6-step 64x32 in registers/nt-store/xcd", 7});
    cand.push_back({0, 1, 0, 3, 12, 1, "planar/3-step 64x12/nt-store/xcd"});
    cand.push_back({0, 1, 0, 3, 12, 0, "planar/3-step 64x12/nt-store"});
    cand.push_back({0, 1, 0, 2, 12, 0, "planar/2-step 64x12/nt-store"});
    cand.push_back({0, 0, 1, 3, 12, 0, "planar/3-step 64x12/alternate"});
    cand.push_back({0, 0, 1, 1, 0, 0, "planar/site/alternate"});
    return cand;
}

// the dominant kernel of a plan, as rocprofv3 names it (minus "lbmk::" and blanks)
inline std::string plan_kernel_name(int fuse, int deep, int pair_ty, int nt, int arith, int esize) {
    char name[96];
    const char* t = esize == 4 ? "float" : "double";
    const bool tall = deep_is_tall(deep) && esize == 4;      // (tall regions and seven-iteration launches: plain stores only)
    const char* nts = nt && !tall && !(deep_is_col(deep) && deep_depth(deep) == 7) ? "true" : "false";
    if (fuse > 2 && deep_is_col(deep)) snprintf(name, sizeof(name), "k_stepc_col<%s,%d,%d,%d,%s,%d>", t, col_rows_per_thread(esize, arith == 0, tall), col_waves(esize, arith == 0, tall), deep_depth(deep), nts, arith);
    else if (fuse > 2 && deep) snprintf(name, sizeof(name), "k_stepd_tile<%s,%s,%d,%d>", t, deep_tile(deep), deep_depth(deep), arith);
    else if (fuse == 4) snprintf(name, sizeof(name), "k_step4_tile<%s,8,%d,%d>", t, esize == 8 ? 1024 : 512, arith);
    else if (fuse > 1) snprintf(name, sizeof(name), "k_step%d_tile<%s,%d,%d,%d>", fuse, t, pair_ty, pair_ty == 12 ? (fuse == 3 ? 1024 : 768) : 512, arith);
    else snprintf(name, sizeof(name), "k_step_site<%s,0,%s,%d>", t, nts, arith);
    return name;
}

// the plan as lbm_set_option pairs: `deep` (which sets the depth of its launches itself) or `fuse`, never both
inline std::string plan_option_string(int layout, int nt, int alternate, int pair_ty, int xcd, int fuse, int deep, int ntl = 0) {
    char b[128];
    snprintf(b, sizeof(b), "layout=%d nt=%d alternate=%d pair_ty=%d xcd=%d", layout, nt, alternate, pair_ty ? pair_ty : 8, xcd);
    std::string s(b);
    if (deep) s += " deep=" + std::to_string(deep);
    else s += " fuse=" + std::to_string(fuse > 0 && fuse <= 4 ? fuse : 1);
    if (ntl) s += " ntl=1";
    return s;
}

}  // namespace lbmk
