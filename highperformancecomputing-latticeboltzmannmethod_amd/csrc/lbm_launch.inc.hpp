// csrc/lbm_launch.inc.hpp — kernel arguments, layouts and the launchers of every step-kernel family (single iteration, fused tiles, deep LDS tiles, registers), forces
// (part of the one host translation unit lbm_hip.hip, which includes it in this place; round 4 split a 2 100-line file by concern)

template <typename T>
KArgs<T> make_kargs(const lbm_ctx* c, int src, int dst, int t) {
    KArgs<T> a;
    a.src = static_cast<const T*>(c->buf[src]);
    a.dst = static_cast<T*>(c->buf[dst]);
    a.plane = (long)c->plane;
    a.pitch = c->pitch;
    a.xoff = c->xoff;
    a.nx = c->nx;
    a.ny_loc = c->nyl;
    a.ny_glob = c->p.ny;
    a.y_start = c->p.y_start;
    a.cyl_x = c->cyl_x;
    a.cyl_y = c->cyl_y;
    a.cyl_r2 = (double)(c->cyl_r * c->cyl_r);
    a.tau_inv = (T)(1.0 / c->p.tau);
    a.u_in = (T)c->p.inlet_velocity;
    a.unstable_t = c->d_unstable;
    a.t = t - c->tbase_host;
    a.t_base = c->d_tbase;
    a.y_lo = 0;
    a.y_cnt = c->nyl;
    a.y_lo2 = 0;
    a.y_cnt2 = 0;
    a.reverse = 0;
    return a;
}

// Strides of the two layouts. Planar: plane stride = whole rows rounded up to k*64 KiB + 4 KiB (nine planes whose
// stride is a multiple of 64 KiB put the nine accesses of a wave on the same HBM channel group: 5.2-5.4 TB/s at +0
// vs 5.9-6.1 TB/s at +1..8 KiB, 4096x1024 fp64). Row-interleaved: the nine sub-rows of a lattice row are adjacent.
inline void configure_layout(lbm_ctx* c, int layout) {
    c->layout = layout;
    if (layout == 1) {
        c->plane = (size_t)c->pitch0;
        c->pitch = Q * c->pitch0;
        c->total = (size_t)c->pitch * (c->nyl + 2 * GR);
    } else {
        const size_t raw = (size_t)c->pitch0 * (c->nyl + 2 * GR) * c->esize;
        const size_t w = 65536;
        c->plane = ((raw + w - 1) / w * w + 4096) / c->esize;
        c->pitch = c->pitch0;
        c->total = (size_t)Q * c->plane;
    }
}
inline size_t buffer_bytes(const lbm_ctx* c) { return c->total * c->esize + 256; }  // +slack: displaced vector load

// Launch one step-family kernel over the local rows [a.y_lo, a.y_lo + a.y_cnt). Instantiated: MODE_STEP in both store
// policies and both arithmetic modes, MODE_COLLIDE_ONLY in both arithmetic modes, MODE_STREAM_ONLY once (no collision in it).
template <typename T, int MODE>
void launch_rows(const lbm_ctx* c, const KArgs<T>& a, hipStream_t s) {
    const bool nt = (MODE == MODE_STEP) && c->use_nt;
    const bool fast = (MODE != MODE_STREAM_ONLY) && c->arith == AR_CONTRACTED;
    const dim3 grid((c->nx + 255) / 256, a.y_cnt + a.y_cnt2), block(256);
#define LBM_K1(NT_, AR_) hipLaunchKernelGGL((k_step_site<T, MODE, NT_, AR_>), grid, block, 0, s, a)
    if constexpr (MODE == MODE_STEP) {
        if (fast) { if (nt) LBM_K1(true, AR_CONTRACTED); else LBM_K1(false, AR_CONTRACTED); }
        else { if (nt) LBM_K1(true, AR_STRICT); else LBM_K1(false, AR_STRICT); }
    } else if constexpr (MODE == MODE_COLLIDE_ONLY) {
        if (fast) LBM_K1(false, AR_CONTRACTED); else LBM_K1(false, AR_STRICT);
    } else {
        LBM_K1(false, AR_STRICT);
    }
#undef LBM_K1
}

// "deep" plans: shape id -> iterations per launch and tile. 1..3: LDS-image tiles (k_stepd_tile: six / seven iterations on
// 64x16 tiles, eight on 32x32; what a grid of a single round of blocks picks); 6 / 7: the register-resident column kernel
// (k_stepc_col, R = 4 rows per thread x 8 waves = a 64 x 32 region per block, two blocks per CU; fp64 strict: 2 rows x 12 waves = 64 x 24) with five / six iterations —
// a plan of either uses both depths, and on a context without strip faces seven iterations too, for what a segment leaves
// over (20 = 7 + 7 + 6; at 4096x1024 fp64 seven iterations run at 160.6 GLUPS against 161.7 for six — eight, 154.3, are
// not built); 9: the same family with seven iterations as the plan's depth (8192x2048 fp64: 175.9 against 169.9); 8: fp32 only, the same kernel on TALL 64 x 48 regions (twelve waves x four rows) with seven iterations, six / eight for
// what a segment leaves over. Ids 4 / 5 were round 2's 32x16 LDS tiles: retired.
// rows of one band of tiles of a launch of `depth` iterations (the edge bands of a strip are one band each)
inline int deep_rows(const lbm_ctx* c, int id, int depth) {
    if (deep_is_col(id)) return col_tile_h(depth, (int)c->esize, c->arith == 0, deep_is_tall(id));
    return id == 3 ? 32 : 16;
}
// A fused kernel over the local rows [a.y_lo, a.y_lo + a.y_cnt): iterations a.t .. a.t + depth - 1 (depth 2..8).
template <typename T>
void launch_fused_rows(const lbm_ctx* c, const KArgs<T>& a, int depth, hipStream_t s) {
    const int shape = c->deep;
    K2Extra<T> e;
    e.feq_in = static_cast<const T*>(c->d_feq);
    e.small = ((c->total + (size_t)c->pitch) * c->esize + 1024 < (size_t(1) << 32)) ? 1 : 0;   // 32-bit byte offsets (+ one row of slack)
    e.xcd = c->xcd;
    e.nt = c->use_nt;
    e.ntl = c->use_ntl;
    const bool fast = c->arith == AR_CONTRACTED;
    if (c->deep_now && deep_is_col(shape)) {    // D iterations with the lattice in registers (k_stepc_col, lbm_col.hip)
        if constexpr (sizeof(T) == 4) {
            if (deep_is_tall(shape)) {
                if (fast) launch_col_tall_contracted(a, e, depth, s); else launch_col_tall_strict(a, e, depth, s);
                return;
            }
        }
        launch_col<T>(a, e, depth, c->use_nt != 0, fast, s);
        return;
    }
    if (c->deep_now) {    // D iterations on a deep LDS tile (k_stepd_tile; whole-domain launches of small grids)
#define LBM_KD(TX_, TY_, D_) do { \
        dim3 gridd((c->nx + TX_ - 1) / TX_, (a.y_cnt + TY_ - 1) / TY_ + (a.y_cnt2 + TY_ - 1) / TY_); \
        if (fast) hipLaunchKernelGGL((k_stepd_tile<T, TX_, TY_, D_, AR_CONTRACTED>), gridd, dim3(TX_ * TY_), 0, s, a, e); \
        else hipLaunchKernelGGL((k_stepd_tile<T, TX_, TY_, D_, AR_STRICT>), gridd, dim3(TX_ * TY_), 0, s, a, e); } while (0)
        switch (shape) {
            case 1: LBM_KD(64, 16, 6); break;
            case 2: LBM_KD(64, 16, 7); break;
            default: LBM_KD(32, 32, 8); break;
        }
#undef LBM_KD
        return;
    }
    const int ty = c->pair_ty;
    dim3 grid((c->nx + 63) / 64, (a.y_cnt + ty - 1) / ty + (a.y_cnt2 + ty - 1) / ty);
#define LBM_KT(K_, TY_, NTH_, G_) do { if (fast) hipLaunchKernelGGL((K_<T, TY_, NTH_, AR_CONTRACTED>), G_, dim3(NTH_), 0, s, a, e); \
                                       else hipLaunchKernelGGL((K_<T, TY_, NTH_, AR_STRICT>), G_, dim3(NTH_), 0, s, a, e); } while (0)
    if (depth == 4) {   // four iterations: 64x8 tiles only (LDS)
        dim3 grid4((c->nx + 63) / 64, (a.y_cnt + 7) / 8 + (a.y_cnt2 + 7) / 8);
        // fp64: 70.5 KB of LDS per block = two blocks per CU, so 1024 threads fill the 32 wave slots; fp32 (35 KB) fills them
        // with four 512-thread blocks (measured: 1024 threads -14 % in fp32, +3 % in fp64)
        constexpr int N4 = sizeof(T) == 8 ? 1024 : 512;
        LBM_KT(k_step4_tile, 8, N4, grid4);
    } else if (depth == 3) {
        if (ty == 12) LBM_KT(k_step3_tile, 12, 1024, grid); else LBM_KT(k_step3_tile, 8, 512, grid);
    } else {
        if (ty == 12) LBM_KT(k_step2_tile, 12, 768, grid); else LBM_KT(k_step2_tile, 8, 512, grid);
    }
#undef LBM_KT
}
inline bool pair_possible(const lbm_ctx*) { return true; }   // partial tiles cover any nx
template <typename T>
int launch_step(lbm_ctx* c, int src, int dst, int t, int mode, hipStream_t s) {
    KArgs<T> a = make_kargs<T>(c, src, dst, t);
    a.reverse = ((mode == MODE_STEP || mode >= 100) && c->alternate && (c->launches_total & 1)) ? 1 : 0;
    switch (mode) {
        case MODE_STEP: launch_rows<T, MODE_STEP>(c, a, s); break;
        case 102: launch_fused_rows<T>(c, a, 2, s); break;     // iterations t, t+1
        case 103: launch_fused_rows<T>(c, a, 3, s); break;     // iterations t, t+1, t+2
        case 104: launch_fused_rows<T>(c, a, 4, s); break;     // iterations t .. t+3 (k_step4_tile, no strip faces)
        case MODE_COLLIDE_ONLY: launch_rows<T, MODE_COLLIDE_ONLY>(c, a, s); break;
        default: break;
    }
    HIPCHK(hipGetLastError());
    return LBM_OK;
}

template <typename T>
int launch_forces(lbm_ctx* c, double* out, int t) {
    if (c->rec) {      // dry run: the force kernel reads this strip's rows of P_t on the compute stream
        ChoreoOp o; o.kind = ChoreoOp::FORCES; o.strip = c->group_k; o.stream = 0; o.buf = c->cur; o.t = t; o.r0 = 0; o.r1 = c->nyl; o.r_strip = c->group_k;
        c->rec->ops.push_back(o);
        return LBM_OK;
    }
    ForceArgs<T> f;
    f.cur = static_cast<const T*>(c->buf[c->cur]);
    f.plane = (long)c->plane; f.pitch = c->pitch; f.xoff = c->xoff;
    f.nx = c->nx; f.ny_loc = c->nyl; f.ny_glob = c->p.ny; f.y_start = c->p.y_start;
    f.cyl_x = c->cyl_x; f.cyl_y = c->cyl_y; f.cyl_r = c->cyl_r; f.cyl_r2 = (double)(c->cyl_r * c->cyl_r);
    f.x0 = std::max(0, c->cyl_x - c->cyl_r - 1);
    f.x1 = std::min(c->nx - 1, c->cyl_x + c->cyl_r + 1);
    f.y0 = std::max(0, c->cyl_y - c->cyl_r - 1 - c->p.y_start);
    f.y1 = std::min(c->nyl - 1, c->cyl_y + c->cyl_r + 1 - c->p.y_start);
    f.out = out; f.t = t;
    hipLaunchKernelGGL((k_forces<T>), dim3(1), dim3(1024), 0, c->stream, f);
    HIPCHK(hipGetLastError());
    return LBM_OK;
}

