// csrc/lbm_tune.inc.hpp — measurement at lbm_initialise: plan (layout / kernel / store and load policy / walk) and strip schedule
// (part of the one host translation unit lbm_hip.hip, which includes it in this place; round 4 split a 2 100-line file by concern)
// ---- plan: pick layout / kernel / store policy / traversal by measurement --------------------------------
// The step is a pure 18-stream copy with arithmetic attached; which formulation the memory system likes best
// depends on the grid (working set vs the 256 MiB Infinity Cache, row length vs channel interleave) and even on
// where the allocation landed physically (measured: the same planar plan runs at 100 us or 110 us per step at
// 4096x1024 fp64 depending on the allocation). All candidates compute bit-identical results, so lbm_initialise
// times each one on the real buffers (12 warm-up iterations, then the faster of two 36-iteration windows) and keeps the fastest together
// with the very allocation it was measured on.

inline void apply_plan(lbm_ctx* c, const Plan& pl) {
    configure_layout(c, pl.layout);
    c->use_nt = pl.nt; c->use_ntl = pl.ntl; c->alternate = pl.alternate; c->fuse = pl.fuse > 0 ? pl.fuse : 1; c->xcd = pl.xcd;
    if (pl.ty) c->pair_ty = pl.ty;
    c->deep = pl.deep;
}

template <typename T>
int time_plan(lbm_ctx* c, float* ms_out, int window = 36) {
    int rc = init_state<T>(c);
    if (rc) return rc;
    auto run = [&](int n) -> int {
        for (int k = 0; k < n;) {
            // far from the end of a call (and with a room that is a multiple of three, so that a three-iteration plan is not
            // handed the four-iteration kernel for a remainder): every launch has the plan's own depth — a window is no
            // multiple of 7 or 8. No strip logic: the probe times local launches.
            const int took = advance<T>(c, 3 * (1 << 18) + (c->trailing_pair ? 0 : 1), 0, false, false);
            if (took < 0) return took;
            k += took;
        }
        return LBM_OK;
    };
    rc = run(12);
    if (rc) return rc;
    *ms_out = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {      // the faster of two windows of `window` iterations (36: six to a dozen fused launches each)
        const int t0 = c->steps_done;
        HIPCHK(hipEventRecord(c->ev_t0, c->stream));
        rc = run(window);
        if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_t1, c->stream));
        HIPCHK(hipEventSynchronize(c->ev_t1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
        *ms_out = std::min(*ms_out, ms / (float)(c->steps_done - t0));     // per iteration
    }
    return LBM_OK;
}

template <typename T>
int choose_plan(lbm_ctx* c) {
    const bool strips = (c->comm && c->nranks > 1) || c->group_n > 1 || c->loopback;
    const Plan fixed = {(strips || c->loopback) ? 1 : c->layout, c->use_nt, c->alternate, c->fuse, c->pair_ty, c->xcd,
                        "fixed by options", c->deep, c->use_ntl};
    const bool p2 = pair_possible(c);
    (void)p2;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    configure_layout(c, 1);
    const size_t need = 2 * buffer_bytes(c);
    // tiny grids are launch/latency bound (nothing to choose); huge ones cannot afford a second live allocation
    const bool can_tune = c->tune && (size_t)c->nx * c->nyl >= (1u << 16) && 2 * need + (1u << 28) < free_b;
    // the strip rule (lbm_plan.hpp): a function of the global grid and the number of strips only, so that every rank —
    // measuring or not — issues the same launch depths
    const int nstrips = c->group_n > 1 ? c->group_n : (c->comm && c->nranks > 1) ? c->nranks : 1;
    PlanQuery q;
    q.nx = c->nx; q.nyl = c->nyl; q.ny_glob = c->p.ny; q.esize = (int)c->esize; q.num_cus = c->num_cus; q.nstrips = nstrips; q.strips = strips;
    q.tune = c->tune != 0; q.can_tune = can_tune; q.faces = face_south(c) || face_north(c);
    const std::vector<Plan> cand = plan_candidates(q, fixed);
    free_buffers(c);                                // a second lbm_initialise starts from no population buffers
    // First round: every candidate once; the three fastest keep their allocations. Final round: those three again with
    // longer windows (candidates within 2 % of each other are common and the first round cannot tell them apart).
    struct Kept { int k; float ms; void* buf[2]; };
    std::vector<Kept> top;
    auto drop_all = [&]() { for (Kept& t : top) for (void*& q : t.buf) if (q) { (void)hipFree(q); q = nullptr; } top.clear(); };
    const bool room = can_tune && 4 * need + (1u << 28) < free_b;     // three kept allocations + the one being probed
    const size_t keep = room ? 3 : 1;
    for (size_t k = 0; k < cand.size(); ++k) {
        apply_plan(c, cand[k]);
        c->buf[0] = c->buf[1] = nullptr;            // keep the best allocations alive while the next one is probed
        int rc = alloc_buffers(c);
        if (rc) { free_buffers(c); drop_all(); return rc; }
        float ms = 0.f;
        if (cand.size() > 1) {
            rc = time_plan<T>(c, &ms);
            if (rc) { free_buffers(c); drop_all(); return rc; }
        }
        top.push_back({(int)k, ms, {c->buf[0], c->buf[1]}});
        c->buf[0] = c->buf[1] = nullptr;
        std::stable_sort(top.begin(), top.end(), [](const Kept& x, const Kept& y) { return x.ms < y.ms; });
        while (top.size() > keep) { for (void* q : top.back().buf) if (q) (void)hipFree(q); top.pop_back(); }
    }
    std::string finalists;
    if (top.size() > 1) {
        for (Kept& t : top) {
            apply_plan(c, cand[(size_t)t.k]);
            c->buf[0] = t.buf[0]; c->buf[1] = t.buf[1];
            // (longer windows: the finalists are often 2-3 % apart — store policy, walk direction — and the alternating walk
            // only shows what it gains from the Infinity Cache once a few launches have gone both ways. Round 3 took the faster
            // of two 120-iteration windows and picked three different plans in three sessions at 16384x4096 fp32: now the MEDIAN
            // of three windows of at least 50 ms each — time_plan returns the faster of two halves, so six in all.)
            const int window = std::max(120, (int)std::ceil(25.0 / std::max(1e-4, (double)t.ms)));
            float w[3] = {0.f, 0.f, 0.f};
            int rc = LBM_OK;
            for (int r = 0; r < 3 && !rc; ++r) rc = time_plan<T>(c, &w[r], window);
            c->buf[0] = c->buf[1] = nullptr;
            if (rc) { drop_all(); return rc; }
            std::sort(w, w + 3);
            t.ms = w[1];
        }
        std::stable_sort(top.begin(), top.end(), [](const Kept& x, const Kept& y) { return x.ms < y.ms; });
        for (const Kept& t : top) {
            char fb[160];
            snprintf(fb, sizeof(fb), "%s%s %.2f", finalists.empty() ? "" : "; ", cand[(size_t)t.k].name.c_str(), t.ms * 1e3f);
            finalists += fb;
        }
        while (top.size() > 1) { for (void* q : top.back().buf) if (q) (void)hipFree(q); top.pop_back(); }
    }
    const int best = top[0].k;
    const float best_ms = top[0].ms;
    void* best_buf[2] = {top[0].buf[0], top[0].buf[1]};
    apply_plan(c, cand[best]);
    c->buf[0] = best_buf[0]; c->buf[1] = best_buf[1];
    c->launches_total = 0;
    c->last_was_pair = false;
    if (cand.size() > 1 && !finalists.empty())
        snprintf(c->plan_desc, sizeof(c->plan_desc), "%s (fastest of %zu measured, %.1f us/iteration; finalists, median of three windows, us/iteration: %s)",
                 cand[best].name.c_str(), cand.size(), best_ms * 1e3f, finalists.c_str());
    else if (cand.size() > 1) snprintf(c->plan_desc, sizeof(c->plan_desc), "%s (fastest of %zu measured, %.1f us/iteration)",
                                       cand[best].name.c_str(), cand.size(), best_ms * 1e3f);
    else snprintf(c->plan_desc, sizeof(c->plan_desc), "%s", cand[best].name.c_str());
    snprintf(c->plan_opts, sizeof(c->plan_opts), "%s", plan_option_string(c->layout, c->use_nt, c->alternate, c->pair_ty, c->xcd, c->fuse, c->deep, c->use_ntl).c_str());
    if (cand.size() > 1 && c->deep && !strips && best_ms > 0.f) {
        // What the shallow launches cost on THIS grid and allocation, for plan_launch's split of a segment's last iterations
        // (a single domain only: the strips of a run must all split alike, so they keep the fixed table).
        const Plan keep = cand[best];
        for (int d = 1; d <= 4; ++d) {
            Plan q = keep;
            q.deep = 0; q.fuse = d; q.ty = d == 4 ? 8 : 12;
            apply_plan(c, q);
            configure_layout(c, keep.layout);
            float ms = 0.f;
            int rc = time_plan<T>(c, &ms);
            if (rc) return rc;
            c->depth_rel[d - 1] = std::max(1.0, (double)ms / (double)best_ms);
        }
        c->depth_rel_measured = true;
        apply_plan(c, keep);
        c->launches_total = 0;
        c->last_was_pair = false;
    }
    return LBM_OK;
}

template <typename T> int do_steps(lbm_ctx** cs, int n, int nsteps, int of);
int allreduce_doubles(lbm_ctx* c, double* vals, int n, int op);

// What the ranks must agree on before the collective trials of tune_strip_schedule: packed so that ONE MIN-reduction yields the
// minimum and (negated) the maximum of every pin. v = {go, pin_overlap or -1, -(pin_overlap or -1), pin_deep or -1, -(...), pin_trim or -1, -(...)}.
inline void strip_pins_pack(bool go, bool overlap_pinned, int overlap, bool deep_pinned, int deep_halo, bool trim_pinned, int trim, double v[7]) {
    const double po = overlap_pinned ? (double)overlap : -1.0, pd = deep_pinned ? (double)deep_halo : -1.0, pt = trim_pinned ? (double)trim : -1.0;
    v[0] = go ? 1.0 : 0.0; v[1] = po; v[2] = -po; v[3] = pd; v[4] = -pd; v[5] = pt; v[6] = -pt;
}
// after the MIN-reduction: false = the ranks disagree (some pinned, some not, or to different values)
inline bool strip_pins_agree(const double v[7], int* go, int* overlap_pinned, int* overlap, int* deep_pinned, int* deep_halo, int* trim_pinned, int* trim) {
    if (v[1] != -v[2] || v[3] != -v[4] || v[5] != -v[6]) return false;
    *go = v[0] > 0.5;
    *overlap_pinned = v[1] >= 0.0; if (*overlap_pinned) *overlap = (int)v[1];
    *deep_pinned = v[3] >= 0.0; if (*deep_pinned) *deep_halo = (int)v[3];
    *trim_pinned = v[5] >= 0.0; if (*trim_pinned) *trim = (int)v[5];
    return true;
}

// Strip schedule by measurement (one rank of a multi-process run; collective: every rank runs the same trials and sees
// the same reduced timings, so all ranks choose alike). The schedules — exchange overlapped with the interior rows of the
// same launch (1), of the next, extended launch (2) or serialised (0); one exchange per two launches (deep halo) or per
// launch — compute identical results; which is fastest depends on the strip height and on the link (overlap costs two extra
// launches and three events per group, which a short strip cannot hide). Each candidate: 60 warm-up + 240 timed iterations
// (forty launch groups of six) with the real transport, MAX over the ranks; then the two fastest are timed again, twice,
// and the faster of the two wins (candidates 2-3 % apart are common: round 2's single window of four groups could not rank
// them). What travels per exchange and face is the same in every schedule — GR rows x 9 populations, one contiguous message
// — so the payload per iteration depends on the iterations between two exchanges only; lbm_strip_schedule() reports it.
template <typename T>
int tune_strip_schedule(lbm_ctx* c) {
    const bool multi = c->comm && (c->nranks > 1 || c->loopback == 2);
    auto describe = [&](const char* how, int tried, double us_per_it) {
        // iterations between two exchanges: a deep launch (up to GR iterations) exchanges after every launch; the
        // three-iteration plans after every launch, or after every second one with the deep halo
        const bool deep_launches = c->deep && deep_depth(c->deep) <= GR;
        const int its = deep_launches ? deep_depth(c->deep) * (deep_pairs(c) ? 2 : 1) : std::min(c->fuse, 3) * (c->deep_halo ? 2 : 1);
        const double face_bytes = (double)face_payload_elems(c) * c->esize;
        int n = snprintf(c->sched_desc, sizeof(c->sched_desc), "overlap=%d deep_halo=%d halo_trim=%d (%s", c->overlap, c->deep_halo, c->halo_trim, how);
        if (tried > 0 && n > 0 && n < (int)sizeof(c->sched_desc))
            n += snprintf(c->sched_desc + n, sizeof(c->sched_desc) - n, " of %d measured, %.2f us/iteration", tried, us_per_it);
        if (n > 0 && n < (int)sizeof(c->sched_desc))
            snprintf(c->sched_desc + n, sizeof(c->sched_desc) - n, "); %.0f B per face and exchange = %.0f B per face and iteration (%d iterations per exchange)",
                     face_bytes, face_bytes / std::max(its, 1), its);
    };
    describe(multi ? "fixed by options" : "default", 0, 0.0);
    if (!multi) return LBM_OK;
    {   // The trials below are COLLECTIVE (send/recv with the neighbours, an all-reduce per schedule): whether they run — and WHICH
        // of them run: a pinned half of the schedule removes trials — must be the same decision on every rank. Strips may differ
        // in height (191 rows over 8 ranks: seven of 24 and one of 23) and, in principle, in their options, so the decision and
        // the pins are reduced over the ranks first (one MIN over {go, pin, -pin, ...}): all of them tune the same list, or the
        // call fails on every rank alike (ranks with different pins ran different numbers of collective trials and
        // the first multi-process lbm_initialise hung in RCCL instead of returning an error).
        double v[7];
        strip_pins_pack(c->tune && c->nyl >= 4 * HR1, c->overlap_pinned, c->overlap, c->deep_pinned, c->deep_halo, c->trim_pinned, c->halo_trim, v);
        int rc = allreduce_doubles(c, v, 7, 2);      // MIN
        if (rc) return rc;
        int go = 0, po = 0, pd = 0, pt = 0, ov = c->overlap, dh = c->deep_halo, tr = c->halo_trim;
        if (!strip_pins_agree(v, &go, &po, &ov, &pd, &dh, &pt, &tr))
            return fail(LBM_ERR_ARG, "the ranks of this run pin different strip schedules (lbm_set_option overlap / deep_halo / halo_trim): set the same on every rank");
        c->overlap_pinned = po != 0; c->deep_pinned = pd != 0; c->trim_pinned = pt != 0;
        if (po) c->overlap = ov;
        if (pd) c->deep_halo = dh;
        if (pt) c->halo_trim = tr;
        describe("fixed by options", 0, 0.0);
        if (!go || (po && pd && pt)) return LBM_OK;
    }
    // (every error return below leaves the context as it came: a trial changes trailing_pair and — the depth trial — deep and fuse)
    struct Restore { lbm_ctx* c; int tp, deep, fuse; bool armed = true;
                     ~Restore() { if (armed) { c->trailing_pair = tp; c->deep = deep; c->fuse = fuse; } } } guard{c, c->trailing_pair, c->deep, c->fuse};
    const int keep_tp = c->trailing_pair;
    c->trailing_pair = 1;
    constexpr int WARM = 60, TIMED = 240;      // (the warm-up is long enough to take the one-off graph capture of a schedule)
    int frame_rows = halo_rows(c);             // ghost rows per face that the exchanges so far have kept fresh
    auto trial = [&](int o, int d, double* worst_ms) -> int {
        c->overlap = o; c->deep_halo = d;
        int rc = LBM_OK;
        // (this used to compare with halo_rows(c) read AFTER the caller had switched c->deep to the eight-iteration shape,
        // so the depth trial never refreshed rows 7-8 a six-row schedule had left stale — timing only, the state is re-initialised)
        const bool deeper = halo_rows(c) > frame_rows;
        frame_rows = halo_rows(c);
        if (deeper) {      // the schedule on trial refreshes a deeper ghost frame: fill it before its first launch reads it
            rc = join_comm(c);
            if (rc) return rc;
            rc = exchange_rccl<T>(c, c->cur, c->stream);
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(c->stream));
            c->mid_pair = false;
        }
        rc = do_steps<T>(&c, 1, WARM, 0);
        if (rc) return rc;
        rc = join_comm(c);
        if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_t0, c->stream));
        rc = do_steps<T>(&c, 1, TIMED, 0);
        if (rc) return rc;
        rc = join_comm(c);
        if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_t1, c->stream));
        HIPCHK(hipEventSynchronize(c->ev_t1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
        *worst_ms = (double)ms;
        return allreduce_doubles(c, worst_ms, 1, 1);   // MAX over the ranks: the job advances at the pace of its slowest strip
    };
    // (overlap, deep_halo); overlap 2 needs launches in pairs. A deep plan exchanges after every launch whether deep_halo is 0 or 1,
    // so its second half of the list is the round-4 schedule instead: twelve rows per TWO launches of six iterations (deep_halo 2)
    static const int shallow[5][2] = {{1, 1}, {0, 1}, {2, 1}, {1, 0}, {0, 0}};
    static const int deepv[5][2] = {{1, 1}, {0, 1}, {1, 2}, {0, 2}, {2, 2}};
    const int (*variants)[2] = (c->deep && deep_depth(c->deep) <= HR1) ? deepv : shallow;
    struct Res { int o, d; double ms; };
    std::vector<Res> res;
    for (int v = 0; v < 5; ++v) {
        Res r{variants[v][0], variants[v][1], 0.0};
        if (c->overlap_pinned && r.o != c->overlap) continue;        // (a pinned half of the schedule stays as set)
        if (c->deep_pinned && r.d != c->deep_halo) continue;
        int rc = trial(r.o, r.d, &r.ms);
        if (rc) return rc;
        res.push_back(r);
    }
    const int tried = (int)res.size();
    std::string trials;      // every schedule's first-round time (MAX over the ranks), for the log: the margin of the choice
    for (const Res& r : res) {
        char tb[48];
        snprintf(tb, sizeof(tb), "%so%dd%d %.2f", trials.empty() ? "" : ", ", r.o, r.d, r.ms * 1e3 / TIMED);
        trials += tb;
    }
    std::stable_sort(res.begin(), res.end(), [](const Res& x, const Res& y) { return x.ms < y.ms; });
    if (res.size() > 2) res.resize(2);
    if (res.size() == 2) {
        for (Res& r : res) {        // every rank re-times the same two in the same order (the reduced timings are identical everywhere)
            double a = 0.0, b = 0.0;
            int rc = trial(r.o, r.d, &a);
            if (!rc) rc = trial(r.o, r.d, &b);
            if (rc) return rc;
            r.ms = std::min(a, b);
        }
        std::stable_sort(res.begin(), res.end(), [](const Res& x, const Res& y) { return x.ms < y.ms; });
    }
    // Strips of 64-191 rows run six iterations per launch on 64x16 LDS tiles by rule (deep 1). With the twelve-row ghost frame the
    // eight-iteration shape (32x32 tiles, deep 3: eight rows per face and exchange, a quarter fewer exchanges per iteration) runs
    // on strips too; which of the two is faster is a COLLECTIVE measurement like the schedule — the launch depth must be the same
    // on every rank, and the reduced timings are (one rank of eight exchanging with itself, 4096x128: 7.36 -> 6.78 us per iteration).
    std::string depth_note;
    if (!res.empty() && c->deep == 1) {
        Res best8{1, 1, 1e30};      // (an eight-iteration launch refreshes eight rows after every launch: deep_halo has no say)
        c->deep = 3; c->fuse = deep_depth(3);
        for (int o = 1; o >= 0; --o) {
            if (c->overlap_pinned && o != c->overlap) continue;
            double a = 0.0, b = 0.0;
            int rc = trial(o, 1, &a);
            if (!rc) rc = trial(o, 1, &b);
            if (rc) return rc;
            if (std::min(a, b) < best8.ms) { best8.o = o; best8.ms = std::min(a, b); }
        }
        char nb[160];
        snprintf(nb, sizeof(nb), "; launch depth measured over the ranks: six iterations (64x16 tiles) %.2f, eight (32x32 tiles) %.2f us/iteration",
                 res[0].ms * 1e3 / TIMED, best8.ms * 1e3 / TIMED);
        depth_note = nb;
        if (best8.ms < res[0].ms) {
            res[0] = best8;
            snprintf(c->plan_desc, sizeof(c->plan_desc), "row-interleaved/8-step 32x32%s/xcd (strip depth measured over the ranks)", c->use_nt ? "/nt-store" : "");
        } else {
            c->deep = 1; c->fuse = deep_depth(1);
        }
        snprintf(c->plan_opts, sizeof(c->plan_opts), "%s", plan_option_string(c->layout, c->use_nt, c->alternate, c->pair_ty, c->xcd, c->fuse, c->deep, c->use_ntl).c_str());
    }
    // The trimmed message (halo_trim 1: 9 hr - 9 of the 9 hr sub-rows of a face in five runs instead of one, 17 % fewer bytes at six rows)
    // on the schedule chosen so far — collective and MAX-reduced like everything above. Between the streams of one GPU five small copies
    // cost more than the bytes they save; what an xGMI link says is for the first run between real peers to measure.
    std::string trim_note;
    if (!res.empty() && !c->trim_pinned) {
        double t[2] = {1e30, 1e30};
        for (int tr = 0; tr < 2; ++tr) {
            c->halo_trim = tr;
            double a = 0.0, b = 0.0;
            int rc = trial(res[0].o, res[0].d, &a);
            if (!rc) rc = trial(res[0].o, res[0].d, &b);
            if (rc) { c->halo_trim = 0; return rc; }
            t[tr] = std::min(a, b);
        }
        c->halo_trim = t[1] < t[0] ? 1 : 0;
        res[0].ms = std::min(t[0], t[1]);
        char nb[128];
        snprintf(nb, sizeof(nb), "; message measured over the ranks: whole rows %.2f, trimmed to the sub-rows that are read %.2f us/iteration", t[0] * 1e3 / TIMED, t[1] * 1e3 / TIMED);
        trim_note = nb;
    }
    guard.armed = false;
    c->trailing_pair = keep_tp;
    if (!res.empty()) {
        c->overlap = res[0].o; c->deep_halo = res[0].d;
        describe("fastest", tried, res[0].ms * 1e3 / TIMED);
        const size_t n = strlen(c->sched_desc);
        if (n + 1 < sizeof(c->sched_desc))
            snprintf(c->sched_desc + n, sizeof(c->sched_desc) - n, "; first round, us/iteration by (overlap, deep_halo): %s%s%s", trials.c_str(), depth_note.c_str(), trim_note.c_str());
    }
    // back to iteration 0 with fresh halos
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->comm_stream));
    c->mid_pair = false; c->comm_issued = false; c->ext_split_pending = false; c->launches_total = 0; c->last_was_pair = false;
    const int big = INT_MAX;
    HIPCHK(hipMemcpyAsync(c->d_unstable, &big, sizeof(int), hipMemcpyHostToDevice, c->stream));
    int rc = init_state<T>(c);
    if (rc) return rc;
    return exchange_rccl<T>(c, c->cur, c->stream);
}

template <typename T>
int do_initialise(lbm_ctx* c) {
    int rc = choose_plan<T>(c);
    if (rc) return rc;
    const int big = INT_MAX;
    HIPCHK(hipMemcpyAsync(c->d_unstable, &big, sizeof(int), hipMemcpyHostToDevice, c->stream));
    rc = init_state<T>(c);
    if (rc) return rc;
    if ((c->comm || c->loopback) && c->group_n <= 1) {   // (a group exchanges once all members are initialised)
        rc = exchange_rccl<T>(c, c->cur, c->stream);
        if (rc) return rc;
        rc = tune_strip_schedule<T>(c);
        if (rc) return rc;
    }
    return LBM_OK;
}

