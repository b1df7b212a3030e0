// csrc/lbm_steps.inc.hpp — hipGraph replay of launch groups, do_steps (one context or an in-process group), snapshots and accessors, host-staged halos
// (part of the one host translation unit lbm_hip.hip, which includes it in this place; round 4 split a 2 100-line file by concern)
// ---- hipGraph replay of launch groups ------------------------------------------------------------------------
// A strip of an N = 8 run (4096 x 128) advances six iterations in ~25 us of GPU time, and one launch group — edge-band
// launch, three event records, three cross-stream waits, one RCCL group, interior launch — costs the host 33-43 us to issue:
// the strip is host-bound (round 2's proxy: 7.7 us per iteration against 5.9-6.1 on the GPU). GRAPH_GROUPS consecutive groups
// are therefore captured ONCE from the very code that issues them eagerly (plan_launch / issue_before / exchange_rccl /
// issue_after under hipStreamBeginCapture on the main stream; the side stream joins the capture through the first event wait
// and is joined back before the capture ends) and replayed with one hipGraphLaunch. An even number of groups returns the
// buffer parity, so one graph serves every replay; the kernels' iteration numbers (first-unstable bookkeeping) are relative
// to a device word that the graph itself advances (k_add_int). Where capture is refused — a transport that cannot be
// captured, an in-process group (its cross-device event waits belong to other captures) — the eager path runs as before.
constexpr int GRAPH_GROUPS = 4;
__global__ void k_add_int(int* p, int v) { *p += v; }
__global__ void k_set_int(int* p, int v) { *p = v; }


inline void graph_drop(lbm_ctx* c) {
    if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    c->giters = 0;
}

// May the next `GRAPH_GROUPS` groups of this context be replayed? (a single context with a device transport on a deep plan
// that exchanges after every launch, far from the end of the call and from any force output)
inline bool graph_wanted(const lbm_ctx* c, int remaining, int of, bool transport) {
    if (!c->use_graph || c->graph_failed || !transport || c->group_n > 1 || c->rec) return false;
    // RCCL send/recv between REAL peers under stream capture has never run anywhere (this round's boxes have one GPU; the
    // one-rank communicator sending to itself captures and replays fine): a multi-rank run takes the graph path only when
    // asked to ("graph" 2) — a refused capture falls back, a hang in an untested collective path would not.
    if (c->nranks > 1 && c->use_graph < 2) return false;
    if (!c->deep || deep_depth(c->deep) > GR || c->mid_pair || c->overlap == 2) return false;      // (pairs of launches: four groups = two whole pairs)
    if (!(face_south(c) || face_north(c))) return false;
    const int depth = deep_depth(c->deep), iters = GRAPH_GROUPS * depth;
    if (remaining < iters + 4 * depth + 1) return false;                       // (plan_launch splits the END of a segment differently)
    if (of > 0 && (c->steps_done % of == 0 || of - c->steps_done % of < iters + 4 * depth + 1)) return false;
    return true;
}

// Replay (capturing first, if need be) GRAPH_GROUPS launch groups. Returns the iterations advanced, 0 if the graph path is
// not available (the caller issues eagerly), < 0 on error.
template <typename T>
int replay_groups(lbm_ctx* c, int remaining, int of, bool transport) {
    const int key[6] = {c->cur, c->overlap, c->deep_halo, c->deep, c->use_nt, c->skip_exchange};
    if (c->gexec && memcmp(key, c->gkey, sizeof(key)) != 0) graph_drop(c);
    // everything queued so far, on both streams, precedes the graph: join the side stream into the main one
    int rc = join_comm(c);
    if (rc) return rc;
    if (c->tbase_host != c->steps_done) {      // the graph's launches carry iteration numbers relative to the device word
        hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, c->d_tbase, c->steps_done);
        c->tbase_host = c->steps_done;
    }
    if (!c->gexec) {
        struct Saved { int cur, steps_done; long launches_total; bool comm_issued, mid_pair, last_was_pair, ext_split, restored; int e0, e1;
                       hipEvent_t ev_main, ev_edge, ev_comm; } sv{c->cur, c->steps_done, c->launches_total, c->comm_issued, c->mid_pair,
                       c->last_was_pair, c->ext_split_pending, c->restored, c->edge_rows[0], c->edge_rows[1], c->ev_main, c->ev_edge, c->ev_comm};
        auto restore = [&]() {
            c->cur = sv.cur; c->steps_done = sv.steps_done; c->launches_total = sv.launches_total; c->comm_issued = sv.comm_issued;
            c->mid_pair = sv.mid_pair; c->last_was_pair = sv.last_was_pair; c->ext_split_pending = sv.ext_split; c->restored = sv.restored;
            c->edge_rows[0] = sv.e0; c->edge_rows[1] = sv.e1; c->ev_main = sv.ev_main; c->ev_edge = sv.ev_edge; c->ev_comm = sv.ev_comm;
        };
        // the capture records and waits for its OWN events, and starts with nothing to wait for (joined above)
        c->ev_main = c->gev_main; c->ev_edge = c->gev_edge; c->ev_comm = c->gev_comm;
        c->comm_issued = false; c->ext_split_pending = false;
        hipGraph_t graph = nullptr;
        auto note = [&](const char* what, hipError_t e) {
            if (!c->graph_note[0]) snprintf(c->graph_note, sizeof(c->graph_note), "%s: %s", what, e == hipSuccess ? g_err : hipGetErrorString(e));
        };
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        bool ok = e == hipSuccess;
        if (!ok) note("hipStreamBeginCapture", e);
        int iters = 0;
        if (ok) {
            for (int g = 0; g < GRAPH_GROUPS && ok; ++g) {
                const int took = advance<T>(c, remaining - iters, of, transport, true);
                ok = took == deep_depth(c->deep);
                if (!ok) { char b[64]; snprintf(b, sizeof(b), "group %d took %d iterations", g, took); note(b, hipSuccess); }
                iters += took > 0 ? took : 0;
            }
            if (ok) { ok = join_comm(c) == LBM_OK; if (!ok) note("join", hipSuccess); }   // the side stream rejoins the origin of the capture
            if (ok) hipLaunchKernelGGL(k_add_int, dim3(1), dim3(1), 0, c->stream, c->d_tbase, iters);
            e = hipStreamEndCapture(c->stream, &graph);
            if (ok && (e != hipSuccess || !graph)) note("hipStreamEndCapture", e);
            ok = ok && e == hipSuccess && graph != nullptr;
        }
        (void)hipGetLastError();
        if (ok) { e = hipGraphInstantiate(&c->gexec, graph, nullptr, nullptr, 0); ok = e == hipSuccess; if (!ok) note("hipGraphInstantiate", e); }
        if (graph) (void)hipGraphDestroy(graph);
        const int cur_after = c->cur;
        restore();
        if (ok && cur_after != sv.cur) { ok = false; note("odd number of buffer flips", hipSuccess); }
        if (!ok) {        // refused: eager from now on
            graph_drop(c);
            c->graph_failed = true;
            (void)hipGetLastError();
            return 0;
        }
        memcpy(c->gkey, key, sizeof(key));
        c->giters = iters;
    }
    HIPCHK(hipGraphLaunch(c->gexec, c->stream));
    // the host-side state as the eager path would have left it; every stream of the graph was joined into the main one
    c->steps_done += c->giters;
    c->tbase_host += c->giters;
    c->launches_total += GRAPH_GROUPS;
    c->last_was_pair = true;
    c->restored = false;
    c->comm_issued = false;
    c->ext_split_pending = false;
    c->mid_pair = false;
    c->graph_replays++;
    return c->giters;
}

// `nsteps` iterations of n strips driven in lockstep by this thread (n == 1: a context on its own, which may talk to
// other PROCESSES through its RCCL communicator). Per launch: every member's part before the exchange, the exchange,
// every member's part after it.
template <typename T>
int do_steps(lbm_ctx** cs, int n, int nsteps, int of) {
    for (int i = 0; i < n; ++i) {
        lbm_ctx* c = cs[i];
        SETDEV(c);
        if (c->timing) HIPCHK(hipEventRecord(c->ev_t0, c->stream));
        if (c->steps_done != cs[0]->steps_done) return fail(LBM_ERR_ARG, "the strips of a group are at different iterations");
    }
    lbm_ctx* c0 = cs[0];
    const bool transport = n > 1 || (c0->comm && c0->nranks > 1) || c0->loopback;   // a device transport is attached
    int launches = 0;
    std::vector<Launch> L((size_t)n);
    if (n > 1 && c0->group_threads && c0->pool) {
        // One host thread per strip: a launch costs a strip ~10 runtime calls (kernels, events, copies), which one thread
        // issuing for 8 GPUs in turn cannot hide behind 20 us kernels. Three rendezvous per launch: every strip's
        // ev_edge is recorded before anybody pulls, every pull is queued before anybody records ev_comm / launches the
        // interior, and every ev_comm is recorded before the next launch looks at its neighbours'.
        // Between two rendezvous NO path returns: every error of a strip goes through report() and the whole group leaves at the
        // next rendezvous, decided once by its last arrival (GroupPool). The job's state lives on the heap, owned by the closure:
        // a thread that comes back after the call has timed out (LBM_ERR_TIMEOUT) still finds it.
        struct Job { std::vector<lbm_ctx*> cs; std::vector<Launch> L; int n, nsteps, of; bool transport; int launches = 0; };
        auto J = std::make_shared<Job>();
        J->cs.assign(cs, cs + n); J->L.resize((size_t)n); J->n = n; J->nsteps = nsteps; J->of = of; J->transport = transport;
        const std::function<void(GroupPool::State&, int)> worker = [J](GroupPool::State& P, int i) {
            lbm_ctx** cs = J->cs.data();
            std::vector<Launch>& L = J->L;
            const int n = J->n, nsteps = J->nsteps, of = J->of;
            const bool transport = J->transport;
            lbm_ctx* c = cs[i];
            lbm_ctx* c0 = cs[0];
            (void)hipSetDevice(c->device);
            auto fault = [&](int launch, int point) -> int {       // TEST ONLY ("debug_fault_*"): this strip fails or sleeps here
                if (c->debug_fault_launch != launch || c->debug_fault_point != point) return LBM_OK;
                c->debug_fault_launch = -1;
                if (c->debug_fault_stall_ms > 0) { std::this_thread::sleep_for(std::chrono::milliseconds(c->debug_fault_stall_ms)); return LBM_OK; }
                return fail(LBM_ERR_HIP, "injected fault: strip %d, launch %d, point %d", i, launch, point);
            };
            int launch = 0;
            for (int k = 0; k < nsteps; ++launch) {
                const int t = c->steps_done;
                int rc = LBM_OK;
                if (of > 0 && t % of == 0) {
                    if (c->log_count >= c->log_cap) rc = fail(LBM_ERR_ARG, "force log full (%d rows): drain it", c->log_cap);
                    if (!rc) rc = join_comm(c);
                    if (!rc) rc = launch_forces<T>(c, c->d_force_log + 3L * c->log_count, t);
                    if (!rc) c->log_count++;
                }
                if (!rc) rc = plan_launch(c, nsteps - k, of, transport, true, &L[(size_t)i]);
                if (!rc) rc = issue_before<T>(c, L[(size_t)i]);
                if (!rc) rc = fault(launch, 0);
                P.report(rc, g_err);
                if (P.arrive(i, launch, 1)) return;
                if (L[(size_t)i].depth != L[0].depth || L[(size_t)i].kind != L[0].kind)
                    P.report(fail(LBM_ERR_ARG, "the strips of a group disagree on the next launch (different options?)"), g_err);
                else if (L[0].kind == KIND_EXCHANGE) {
                    int rc2 = LBM_OK;
                    if (c0->group_transport == 0) rc2 = pull_halos<T>(cs, n, i, L[0].dst);
                    else if (i == 0) rc2 = exchange_group<T>(cs, n, L[0].dst);      // RCCL: one group call, one thread
                    P.report(rc2, g_err);
                }
                P.report(fault(launch, 1), g_err);
                if (P.arrive(i, launch, 2)) return;
                P.report(issue_after<T>(c, L[(size_t)i]), g_err);
                P.report(fault(launch, 2), g_err);
                if (P.arrive(i, launch, 3)) return;
                k += L[(size_t)i].depth;       // (its own copy: strip 0 may already be planning the next launch into L[0])
                if (i == 0) ++J->launches;
            }
        };
        const int prc = c0->pool->run(worker, c0->wait_timeout_ms);
        launches = J->launches;
        if (prc != LBM_OK) return prc;
        HIPCHK(hipSetDevice(c0->device));
    } else
    for (int k = 0; k < nsteps;) {
        const int t = c0->steps_done;
        if (n == 1 && graph_wanted(c0, nsteps - k, of, transport)) {
            const int took = replay_groups<T>(c0, nsteps - k, of, transport);
            if (took < 0) return took;
            if (took > 0) { k += took; launches += GRAPH_GROUPS; continue; }
        }
        for (int i = 0; i < n; ++i) {
            lbm_ctx* c = cs[i];
            SETDEV(c);
            if (of > 0 && t % of == 0) {
                if (c->log_count >= c->log_cap) return fail(LBM_ERR_ARG, "force log full (%d rows): drain it", c->log_cap);
                int rc = join_comm(c);      // the edge bands of the previous launch live on the side stream
                if (rc) return rc;
                rc = launch_forces<T>(c, c->d_force_log + 3L * c->log_count, t);
                if (rc) return rc;
                c->log_count++;
            }
            int rc = plan_launch(c, nsteps - k, of, transport, true, &L[i]);
            if (rc) return rc;
            if (L[i].depth != L[0].depth || L[i].kind != L[0].kind)
                return fail(LBM_ERR_ARG, "the strips of a group disagree on the next launch (different options?)");
            rc = issue_before<T>(c, L[i]);
            if (rc) return rc;
        }
        if (L[0].kind == KIND_EXCHANGE) {
            int rc = n > 1 ? exchange_group<T>(cs, n, L[0].dst) : exchange_rccl<T>(c0, L[0].dst, exchange_stream(c0));
            if (rc) return rc;
        }
        for (int i = 0; i < n; ++i) {
            SETDEV(cs[i]);
            int rc = issue_after<T>(cs[i], L[i]);
            if (rc) return rc;
        }
        k += L[0].depth;
        ++launches;
    }
    for (int i = 0; i < n; ++i) {
        lbm_ctx* c = cs[i];
        if (!c->timing) continue;
        HIPCHK(hipSetDevice(c->device));
        int jr = join_comm(c);
        if (jr) return jr;
        HIPCHK(hipEventRecord(c->ev_t1, c->stream));
        c->timed_launches = launches;
        c->timed_steps = nsteps;
    }
    return LBM_OK;
}

// The halos of a freshly initialised / restored group: every member's edge rows of buf[cur] to its neighbours.
template <typename T>
int refresh_group_halos(lbm_ctx** cs, int n) {
    for (int i = 0; i < n; ++i) {
        lbm_ctx* c = cs[i];
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipStreamSynchronize(c->comm_stream));
        HIPCHK(hipEventRecord(c->ev_edge, exchange_stream(c)));
        if (c->cur != cs[0]->cur) return fail(LBM_ERR_ARG, "the strips of a group are in different buffer phases");
    }
    int rc = exchange_group<T>(cs, n, cs[0]->cur);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        lbm_ctx* c = cs[i];
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipEventRecord(c->ev_comm, exchange_stream(c)));
        c->comm_issued = true;
        c->mid_pair = false;
    }
    for (int i = 0; i < n; ++i) {
        HIPCHK(hipSetDevice(cs[i]->device));
        HIPCHK(hipStreamSynchronize(exchange_stream(cs[i])));
    }
    return LBM_OK;
}

template <typename T>
int do_macros(lbm_ctx* c, bool want_max) {
    const size_t n = (size_t)c->nx * c->nyl;
    if (!c->d_macro) HIPCHK(hipMalloc(&c->d_macro, 3 * n * sizeof(double)));
    MacroArgs<T> m;
    m.old = static_cast<const T*>(c->buf[c->cur ^ 1]);
    m.plane = (long)c->plane; m.pitch = c->pitch; m.xoff = c->xoff;
    m.nx = c->nx; m.ny_loc = c->nyl; m.ny_glob = c->p.ny; m.y_start = c->p.y_start;
    m.cyl_x = c->cyl_x; m.cyl_y = c->cyl_y; m.cyl_r2 = (double)(c->cyl_r * c->cyl_r);
    m.u_in = (T)c->p.inlet_velocity;
    m.initial = (c->steps_done == 0);
    m.rho = c->d_macro; m.ux = c->d_macro + n; m.uy = c->d_macro + 2 * n;
    m.max_usq_bits = want_max ? c->d_maxbits : nullptr;
    if (want_max) HIPCHK(hipMemsetAsync(c->d_maxbits, 0, sizeof(unsigned long long), c->stream));
    dim3 grid((c->nx + 255) / 256, c->nyl), block(256);
    hipLaunchKernelGGL((k_macros<T>), grid, block, 0, c->stream, m);
    HIPCHK(hipGetLastError());
    return LBM_OK;
}

template <typename T>
int do_populations(lbm_ctx* c, int which, double* aos) {
    const int tnx = c->nx + 2, tny = c->nyl + 2;
    const void* srcbuf = c->buf[c->cur ^ 1];
    const bool initial = (c->steps_done == 0);
    if (which == 0 && !initial) {
        if (!c->scratch) HIPCHK(hipMalloc(&c->scratch, buffer_bytes(c)));
        KArgs<T> a = make_kargs<T>(c, c->cur ^ 1, c->cur ^ 1, 0);
        a.dst = static_cast<T*>(c->scratch);
        launch_rows<T, MODE_STREAM_ONLY>(c, a, c->stream);
        HIPCHK(hipGetLastError());
        srcbuf = c->scratch;
    }
    std::vector<T> host(c->total);
    HIPCHK(hipMemcpyAsync(host.data(), srcbuf, c->total * c->esize, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int gy = 0; gy < tny; ++gy)
        for (int gx = 0; gx < tnx; ++gx) {
            const bool ghost = (gy == 0 || gy == tny - 1 || gx == 0 || gx == tnx - 1);
            double* o = aos + ((size_t)gy * tnx + gx) * Q;
            // ghost cells of f_current are never written after Grid::initialise (LBMGrid.h:196-213); those of
            // f_next keep the initial equilibrium until the first exchange_ghost_cells
            const bool analytic = ghost && (which == 0 || initial);
            for (int i = 0; i < Q; ++i)
                o[i] = analytic ? (double)(T)c->feq_in[i]
                                : (double)host[(size_t)i * c->plane + (size_t)(gy + GR - 1) * c->pitch + c->xoff + gx - 1];
        }
    return LBM_OK;
}

template <typename T>
int do_halo_export(lbm_ctx* c, double* south_out, double* north_out) {
    const T* base = static_cast<const T*>(c->buf[c->cur]);
    const size_t n = (size_t)HR1 * Q * c->nx;
    dim3 grid((c->nx + 255) / 256, HR1 * Q), block(256);
    if (south_out) {   // my bottom HR1 interior rows
        hipLaunchKernelGGL((k_halo_pack<T>), grid, block, 0, c->stream, base, (long)c->plane, c->pitch, c->xoff, c->nx, GR,
                           c->d_halo);
        HIPCHK(hipMemcpyAsync(south_out, c->d_halo, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    if (north_out) {   // my top HR1 interior rows
        hipLaunchKernelGGL((k_halo_pack<T>), grid, block, 0, c->stream, base, (long)c->plane, c->pitch, c->xoff, c->nx,
                           c->nyl + GR - HR1, c->d_halo + n);
        HIPCHK(hipMemcpyAsync(north_out, c->d_halo + n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return LBM_OK;
}

template <typename T>
int do_halo_import(lbm_ctx* c, const double* south_in, const double* north_in) {
    T* base = static_cast<T*>(c->buf[c->cur]);
    const size_t n = (size_t)HR1 * Q * c->nx;
    dim3 grid((c->nx + 255) / 256, HR1 * Q), block(256);
    if (south_in) {    // -> south ghost rows gy = GR-HR1 .. GR-1
        HIPCHK(hipMemcpyAsync(c->d_halo + 2 * n, south_in, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL((k_halo_unpack<T>), grid, block, 0, c->stream, base, (long)c->plane, c->pitch, c->xoff, c->nx, GR - HR1,
                           c->d_halo + 2 * n);
    }
    if (north_in) {    // -> north ghost rows gy = nyl+GR .. nyl+GR+HR1-1
        HIPCHK(hipMemcpyAsync(c->d_halo + 3 * n, north_in, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL((k_halo_unpack<T>), grid, block, 0, c->stream, base, (long)c->plane, c->pitch, c->xoff, c->nx,
                           c->nyl + GR, c->d_halo + 3 * n);
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return LBM_OK;
}

// Grid::f_current(x,y,i) written by a client (LBMGrid.h:115): the pre-collision state of the next iteration. The interior
// cells of `aos` are packed into the scratch buffer and re-collided into buf[cur] (collision_step skips solid cells).
template <typename T>
int do_set_f_current(lbm_ctx* c, const double* aos) {
    if (!c->scratch) HIPCHK(hipMalloc(&c->scratch, buffer_bytes(c)));
    std::vector<T> host(c->total);
    const int tnx = c->nx + 2;
    for (int y = 0; y < c->nyl; ++y)
        for (int x = 0; x < c->nx; ++x) {
            const double* v = aos + ((size_t)(y + 1) * tnx + (x + 1)) * Q;
            for (int i = 0; i < Q; ++i) host[(size_t)i * c->plane + (size_t)(y + GR) * c->pitch + c->xoff + x] = (T)v[i];
        }
    HIPCHK(hipMemcpyAsync(c->scratch, host.data(), c->total * c->esize, hipMemcpyHostToDevice, c->stream));
    KArgs<T> a = make_kargs<T>(c, c->cur, c->cur, c->steps_done);
    a.src = static_cast<const T*>(c->scratch);
    launch_rows<T, MODE_COLLIDE_ONLY>(c, a, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LBM_OK;
}

#define DISPATCH(c, call_d, call_f) ((c)->p.precision == LBM_PRECISION_F32 ? (call_f) : (call_d))

