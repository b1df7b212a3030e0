/* include/lbm_hip.h — C-ABI of the MI355X-native D2Q9-BGK timestep (liblbm_hip.so).
 *
 * The reference (LGMOak/HighPerformanceComputing-LatticeBoltzmannMethod) has no FFI/plugin seam: it is a
 * header-only C++ class library driven by src/main.cpp:11-20. The seam below is therefore the set of calls
 * the reference's own classes make on the hot path, one entry point per reference member, so that
 * LBM::Solver / LBM::Grid / LBM::IOManager can be re-hosted on it (see INTEGRATION.md and the C++ mirror in
 * highperformancecomputing-latticeboltzmannmethod_amd/host/). Plain C types only; all output buffers are
 * caller-owned host memory; all device memory is owned by the context; no exceptions cross the boundary.
 *
 * Return convention: 0 = ok; LBM_ERR_* (<0) = failure, text via lbm_last_error(). There is NO CPU fallback:
 * without a usable HIP device every entry point that touches state fails with LBM_ERR_HIP.
 *
 * Time convention: a context has completed `steps_done` loop bodies of Solver::run (LBMSolver.h:48-76).
 * The device holds the post-collision populations of loop body t = steps_done (the reference's f_next right
 * after collision_step() of iteration t) and those of iteration t-1, so
 *   lbm_get_forces          == IOManager::record_forces(t = steps_done, ...)      (LBMIO.h:114-168)
 *   lbm_get_macros          == rho/ux/uy as left by iteration steps_done-1         
 *   lbm_get_populations     == f_current / f_next as left by iteration steps_done-1
 */
#ifndef LBM_HIP_H
#define LBM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_OK            0
#define LBM_ERR_ARG      -1   /* bad argument / bad state */
#define LBM_ERR_HIP      -2   /* HIP runtime error or no device */
#define LBM_ERR_COMM     -3   /* RCCL error / communicator not initialised */
#define LBM_ERR_ALLOC    -4
#define LBM_ERR_TIMEOUT  -5   /* a host-side wait of the library ran into its bound (option "wait_timeout_ms", LBM_WAIT_TIMEOUT_MS): the
                               * text names who was waited for and where; a group that timed out is unusable, destroy it */

#define LBM_PRECISION_F64 0
#define LBM_PRECISION_F32 1   /* build-only variant; the reference has no fp32 path */

/* Physics: field-for-field LBM::SimulationParams (LBMConfig.h:36-51) minus the run-control fields, which
 * stay with the host-side Solver. Strip: the row strip [y_start, y_start+local_ny) of the global ny rows this
 * context owns (replaces Grid::initialise_2d_topology, LBMGrid.h:347-364; whole domain: y_start=0,
 * local_ny=ny). */
typedef struct lbm_params {
    double tau;              /* LBMConfig.h:37 */
    double inlet_velocity;   /* LBMConfig.h:38 */
    int    nx, ny;           /* LBMConfig.h:39-40, global interior size */
    double cylinder_x;       /* LBMConfig.h:46, fraction of nx */
    double cylinder_y;       /* LBMConfig.h:47, fraction of ny */
    double cylinder_radius;  /* LBMConfig.h:48, fraction of ny */
    int    y_start;          /* first global row of this strip */
    int    local_ny;         /* rows in this strip (0 => ny - y_start) */
    int    precision;        /* LBM_PRECISION_* */
    int    force_log_capacity; /* rows of the device-resident force log (0 => 4096) */
} lbm_params;

typedef struct lbm_ctx lbm_ctx;

typedef struct lbm_force_row { int timestep; double fx, fy; } lbm_force_row;

const char* lbm_last_error(void);
int  lbm_device_count(void);

/* Grid::Grid (LBMGrid.h:57-90): creates the context (streams, logs); the two SoA population buffers are allocated by
 * lbm_initialise, whose measured plan decides their layout. */
int  lbm_create(const lbm_params* p, int device, lbm_ctx** out);
void lbm_destroy(lbm_ctx* c);

/* Grid::setup_geometry + Grid::initialise (LBMGrid.h:152-246) followed by collision_step() of iteration 0
 * (LBMSolver.h:84-126). *solid_count_out (nullable) = solid cells in this strip (LBMGrid.h:158-175). */
int  lbm_initialise(lbm_ctx* c, int* solid_count_out);

/* `nsteps` loop bodies of Solver::run (LBMSolver.h:49-60): exchange + stream + BCs + stability of iteration
 * t, fused with collision_step() of iteration t+1; up to six (eight on grids of a single round of blocks) consecutive
 * iterations share one kernel launch, as many as the measured plan fuses
 * (intermediate states stay in LDS; results are bit-identical to one launch per iteration). Asynchronous on the
 * context's streams. If output_frequency > 0, record_forces is evaluated on-device for every
 * t % output_frequency == 0 inside the range (LBMSolver.h:52-54) and appended to the force log. The last iteration
 * of a call is always a launch of its own, so that the snapshots below refer to iteration steps_done-1. With a
 * communicator attached (lbm_comm_init) the strip's six edge rows per face are exchanged once per launch of five / six
 * iterations (strips of 64 rows or more) or once per two launches of up to three (replaces
 * Grid::exchange_ghost_cells, LBMGrid.h:249-283). */
int  lbm_step(lbm_ctx* c, int nsteps, int output_frequency);

/* Waits for everything queued on the context's two streams. Default: the blocking hipStreamSynchronize (the fastest way to wait; a
 * watchdog thread prints one "lbm_hip: STALL: ..." line on stderr — strip, stream, iteration — if it outlives LBM_WAIT_TIMEOUT_MS, five
 * minutes by default). With the option "wait_timeout_ms" set: a bounded poll that returns LBM_ERR_TIMEOUT instead of blocking on. The
 * reference's counterpart is the implicit completion of every loop body (it is synchronous, LBMSolver.h:48-76). */
int  lbm_sync(lbm_ctx* c);
int  lbm_steps_done(const lbm_ctx* c);

/* Grid::check_stability (LBMGrid.h:285-317), evaluated inside every step kernel; returns the first
 * iteration t whose post-BC populations were non-finite or outside [-1e5, 1e5] (the t the reference prints at
 * LBMSolver.h:62), or -1. Synchronises. */
int  lbm_first_unstable_step(lbm_ctx* c, int* t_out);

/* IOManager::record_forces (LBMIO.h:133-168) for t = steps_done: this strip's partial sums. Synchronises. */
int  lbm_get_forces(lbm_ctx* c, double* fx, double* fy);
/* Rows appended by lbm_step (this strip's partial sums); returns the number copied, clears the log. */
int  lbm_drain_force_log(lbm_ctx* c, lbm_force_row* rows, int max_rows);

/* rho/ux/uy of this strip, row-major [local_ny][nx] (LBMGrid.h:109-111; the layout gathered at
 * LBMSolver.h:340-357). Any pointer may be NULL. Synchronises. */
int  lbm_get_macros(lbm_ctx* c, double* rho, double* ux, double* uy);
/* Grid::max_velocity (LBMGrid.h:319-344) before the cross-rank MAX and sqrt: max(ux^2+uy^2) of this strip. */
int  lbm_max_velocity_sq(lbm_ctx* c, double* out);

/* Debug/parity accessor: ghost-inclusive AoS [(local_ny+2)][(nx+2)][9] exactly as Grid::f_current /
 * Grid::f_next index it (LBMGrid.h:105-107,116-119). which: 0 = f_current, 1 = f_next. */
int  lbm_get_populations(lbm_ctx* c, int which, double* aos);
/* The write side of Grid::f_current (LBMGrid.h:115: `double& f_current(x,y,i)`): replaces the pre-collision state of the
 * next iteration by the interior cells of `aos` (same ghost-inclusive layout as lbm_get_populations; ghost entries are
 * ignored, they belong to the halo logic) and redoes collision_step() on it. Custom initial conditions enter here.
 * Until the next lbm_step the snapshots still show the values before the call. Synchronises. */
int  lbm_set_f_current(lbm_ctx* c, const double* aos);
/* Grid::is_solid (LBMGrid.h:146-148) for this strip, [local_ny][nx] bytes. */
int  lbm_get_solid(lbm_ctx* c, unsigned char* mask);

/* ---- strip halo exchange (replaces Grid::exchange_ghost_cells, LBMGrid.h:249-283) ----
 * Device path: RCCL send/recv of the LBM_HALO_ROWS edge rows per face (one contiguous run in the row-interleaved
 * layout) once per launch of up to six iterations (or 2 x LBM_HALO_ROWS rows once per two such launches: option "deep_halo" 2,
 * measured at lbm_initialise), on a side stream, overlapped with the interior update. `id128` is the 128-byte
 * ncclUniqueId produced by lbm_comm_unique_id on rank 0 and distributed by the launcher. Ranks are ordered bottom (0)
 * to top; attach the communicator before lbm_initialise. */
int  lbm_comm_unique_id(void* id128);
int  lbm_comm_init(lbm_ctx* c, int rank, int nranks, const void* id128);
/* Sum the partial force sums / max / min across strips (the reference's MPI_Reduce/MPI_Allreduce at
 * LBMIO.h:167-168, LBMGrid.h:315,342). In place, host values, n doubles. op: 0 sum, 1 max, 2 min. */
int  lbm_comm_allreduce(lbm_ctx* c, double* vals, int n, int op);
/* In-process strips (replaces the decomposition Grid::initialise_2d_topology builds, LBMGrid.h:347-392, when ONE process
 * drives several GPUs, e.g. `lbm_solver --gpus 8`): n contexts created for consecutive strips (bottom to top, covering
 * all ny rows; any devices) are linked into a group and then initialised and stepped together by the calling thread.
 * transport 0: every strip pulls its neighbours' edge rows with hipMemcpyPeerAsync over xGMI (plain device copies when
 * two strips share a device) on its side stream, behind the neighbour's edge-rows event; transport 1: RCCL, one
 * communicator per member (ncclCommInitAll; distinct devices), all members' ncclSend/ncclRecv in one group call.
 * The launch sequence, the exchange cadence and the overlap are those of lbm_step. Per-strip results
 * (lbm_get_macros, lbm_drain_force_log, lbm_first_unstable_step, ...) are read member by member and combined by the
 * caller (sum of forces, min of the unstable step, rows concatenated by y_start: LBMSolver.h:269-362, LBMIO.h:167-168).
 * lbm_group_refresh_halos: after lbm_load_state on every member.
 * Errors: a member's failure (or LBM_ERR_TIMEOUT: a strip thread that did not reach a rendezvous within "wait_timeout_ms") makes
 * lbm_group_step return that error on the caller with every strip thread parked again; the launch in flight was abandoned half-way,
 * so the populations are undefined until lbm_group_initialise (or lbm_load_state + lbm_group_refresh_halos) — after a time-out the
 * group refuses further work and is only good for lbm_destroy. */
int  lbm_group_link(lbm_ctx** ctxs, int n, int transport);
int  lbm_group_initialise(lbm_ctx** ctxs, int n, int* solid_total_out);
int  lbm_group_step(lbm_ctx** ctxs, int n, int nsteps, int output_frequency);
int  lbm_group_refresh_halos(lbm_ctx** ctxs, int n);
/* What this process actually bound at run time (another library loaded first may have brought its own RCCL / HIP):
 * ncclGetVersion, hipRuntimeGetVersion, hipDriverGetVersion. Any pointer may be NULL. */
int  lbm_runtime_versions(int* rccl, int* hip_runtime, int* hip_driver);
/* hipMemGetInfo of a device (leak checks; sizing of strips). Any pointer may be NULL. */
int  lbm_device_memory(int device, unsigned long long* free_bytes, unsigned long long* total_bytes);
/* The exchange schedule of a strip with a communicator ("overlap=.. deep_halo=.. (..)"): measured at lbm_initialise
 * over the four schedules (collective; MAX over the ranks) unless pinned with lbm_set_option. */
const char* lbm_strip_schedule(const lbm_ctx* c);
/* Host-staged path (the buffers the reference hands to MPI_Isend/Irecv, LBMGrid.h:255-276). Each face buffer is
 * [LBM_HALO_ROWS][9][nx] doubles: the LBM_HALO_ROWS (= 6) interior rows next to that face, bottom row first, all nine
 * populations (six rows: one launch of up to six fused iterations — or two launches of up to three, the first one
 * recomputing three of the neighbour's rows — may run between two exchanges).
 * export: south_out = my bottom rows, north_out = my top rows; import: south_in -> my south ghost rows (= the south
 * neighbour's north_out), north_in -> my north ghost rows. NULL = that side is a physical wall. The caller exchanges
 * after lbm_initialise and after EVERY lbm_step call, and calls lbm_step so that it issues at most TWO launches
 * (e.g. nsteps <= 4 by default = a fused launch of three iterations + a single one, or nsteps = 6 with the option
 * "trailing_pair" 1; with a "deep" plan ONE launch of up to six iterations; lbm_step refuses a call that would need
 * more). Used by
 * MPI-hosted callers and by the 2-rank tests. */
#define LBM_HALO_ROWS 6
int  lbm_halo_export(lbm_ctx* c, double* south_out, double* north_out);
int  lbm_halo_import(lbm_ctx* c, const double* south_in, const double* north_in);

/* Checkpoint / restart (the reference has none): the strip's post-collision populations and the
 * iteration counter. lbm_load_state needs an initialised context created with the same parameters; the macro /
 * population snapshots become available again after the next lbm_step. */
int  lbm_save_state(lbm_ctx* c, const char* path);
int  lbm_load_state(lbm_ctx* c, const char* path);

/* Tuning/diagnostics (not part of the reference surface). Keys, all to be set before lbm_initialise:
 *   "tune" 1|0    time the candidate plans at lbm_initialise and keep the fastest (default 1); with 0 the plan is
 *                 "layout" 0 planar|1 row-interleaved,
 *                 "nt" non-temporal stores, "ntl" non-temporal level-1 loads of the register kernel ("deep" 6 / 7),
 *                 "alternate" alternate the row walk direction per launch,
 *                 "fuse" 1|2|3|4 iterations fused per launch through LDS (k_step2_tile / k_step3_tile / k_step4_tile, the
 *                 last only for a context without strip faces;
 *                 "pair" 1 == "fuse" 2), "pair_ty" 8|12 tile height, "xcd" XCD-aware tile walk,
 *                 "trailing_pair" 1 lets an lbm_step call end on a fused launch (snapshots then need one more step)
 *                 "deep" 1..3, 6..9: more iterations per launch. 1 / 2 / 3: six / seven / eight iterations on an LDS-filling
 *                 64x16 / 64x16 / 32x32 tile (k_stepd_tile, 1024-thread blocks: grids of a single round of blocks); 6 / 7:
 *                 five / six iterations with the lattice of a 64x32 region held in registers (k_stepc_col, 512-thread
 *                 blocks, two per CU: large grids; a plan of this family uses both depths to split a call without a slow
 *                 tail; 9: the same with seven iterations as the plan's depth: the largest grids); 8 (fp32 contexts only, LBM_ERR_ARG otherwise): seven iterations on a 64x48 region in registers
 *                 (twelve waves x four rows; six / eight iterations for what a call leaves over): large fp32 grids.
 *                 Strips use 1, 6, 7 by rule with one exchange per launch; 4 / 5 (round 2's 32x16 LDS tiles) are retired,
 *                 "arith" 0 strict IEEE collision (bit-identical to the CPU oracle for normal-range operands: lbm_debug_strict_div2) | 1 FMA-contracted (<= 1e-10)
 *   strips:       "overlap" 0 launch and exchange serialised | 1 edge bands first, the exchange overlapped with the interior
 *                 rows of the SAME launch | 2 the exchange overlapped with the interior rows of the NEXT (extended)
 *                 launch; "deep_halo" 0 an exchange of LBM_HALO_ROWS rows after every launch | 1 after every second launch of a
 *                 plan of up to three iterations per launch (a deep plan still exchanges after every launch) | 2 a deep plan too:
 *                 2 x LBM_HALO_ROWS rows after every second launch of up to six iterations (both measured at lbm_initialise
 *                 when a communicator is attached, unless set here),
 *                 "halo_trim" 0 an exchange carries all nine populations of every row of a face in ONE contiguous message | 1 only the
 *                 sub-rows the receiver's launches read (the outermost row's three inbound populations, the next row's six, every
 *                 other row's nine: 9 hr - 9 of 9 hr sub-rows, five messages per face); measured at lbm_initialise like the rest,
 *                 "skip_exchange" 1 (diagnostic: no halo traffic, results invalid),
 *                 "wait_timeout_ms" bound of every host-side wait (rendezvous of a group's threads, lbm_sync, the drains of
 *                 lbm_destroy; 0 = LBM_WAIT_TIMEOUT_MS or five minutes): LBM_ERR_TIMEOUT names who was waited for,
 *                 "graph" 0|1|2 replay the launch groups of a deep strip plan from a captured hipGraph: 1 (default) where the
 *                 transport is local to the process, 2 also between the ranks of a communicator (RCCL under capture:
 *                 exercised with a one-rank communicator only so far)
 *   "timing" 1    record HIP events around each lbm_step call (lbm_last_step_kernel_ms). */
int  lbm_set_option(lbm_ctx* c, const char* key, long value);
/* Average device time per step-kernel launch (ms) measured with HIP events on the context's stream around
 * the last lbm_step call; 0 if events were not enabled via lbm_set_option(c, "timing", 1). */
int  lbm_last_step_kernel_ms(lbm_ctx* c, double* ms_per_launch);
/* Same measurement, unreduced: device milliseconds of the last lbm_step call, the step-kernel launches it issued and
 * the iterations it advanced (a fused launch advances two to eight). */
int  lbm_last_step_stats(lbm_ctx* c, double* ms_total, int* launches, int* iterations);
/* How many times a captured hipGraph of four launch groups has been replayed for this context so far (a strip with a device
 * transport on a deep plan replays its launch groups instead of issuing them call by call; option "graph" 0 turns that off;
 * 0 also where the capture was refused and the eager path runs). No reference counterpart: the reference has no GPU path. */
long lbm_graph_replays(const lbm_ctx* c);
const char* lbm_kernel_name(const lbm_ctx* c);
/* The plan lbm_initialise settled on (layout / kernel / store policy / traversal), for logs; where it was measured, with the
 * finalists' times (median of three windows each) so that the margin of the choice is visible. */
const char* lbm_plan(const lbm_ctx* c);
/* The same plan as lbm_set_option pairs, "layout=1 nt=0 alternate=1 pair_ty=12 xcd=1 deep=7": set
 * on a fresh context together with "tune" 0 they reproduce the plan in another process (bench.py's counter passes run
 * the benchmarked plan in a child process under rocprofv3). No reference counterpart. */
const char* lbm_plan_options(const lbm_ctx* c);
/* Test hook (device 0): the strict collision's two divisions by rho (LBMSolver.h:108-109) share one reciprocal chain
 * (csrc/lbm_kernels.hpp strict_div2); this runs it beside the compiler's IEEE divisions: q1,q2 = strict_div2(a1,a2,b), r1,r2 = a1/b,
 * a2/b, n host doubles each. Bit-identical for denominators in [2^-20, 2^20] and numerators 0 or of magnitude in [2^-400, 2^400]. */
int lbm_debug_strict_div2(const double* a1, const double* a2, const double* b, int n, double* q1, double* q2, double* r1, double* r2);
/* Test hook, callable without a device: what the ranks of a strip run agree on before the collective schedule trials of
 * lbm_initialise (csrc/lbm_hip.hip tune_strip_schedule). per_rank7 = nranks x {may tune, overlap pinned, overlap, deep_halo pinned,
 * deep_halo, halo_trim pinned, halo_trim}; agreed7 = the same seven as agreed (-1 where nothing is pinned). LBM_ERR_ARG when the ranks pin different
 * schedules (they would otherwise run different numbers of collective trials). Replaces nothing in the reference. */
int lbm_debug_strip_pins(const int* per_rank7, int nranks, int* agreed7);
/* Test hook, callable without a device: the runs of the halo message of one face of `hr` rows (option "halo_trim"; csrc/lbm_strips.inc.hpp
 * face_runs) as up to five {first sub-row, sub-rows} pairs relative to the first sub-row of the block (row-interleaved layout: nine
 * sub-rows per lattice row); south_block != 0: the block lies below the strip it borders (a sender's top rows / a receiver's south ghost
 * rows). Returns the number of runs. Replaces the nine-values-per-edge-cell buffers of pack_data_for_sending, LBMGrid.h:395-440. */
int lbm_debug_face_runs(int hr, int trim, int south_block, int* runs10);
/* Test hook, callable without a device: the candidate plans lbm_initialise would time on a whole-domain context of this grid
 * (csrc/lbm_plan.hpp), one per line: "name|lbm_set_option pairs|dominant kernel|iterations per launch". */
int lbm_debug_plan_candidates(int nx, int ny, int precision, int arith, int num_cus, char* out, int cap);
/* Test hook, callable without a device: the host threads that drive the strips of an in-process group (csrc/lbm_ctx.hpp GroupPool) on a
 * dummy job of `rounds` rounds with one rendezvous each, `repeat` times on one pool; in the last run strip `fail_strip` reports an
 * injected error in round `fail_round` and strip `stall_strip` sleeps `stall_ms` before the rendezvous of round `stall_round` (-1:
 * nobody). Returns what the last run returned: LBM_OK, the injected LBM_ERR_HIP (every thread left at the same rendezvous), or
 * LBM_ERR_TIMEOUT naming the strip that was waited for (bound: timeout_ms; 0 = LBM_WAIT_TIMEOUT_MS). *rendezvous_out = rendezvous strip
 * 0 passed in the last run. The reference's counterpart are the implicit barriers of its OpenMP regions (LBMSolver.h:87,131). */
int lbm_debug_group_pool(int n, int rounds, int fail_strip, int fail_round, int stall_strip, int stall_round, int stall_ms, long timeout_ms, int repeat,
                         int* rendezvous_out);
/* Test hook, callable without a device: a DRY RUN of the launch choreography of a strip run and its check (csrc/lbm_choreo.inc.hpp). The
 * functions that issue a launch group (plan_launch, issue_before, the exchanges, issue_after) run on contexts without a device and
 * record every kernel (with the rows it writes and, through its depth, reads), event record, cross-stream wait, copy, send and receive;
 * the record is replayed with vector clocks. Returns the number of violations — RACE: two accesses to the same row of the same buffer, at
 * least one of them a write, that no event orders; STALE: a launch (or the force kernel) reads a row that does not hold the iteration it
 * needs — or < 0; `out` receives their description (and, with dump != 0, every recorded operation). What it replaces: the ordering the
 * reference gets from MPI_Waitall before unpack_received_data (LBMGrid.h:278-283).
 * bounds2 = nstrips x {y_start, rows}; transport 0 in-process group with peer copies, 1 in-process group over RCCL, 2 ONE strip as a rank
 * of a multi-process RCCL run (its faces follow from y_start / rows / ny), 3 ONE strip exchanging with itself (loopback copies);
 * options = "key=value ..." as for lbm_set_option (the plan must be pinned: nothing is measured); calls2 = ncalls x {nsteps, output_frequency}. */
int lbm_debug_choreography(int nx, int ny, const int* bounds2, int nstrips, int precision, int transport, const char* options,
                           const int* calls2, int ncalls, int dump, char* out, int cap);
/* Test hook, callable without a device: exchange_rccl's multi-rank branch — the one piece of the library no run has executed yet (it needs two
 * GPUs) — run DRY on every rank of an `nranks`-process strip run: each rank issues its launch groups as in lbm_debug_choreography and the
 * posting loops hand their sends / receives (peer, offset, count) to a transcript instead of RCCL. RCCL pairs the k-th send to a peer with the
 * peer's k-th receive from the sender: the hook checks, for every pair of neighbours and both directions, that the sequences have the same
 * length and counts and that every message leaves and lands at the same offset inside its block of edge / ghost rows, and that all ranks
 * issue the same number of exchanges. Returns the number of mismatches (0 = the ranks would pair up) or < 0; `out` describes them.
 * options: for every rank; options_rank1 (nullable): additionally for rank 1 only — ranks that disagree (e.g. on "halo_trim") must be
 * flagged. Replaces the tag / count matching of the reference's MPI_Isend / MPI_Irecv pairs (LBMGrid.h:255-276). */
int lbm_debug_p2p_matching(int nx, int ny, const int* bounds2, int nranks, int precision, const char* options, const char* options_rank1,
                           const int* calls2, int ncalls, char* out, int cap);
/* SHA-256 (16 hex digits) of the sources this binary was compiled from (csrc/ and this header); build.py rebuilds
 * when it differs from the tree, bench.py prints it. */
const char* lbm_build_id(void);

#ifdef __cplusplus
}
#endif
#endif
