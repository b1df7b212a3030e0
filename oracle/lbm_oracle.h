/* oracle/lbm_oracle.h — TEST INFRASTRUCTURE ONLY (never linked into the product path).
 *
 * CPU restatement, written from scratch, of the reference's D2Q9-BGK timestep
 * (/root/reference/include/LBMSolver.h:48-76 and the functions it calls). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Parity status: PINNED — the reference ships no tests or golden vectors of its own, so this
 * restatement is pinned against outputs of the unmodified reference compiled in the build
 * container (oracle/_ref/ref_driver, 1 rank, OMP_NUM_THREADS=1) and committed as fixtures in
 * tests/golden/ by oracle/make_fixtures.py; tests/test_oracle_golden.py checks them.
 *
 * Layout mirrors the reference (so that the f_current/f_next accessors compare 1:1):
 *   populations AoS, ghost-inclusive:  f[(gy*(nx+2) + gx)*9 + i]        (LBMGrid.h:105-107)
 *   macros interior-only row-major:    rho[y*nx + x]                     (LBMGrid.h:109-111)
 * A handle can describe the whole domain (y_start=0, local_ny=ny) or one row strip of it
 * (the build's multi-GPU decomposition); strips reproduce the 1-rank result exactly.
 */
#ifndef LBM_ORACLE_H
#define LBM_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double tau;             /* LBMConfig.h:37 */
    double inlet_velocity;  /* LBMConfig.h:38 */
    int nx, ny;             /* LBMConfig.h:39-40 (global interior size) */
    double cylinder_x, cylinder_y, cylinder_radius; /* fractions, LBMConfig.h:46-48 */
} lbmo_params;

typedef struct lbmo lbmo;

lbmo*  lbmo_create(const lbmo_params* p, int y_start, int local_ny);
void   lbmo_destroy(lbmo* s);
int    lbmo_solid_count(const lbmo* s);                 /* LBMGrid.h:152-173 */
void   lbmo_initialise(lbmo* s);                        /* LBMGrid.h:185-246 */
void   lbmo_collide(lbmo* s);                           /* LBMSolver.h:84-126 */
void   lbmo_forces(const lbmo* s, double* fx, double* fy); /* LBMIO.h:114-168 (partial sum of this strip) */
void   lbmo_exchange_physical(lbmo* s);                 /* LBMGrid.h:249-283 at physical boundaries (N1/N2) */
void   lbmo_get_edge_row(const lbmo* s, int north, double* buf /* nx*9 */);  /* pack, LBMGrid.h:419-441 */
void   lbmo_set_ghost_row(lbmo* s, int north, const double* buf /* nx*9 */); /* unpack, LBMGrid.h:468-490 */
void   lbmo_stream(lbmo* s);                            /* LBMSolver.h:128-145 */
void   lbmo_boundaries(lbmo* s);                        /* LBMSolver.h:147-265, sequential order */
int    lbmo_check_stability(const lbmo* s);             /* LBMGrid.h:285-317; 1 = stable */
double lbmo_max_velocity_sq(const lbmo* s);             /* LBMGrid.h:319-344 (before sqrt/allreduce) */
int    lbmo_step(lbmo* s);                              /* whole-domain loop body, LBMSolver.h:49-60; 1 = stable */
int    lbmo_run(lbmo* s, int nsteps);                   /* nsteps loop bodies; returns first unstable t or -1 */

double*        lbmo_rho(lbmo* s);
double*        lbmo_ux(lbmo* s);
double*        lbmo_uy(lbmo* s);
double*        lbmo_f_current(lbmo* s);
double*        lbmo_f_next(lbmo* s);
unsigned char* lbmo_solid(lbmo* s);
int            lbmo_threads(void);
void           lbmo_set_threads(int n);

/* derived parameters, LBMConfig.h:54-65 */
double lbmo_nu(const lbmo_params* p);
double lbmo_reynolds(const lbmo_params* p);
int    lbmo_cylinder_x_cells(const lbmo_params* p);
int    lbmo_cylinder_y_cells(const lbmo_params* p);
int    lbmo_cylinder_radius_cells(const lbmo_params* p);

#ifdef __cplusplus
}
#endif
#endif
