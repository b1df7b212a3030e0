/* oracle/lbm_oracle.c — TEST INFRASTRUCTURE ONLY. See lbm_oracle.h for scope and parity status.
 *
 * Each function restates one reference function (file:line cited at each). Same phase structure
 * as the reference (separate collide / exchange / stream / BC / stability sweeps over an AoS,
 * ghost-inclusive array), same arithmetic order, but strict IEEE arithmetic
 * (the reference is built with -ffast-math; the two agree to ~1e-13 over the fixture windows)
 * and a race-free sequential boundary order bottom -> top -> inlet -> outlet -> cylinder
 * (the reference's `nowait` loops race at the four corner cells when threaded).
 */
#include "lbm_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define Q 9
/* LBMConfig.h:13-34 */
static const int CX[Q] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
static const int CY[Q] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
static const double W[Q] = {4.0/9.0, 1.0/9.0, 1.0/9.0, 1.0/9.0, 1.0/9.0,
                            1.0/36.0, 1.0/36.0, 1.0/36.0, 1.0/36.0};
static const int OPP[Q] = {0, 3, 4, 1, 2, 7, 8, 5, 6};

struct lbmo {
    lbmo_params p;
    int nx, ny_loc, y_start;   /* strip = global rows [y_start, y_start+ny_loc) */
    int tnx, tny;              /* ghost-inclusive */
    int is_bottom, is_top;     /* strip touches the physical wall */
    double *fc, *fn;           /* f_current, f_next */
    double *rho, *ux, *uy;
    unsigned char* solid;      /* interior only */
    int cyl_x, cyl_y, cyl_r;
};

double lbmo_nu(const lbmo_params* p) { return (p->tau - 0.5) / 3.0; }
double lbmo_reynolds(const lbmo_params* p) {
    double D = 2.0 * p->cylinder_radius * p->ny;
    return (p->inlet_velocity * D) / lbmo_nu(p);
}
int lbmo_cylinder_x_cells(const lbmo_params* p) { return (int)(p->cylinder_x * p->nx); }
int lbmo_cylinder_y_cells(const lbmo_params* p) { return (int)(p->cylinder_y * p->ny); }
int lbmo_cylinder_radius_cells(const lbmo_params* p) { return (int)(p->cylinder_radius * p->ny); }
int lbmo_threads(void) { return omp_get_max_threads(); }
void lbmo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

static inline size_t fidx(const lbmo* s, int gx, int gy) { return ((size_t)gy * s->tnx + gx) * Q; }
static inline size_t midx(const lbmo* s, int x, int y) { return (size_t)y * s->nx + x; }

/* LBMGrid.h:152-173 — pure function of GLOBAL coordinates; cells outside the domain do not exist */
static inline int solid_global(const lbmo* s, int gxg, int gyg) {
    if (gxg < 0 || gxg >= s->nx || gyg < 0 || gyg >= s->p.ny) return 0;
    double dx = gxg - s->cyl_x, dy = gyg - s->cyl_y;
    return (dx * dx + dy * dy) <= (double)(s->cyl_r * s->cyl_r);
}

lbmo* lbmo_create(const lbmo_params* p, int y_start, int local_ny) {
    lbmo* s = (lbmo*)calloc(1, sizeof(lbmo));
    s->p = *p;
    s->nx = p->nx; s->ny_loc = local_ny; s->y_start = y_start;
    s->tnx = s->nx + 2; s->tny = local_ny + 2;
    s->is_bottom = (y_start == 0);
    s->is_top = (y_start + local_ny == p->ny);
    size_t nf = (size_t)s->tnx * s->tny * Q, nm = (size_t)s->nx * local_ny;
    s->fc = (double*)calloc(nf, sizeof(double));
    s->fn = (double*)calloc(nf, sizeof(double));
    s->rho = (double*)malloc(nm * sizeof(double));
    s->ux = (double*)calloc(nm, sizeof(double));
    s->uy = (double*)calloc(nm, sizeof(double));
    s->solid = (unsigned char*)calloc(nm, 1);
    for (size_t k = 0; k < nm; ++k) s->rho[k] = 1.0;  /* LBMGrid.h:73 */
    s->cyl_x = lbmo_cylinder_x_cells(p);
    s->cyl_y = lbmo_cylinder_y_cells(p);
    s->cyl_r = lbmo_cylinder_radius_cells(p);
    for (int y = 0; y < local_ny; ++y)
        for (int x = 0; x < s->nx; ++x)
            s->solid[midx(s, x, y)] = (unsigned char)solid_global(s, x, y_start + y);
    return s;
}

void lbmo_destroy(lbmo* s) {
    if (!s) return;
    free(s->fc); free(s->fn); free(s->rho); free(s->ux); free(s->uy); free(s->solid); free(s);
}

int lbmo_solid_count(const lbmo* s) {
    int n = 0;
    for (size_t k = 0; k < (size_t)s->nx * s->ny_loc; ++k) n += s->solid[k];
    return n;
}

/* LBMUtils.h:9-12 (dir 0) and :22-65 (dirs 1-8): bracket order ((1+3cu)-1.5u^2)+4.5cu^2, then (w*rho)*bracket */
static void feq_init(double rho, double ux, double uy, double* f) {
    const double usq = ux * ux + uy * uy;
    f[0] = W[0] * rho * (1.0 - 1.5 * usq);
    const double t3 = 1.5 * usq;
    for (int i = 1; i < Q; ++i) {
        const double cu = (double)CX[i] * ux + (double)CY[i] * uy;
        const double br = ((1.0 + 3.0 * cu) - t3) + 4.5 * (cu * cu);
        f[i] = (W[i] * rho) * br;
    }
}

/* LBMGrid.h:185-246 */
void lbmo_initialise(lbmo* s) {
    double fe[Q], fs[Q];
    feq_init(1.0, s->p.inlet_velocity, 0.0, fe);
    feq_init(1.0, 0.0, 0.0, fs);
    for (int gy = 0; gy < s->tny; ++gy)
        for (int gx = 0; gx < s->tnx; ++gx) {
            double* a = s->fc + fidx(s, gx, gy);
            double* b = s->fn + fidx(s, gx, gy);
            for (int i = 0; i < Q; ++i) { a[i] = fe[i]; b[i] = fe[i]; }
        }
    for (int y = 0; y < s->ny_loc; ++y)
        for (int x = 0; x < s->nx; ++x) {
            size_t m = midx(s, x, y);
            s->rho[m] = 1.0; s->uy[m] = 0.0;
            if (!s->solid[m]) { s->ux[m] = s->p.inlet_velocity; }
            else {
                s->ux[m] = 0.0;
                double* a = s->fc + fidx(s, x + 1, y + 1);
                double* b = s->fn + fidx(s, x + 1, y + 1);
                for (int i = 0; i < Q; ++i) { a[i] = fs[i]; b[i] = fs[i]; }
            }
        }
    /* Strip mode only: a ghost row that stands for an interior row of the global domain will be
     * overwritten by lbmo_set_ghost_row before it is ever read; nothing to do here. */
}

/* LBMSolver.h:84-126 */
void lbmo_collide(lbmo* s) {
    const double tau_inv = 1.0 / s->p.tau;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < s->ny_loc; ++y)
        for (int x = 0; x < s->nx; ++x) {
            const size_t m = midx(s, x, y);
            if (s->solid[m]) continue;
            const double* f = s->fc + fidx(s, x + 1, y + 1);
            double* fo = s->fn + fidx(s, x + 1, y + 1);
            double r = 0.0, vx = 0.0, vy = 0.0;
            for (int i = 0; i < Q; ++i) {
                r += f[i];
                vx += (double)CX[i] * f[i];
                vy += (double)CY[i] * f[i];
            }
            vx /= r; vy /= r;
            s->rho[m] = r; s->ux[m] = vx; s->uy[m] = vy;
            const double usq = vx * vx + vy * vy;
            for (int i = 0; i < Q; ++i) {
                const double cu = (double)CX[i] * vx + (double)CY[i] * vy;
                const double feq = W[i] * r * (1.0 + 3.0 * cu + 4.5 * cu * cu - 1.5 * usq);
                fo[i] = f[i] - tau_inv * (f[i] - feq);
            }
        }
}

/* LBMIO.h:114-168. The reference walks solid cells and only counts links whose fluid end is inside the
 * caller's own interior; for one rank that is every solid->fluid link of the domain. For a strip we walk
 * the solid cells of the global mask in rows [y_start-1, y_start+ny_loc] and count the links whose FLUID
 * end this strip owns, so the strip partials sum to the 1-rank value. */
void lbmo_forces(const lbmo* s, double* fx_out, double* fy_out) {
    double fx = 0.0, fy = 0.0;
    for (int yg = s->y_start - 1; yg <= s->y_start + s->ny_loc; ++yg)
        for (int x = 0; x < s->nx; ++x) {
            if (!solid_global(s, x, yg)) continue;
            for (int i = 1; i < Q; ++i) {
                const int fxx = x - CX[i], fyg = yg - CY[i];
                const int fyl = fyg - s->y_start;
                if (fxx >= 0 && fxx < s->nx && fyl >= 0 && fyl < s->ny_loc && !s->solid[midx(s, fxx, fyl)]) {
                    const double fi = s->fn[fidx(s, fxx + 1, fyl + 1) + i];
                    fx += 2.0 * (double)CX[i] * fi;
                    fy += 2.0 * (double)CY[i] * fi;
                }
            }
        }
    *fx_out = fx; *fy_out = fy;
}

/* LBMGrid.h:249-283 with every E/W peer = MPI_PROC_NULL: the zero-initialised receive buffers are
 * unpacked into the ghost columns of rows gy=1..ny_loc. Physical N/S ghost rows are not
 * touched (N2). */
void lbmo_exchange_physical(lbmo* s) {
    for (int gy = 1; gy <= s->ny_loc; ++gy) {
        double* w = s->fn + fidx(s, 0, gy);
        double* e = s->fn + fidx(s, s->tnx - 1, gy);
        for (int i = 0; i < Q; ++i) { w[i] = 0.0; e[i] = 0.0; }
    }
}

/* LBMGrid.h:419-441 (pack of the north / south edge row of f_next) */
void lbmo_get_edge_row(const lbmo* s, int north, double* buf) {
    const int gy = north ? s->ny_loc : 1;
    memcpy(buf, s->fn + fidx(s, 1, gy), (size_t)s->nx * Q * sizeof(double));
}

/* LBMGrid.h:468-490 (unpack into the north / south ghost row). In the global picture the two end cells
 * of that row are E/W ghost-column cells of an interior row, i.e. zero (N1): set them so that a strip
 * reproduces the 1-rank f_current exactly, including the dead pulls at x=0 / x=nx-1. */
void lbmo_set_ghost_row(lbmo* s, int north, const double* buf) {
    const int gy = north ? s->tny - 1 : 0;
    memcpy(s->fn + fidx(s, 1, gy), buf, (size_t)s->nx * Q * sizeof(double));
    double* w = s->fn + fidx(s, 0, gy);
    double* e = s->fn + fidx(s, s->tnx - 1, gy);
    for (int i = 0; i < Q; ++i) { w[i] = 0.0; e[i] = 0.0; }
}

/* LBMSolver.h:128-145 */
void lbmo_stream(lbmo* s) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < s->ny_loc; ++y)
        for (int x = 0; x < s->nx; ++x) {
            const int gx = x + 1, gy = y + 1;
            double* d = s->fc + fidx(s, gx, gy);
            for (int i = 0; i < Q; ++i)
                d[i] = s->fn[fidx(s, gx - CX[i], gy - CY[i]) + i];
        }
}

/* LBMSolver.h:147-265 in sequential loop order (N3) */
void lbmo_boundaries(lbmo* s) {
    const int nx = s->nx, nyl = s->ny_loc;
    if (s->is_bottom) {                                   /* :152-164 */
#pragma omp parallel for schedule(static)
        for (int x = 0; x < nx; ++x) {
            if (s->solid[midx(s, x, 0)]) continue;
            double* f = s->fc + fidx(s, x + 1, 1);
            f[2] = f[4]; f[5] = f[7]; f[6] = f[8];
        }
    }
    if (s->is_top) {                                      /* :166-177 */
#pragma omp parallel for schedule(static)
        for (int x = 0; x < nx; ++x) {
            if (s->solid[midx(s, x, nyl - 1)]) continue;
            double* f = s->fc + fidx(s, x + 1, nyl);
            f[4] = f[2]; f[7] = f[5]; f[8] = f[6];
        }
    }
    {                                                     /* inlet :180-207 */
        const double u_in = s->p.inlet_velocity;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < nyl; ++y) {
            if (s->solid[midx(s, 0, y)]) continue;
            double* f = s->fc + fidx(s, 1, y + 1);
            const double rho_bc = (f[0] + f[2] + f[4] + 2.0 * (f[3] + f[6] + f[7])) / (1.0 - u_in);
            f[1] = f[3] + (2.0 / 3.0) * rho_bc * u_in;
            f[5] = f[7] - 0.5 * (f[2] - f[4]) + (1.0 / 6.0) * rho_bc * u_in;
            f[8] = f[6] + 0.5 * (f[2] - f[4]) + (1.0 / 6.0) * rho_bc * u_in;
            const size_t m = midx(s, 0, y);
            s->rho[m] = rho_bc; s->ux[m] = u_in; s->uy[m] = 0.0;
        }
    }
    {                                                     /* outlet :210-236 */
#pragma omp parallel for schedule(static)
        for (int y = 0; y < nyl; ++y) {
            if (s->solid[midx(s, nx - 1, y)]) continue;
            double* f = s->fc + fidx(s, nx, y + 1);
            const double rho_out = 1.0;
            const double u_out = -1.0 + (f[0] + f[2] + f[4] + 2.0 * (f[1] + f[5] + f[8])) / rho_out;
            f[3] = f[1] - (2.0 / 3.0) * rho_out * u_out;
            f[6] = f[8] - 0.5 * (f[2] - f[4]) - (1.0 / 6.0) * rho_out * u_out;
            f[7] = f[5] + 0.5 * (f[2] - f[4]) - (1.0 / 6.0) * rho_out * u_out;
            const size_t m = midx(s, nx - 1, y);
            s->rho[m] = rho_out; s->ux[m] = u_out; s->uy[m] = 0.0;
        }
    }
    /* cylinder :240-263 */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < nyl; ++y)
        for (int x = 0; x < nx; ++x) {
            const size_t m = midx(s, x, y);
            if (!s->solid[m]) continue;
            double* f = s->fc + fidx(s, x + 1, y + 1);
            double t[Q];
            for (int i = 0; i < Q; ++i) t[i] = f[i];
            for (int i = 0; i < Q; ++i) f[i] = t[OPP[i]];
            s->ux[m] = 0.0; s->uy[m] = 0.0;
        }
}

/* LBMGrid.h:285-317 + LBMUtils.h:129-131. The vector part flags NaN, >1e5, <-1e5; the scalar tail (the last
 * total%4 values) flags !isfinite or |v|>=1e5. Ghost cells and solid cells are scanned too (N8). */
int lbmo_check_stability(const lbmo* s) {
    const size_t total = (size_t)s->tnx * s->tny * Q;
    const size_t simd_end = (total / 4) * 4;
    int stable = 1;
#pragma omp parallel for schedule(static) reduction(&& : stable)
    for (size_t k = 0; k < simd_end; ++k) {
        const double v = s->fc[k];
        if (!(v == v) || v > 1e5 || v < -1e5) stable = 0;
    }
    for (size_t k = simd_end; k < total; ++k) {
        const double v = s->fc[k];
        if (!(isfinite(v) && fabs(v) < 1e5)) stable = 0;
    }
    return stable;
}

/* LBMGrid.h:319-344 (value before MPI_Allreduce(MAX) and sqrt) */
double lbmo_max_velocity_sq(const lbmo* s) {
    double mx = 0.0;
    const size_t n = (size_t)s->nx * s->ny_loc;
#pragma omp parallel for schedule(static) reduction(max : mx)
    for (size_t k = 0; k < n; ++k) {
        const double v = s->ux[k] * s->ux[k] + s->uy[k] * s->uy[k];
        if (v > mx) mx = v;
    }
    return mx;
}

/* Loop body of Solver::run for ONE rank owning the whole domain, LBMSolver.h:49-60 (forces are called
 * separately by the caller at the cadence of :52-54, between collide and exchange: use the phase calls). */
int lbmo_step(lbmo* s) {
    lbmo_collide(s);
    lbmo_exchange_physical(s);
    lbmo_stream(s);
    lbmo_boundaries(s);
    return lbmo_check_stability(s);
}

int lbmo_run(lbmo* s, int nsteps) {
    for (int t = 0; t < nsteps; ++t)
        if (!lbmo_step(s)) return t;
    return -1;
}

double* lbmo_rho(lbmo* s) { return s->rho; }
double* lbmo_ux(lbmo* s) { return s->ux; }
double* lbmo_uy(lbmo* s) { return s->uy; }
double* lbmo_f_current(lbmo* s) { return s->fc; }
double* lbmo_f_next(lbmo* s) { return s->fn; }
unsigned char* lbmo_solid(lbmo* s) { return s->solid; }
