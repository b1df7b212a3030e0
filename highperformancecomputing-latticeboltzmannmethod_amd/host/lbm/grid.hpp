// host/lbm/grid.hpp — LBM::Grid: the read surface of the reference's Grid (LBMGrid.h:105-150,285,319) backed by the
// HIP library. The populations live on the GPU (SoA planes, include/lbm_hip.h); the accessors below serve host
// mirrors that are refreshed lazily from the device whenever the device state has advanced.
#pragma once
#include "../../../include/lbm_hip.h"
#include "params.hpp"

#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace LBM {

class Grid {
public:
    Grid(const SimulationParams& p, const BackendOptions& opt = {}) : nx_(p.nx), ny_(p.ny) {
        lbm_params lp{};
        lp.tau = p.tau; lp.inlet_velocity = p.inlet_velocity; lp.nx = p.nx; lp.ny = p.ny;
        lp.cylinder_x = p.cylinder_x; lp.cylinder_y = p.cylinder_y; lp.cylinder_radius = p.cylinder_radius;
        lp.y_start = 0; lp.local_ny = p.ny;
        lp.precision = opt.fp32 ? LBM_PRECISION_F32 : LBM_PRECISION_F64;
        lp.force_log_capacity = 0;
        check(lbm_create(&lp, opt.device, &ctx_), "lbm_create");
        check(lbm_set_option(ctx_, "tune", opt.tune ? 1 : 0), "lbm_set_option");
        if (!opt.quiet) {   // the banner of Grid::Grid (LBMGrid.h:92-102), restated for this backend
            std::printf("MI355X HIP Grid\n  Global domain: %dx%d\n  GPUs: 1 (row strips)\n  Local with ghosts: %dx%d\n"
                        "  Ghost layers: 1\n  Precision: %s\n  Memory on device: %.2f MB\n",
                        nx_, ny_, nx_ + 2, ny_ + 2, opt.fp32 ? "fp32" : "fp64",
                        2.0 * 9.0 * (nx_ + 2.0) * (ny_ + 2.0) * (opt.fp32 ? 4 : 8) / (1024.0 * 1024.0));
        }
    }
    ~Grid() { lbm_destroy(ctx_); }
    Grid(const Grid&) = delete;
    Grid& operator=(const Grid&) = delete;

    // ---- geometry / sizes (LBMGrid.h:129-150) ----
    int x_start() const { return 0; }
    int y_start() const { return 0; }
    int local_nx() const { return nx_; }
    int local_ny() const { return ny_; }
    int total_nx() const { return nx_ + 2; }
    int total_ny() const { return ny_ + 2; }
    int global_nx() const { return nx_; }
    int global_ny() const { return ny_; }
    int mpi_rank() const { return 0; }
    int mpi_size() const { return 1; }
    bool is_left_boundary() const { return true; }
    bool is_right_boundary() const { return true; }
    bool is_bottom_boundary() const { return true; }
    bool is_top_boundary() const { return true; }
    bool is_solid(int x, int y) const { ensure_solid(); return solid_[idx(x, y)] != 0; }

    // ---- macroscopic fields, interior coordinates (LBMGrid.h:124-129) ----
    double rho(int x, int y) const { ensure_macros(); return rho_[idx(x, y)]; }
    double ux(int x, int y) const { ensure_macros(); return ux_[idx(x, y)]; }
    double uy(int x, int y) const { ensure_macros(); return uy_[idx(x, y)]; }
    const std::vector<double>& rho_field() const { ensure_macros(); return rho_; }
    const std::vector<double>& ux_field() const { ensure_macros(); return ux_; }
    const std::vector<double>& uy_field() const { ensure_macros(); return uy_; }

    // ---- populations, ghost-inclusive coordinates (LBMGrid.h:116-119); debug/parity path ----
    double f_current(int gx, int gy, int i) const { ensure_f(0); return fc_[fidx(gx, gy, i)]; }
    double f_next(int gx, int gy, int i) const { ensure_f(1); return fn_[fidx(gx, gy, i)]; }

    // Grid::check_stability (LBMGrid.h:285-317): evaluated on the device inside every step kernel.
    bool check_stability() const { return first_unstable_step() < 0; }
    int first_unstable_step() const {
        int t = -1;
        check(lbm_first_unstable_step(ctx_, &t), "lbm_first_unstable_step");
        return t;
    }
    // Grid::max_velocity (LBMGrid.h:319-344)
    double max_velocity() const {
        double v = 0.0;
        check(lbm_max_velocity_sq(ctx_, &v), "lbm_max_velocity_sq");
        return std::sqrt(v);
    }

    // ---- device control used by Solver / IOManager ----
    int setup_and_initialise() {   // setup_geometry + initialise (LBMGrid.h:152-246) + collision of iteration 0
        int solid = 0;
        check(lbm_initialise(ctx_, &solid), "lbm_initialise");
        invalidate();
        return solid;
    }
    void advance(int nsteps, int output_frequency) {
        check(lbm_step(ctx_, nsteps, output_frequency), "lbm_step");
        invalidate();
    }
    int steps_done() const { return lbm_steps_done(ctx_); }
    void forces_now(double& fx, double& fy) const { check(lbm_get_forces(ctx_, &fx, &fy), "lbm_get_forces"); }
    std::vector<lbm_force_row> drain_force_log() const {
        std::vector<lbm_force_row> rows(4096);
        const int n = lbm_drain_force_log(ctx_, rows.data(), (int)rows.size());
        check(n, "lbm_drain_force_log");
        rows.resize(n);
        return rows;
    }
    // checkpoint / restart (build-only feature; the reference keeps its state in memory only)
    void save_state(const std::string& path) const { check(lbm_save_state(ctx_, path.c_str()), "lbm_save_state"); }
    void load_state(const std::string& path) {
        check(lbm_load_state(ctx_, path.c_str()), "lbm_load_state");
        invalidate();
    }
    const char* plan() const { return lbm_plan(ctx_); }
    lbm_ctx* handle() const { return ctx_; }

private:
    static void check(int rc, const char* what) {
        if (rc < 0) throw std::runtime_error(std::string(what) + ": " + lbm_last_error());
    }
    size_t idx(int x, int y) const { return static_cast<size_t>(y) * nx_ + x; }
    size_t fidx(int gx, int gy, int i) const { return (static_cast<size_t>(gy) * (nx_ + 2) + gx) * Q + i; }
    void invalidate() { macros_ok_ = false; f_ok_[0] = f_ok_[1] = false; }
    void ensure_macros() const {
        if (macros_ok_) return;
        const size_t n = static_cast<size_t>(nx_) * ny_;
        rho_.resize(n); ux_.resize(n); uy_.resize(n);
        check(lbm_get_macros(ctx_, rho_.data(), ux_.data(), uy_.data()), "lbm_get_macros");
        macros_ok_ = true;
    }
    void ensure_f(int which) const {
        if (f_ok_[which]) return;
        auto& v = which == 0 ? fc_ : fn_;
        v.resize(static_cast<size_t>(nx_ + 2) * (ny_ + 2) * Q);
        check(lbm_get_populations(ctx_, which, v.data()), "lbm_get_populations");
        f_ok_[which] = true;
    }
    void ensure_solid() const {
        if (!solid_.empty()) return;
        solid_.resize(static_cast<size_t>(nx_) * ny_);
        check(lbm_get_solid(ctx_, solid_.data()), "lbm_get_solid");
    }

    int nx_, ny_;
    lbm_ctx* ctx_ = nullptr;
    mutable std::vector<double> rho_, ux_, uy_, fc_, fn_;
    mutable std::vector<unsigned char> solid_;
    mutable bool macros_ok_ = false;
    mutable bool f_ok_[2] = {false, false};
};

}  // namespace LBM
