// host/lbm/grid.hpp — LBM::Grid: the surface of the reference's Grid (LBMGrid.h:105-150,285,319) backed by the HIP
// library. The populations live on the GPU(s) (SoA planes, include/lbm_hip.h); the accessors below serve host mirrors
// that are refreshed lazily from the device whenever the device state has advanced.
//
// Decomposition. The reference's Grid is one MPI rank's block of a 2-D Cartesian topology, and Solver / IOManager gather
// across ranks (LBMGrid.h:347-392, LBMSolver.h:269-362, LBMIO.h:167-168,225-300). Here ONE Grid is the whole lattice and
// owns N row strips, one lbm_ctx each, on one or several GPUs of the node (lbm_group_*): the strips advance in lockstep
// with their halo rows exchanged device to device, and this class does the gathers — rows concatenated by y_start,
// force partial sums added, the stability word reduced by min, max|u| by max.
#pragma once
#include "../../../include/lbm_hip.h"
#include "params.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace LBM {

class Grid {
public:
    Grid(const SimulationParams& p, const BackendOptions& opt = {}) : nx_(p.nx), ny_(p.ny) {
        const int ndev = std::max(1, std::min(opt.gpus, lbm_device_count()));
        int nstrips = opt.strips > 0 ? opt.strips : ndev;
        if (nstrips > ny_) nstrips = ny_;
        // Grid::initialise_2d_topology (LBMGrid.h:347-364) as a 1-D split of the rows: the first ny % n strips get one more
        const int base = ny_ / nstrips, rem = ny_ % nstrips;
        int y = 0;
        for (int k = 0; k < nstrips; ++k) {
            const int n = base + (k < rem ? 1 : 0);
            lbm_params lp{};
            lp.tau = p.tau; lp.inlet_velocity = p.inlet_velocity; lp.nx = p.nx; lp.ny = p.ny;
            lp.cylinder_x = p.cylinder_x; lp.cylinder_y = p.cylinder_y; lp.cylinder_radius = p.cylinder_radius;
            lp.y_start = y; lp.local_ny = n;
            lp.precision = opt.fp32 ? LBM_PRECISION_F32 : LBM_PRECISION_F64;
            lp.force_log_capacity = 0;
            lbm_ctx* c = nullptr;
            const int dev = (opt.device + k % ndev) % std::max(1, lbm_device_count());
            check(lbm_create(&lp, dev, &c), "lbm_create");
            ctx_.push_back(c);
            y0_.push_back(y);
            nyl_.push_back(n);
            devs_.push_back(dev);
            check(lbm_set_option(c, "tune", opt.tune ? 1 : 0), "lbm_set_option");
            check(lbm_set_option(c, "arith", opt.contracted ? 1 : 0), "lbm_set_option");
            y += n;
        }
        check(lbm_group_link(ctx_.data(), (int)ctx_.size(), opt.rccl ? 1 : 0), "lbm_group_link");
        if (!opt.quiet) {   // the banner of Grid::Grid (LBMGrid.h:92-102), restated for this backend
            std::printf("MI355X HIP Grid\n  Global domain: %dx%d\n  GPUs: %d, row strips: %d (halo transport: %s)\n"
                        "  Local with ghosts: %dx%d\n  Ghost layers: 1\n  Precision: %s, collision arithmetic: %s\n"
                        "  Memory on device: %.2f MB\n",
                        nx_, ny_, ndev, nstrips, nstrips == 1 ? "none" : (opt.rccl ? "RCCL send/recv" : "peer copies"),
                        nx_ + 2, ny_ + 2, opt.fp32 ? "fp32" : "fp64", opt.contracted ? "FMA-contracted" : "strict IEEE",
                        2.0 * 9.0 * (nx_ + 2.0) * (ny_ + 2.0) * (opt.fp32 ? 4 : 8) / (1024.0 * 1024.0));
        }
    }
    ~Grid() {
        for (lbm_ctx* c : ctx_) lbm_destroy(c);
    }
    Grid(const Grid&) = delete;
    Grid& operator=(const Grid&) = delete;

    // ---- geometry / sizes (LBMGrid.h:129-150): this Grid is the whole lattice ----
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
    // the strips behind it
    int num_strips() const { return (int)ctx_.size(); }
    int strip_y_start(int k) const { return y0_[(size_t)k]; }
    int strip_rows(int k) const { return nyl_[(size_t)k]; }
    int strip_device(int k) const { return devs_[(size_t)k]; }

    // ---- macroscopic fields, interior coordinates (LBMGrid.h:124-129). The non-const forms hand out the host mirror,
    // as the reference hands out its arrays: values written there are overwritten by the next iteration's collision. ----
    const double& rho(int x, int y) const { ensure_macros(); return rho_[idx(x, y)]; }
    const double& ux(int x, int y) const { ensure_macros(); return ux_[idx(x, y)]; }
    const double& uy(int x, int y) const { ensure_macros(); return uy_[idx(x, y)]; }
    double& rho(int x, int y) { ensure_macros(); return rho_[idx(x, y)]; }
    double& ux(int x, int y) { ensure_macros(); return ux_[idx(x, y)]; }
    double& uy(int x, int y) { ensure_macros(); return uy_[idx(x, y)]; }
    const std::vector<double>& rho_field() const { ensure_macros(); return rho_; }
    const std::vector<double>& ux_field() const { ensure_macros(); return ux_; }
    const std::vector<double>& uy_field() const { ensure_macros(); return uy_; }

    // ---- populations, ghost-inclusive coordinates (LBMGrid.h:113-122). f_current is the state the next iteration's
    // collision reads: values written through the non-const forms are uploaded before the next iteration (interior
    // cells; ghost cells are owned by the halo logic). f_next is dead between iterations in the reference (collision
    // overwrites it), so writes to it stay in the host mirror. ----
    const double& f_current(int gx, int gy, int i) const { ensure_f(0); return fc_[fidx(gx, gy, i)]; }
    const double& f_next(int gx, int gy, int i) const { ensure_f(1); return fn_[fidx(gx, gy, i)]; }
    double& f_current(int gx, int gy, int i) { ensure_f(0); fc_dirty_ = true; return fc_[fidx(gx, gy, i)]; }
    double& f_next(int gx, int gy, int i) { ensure_f(1); return fn_[fidx(gx, gy, i)]; }
    double* f_current_ptr(int gx, int gy) { ensure_f(0); fc_dirty_ = true; return &fc_[fidx(gx, gy, 0)]; }
    double* f_next_ptr(int gx, int gy) { ensure_f(1); return &fn_[fidx(gx, gy, 0)]; }

    // Grid::check_stability (LBMGrid.h:285-317): evaluated on the device inside every step kernel; min over the strips
    // (the reference's MPI_Allreduce(MIN) of the flag, :315).
    bool check_stability() const { return first_unstable_step() < 0; }
    int first_unstable_step() const {
        int first = -1;
        for (lbm_ctx* c : ctx_) {
            int t = -1;
            check(lbm_first_unstable_step(c, &t), "lbm_first_unstable_step");
            if (t >= 0 && (first < 0 || t < first)) first = t;
        }
        return first;
    }
    // Grid::max_velocity (LBMGrid.h:319-344): max over the strips, then the square root
    double max_velocity() const {
        double m = 0.0;
        for (lbm_ctx* c : ctx_) {
            double v = 0.0;
            check(lbm_max_velocity_sq(c, &v), "lbm_max_velocity_sq");
            m = std::max(m, v);
        }
        return std::sqrt(m);
    }

    // ---- device control used by Solver / IOManager ----
    int setup_and_initialise() {   // setup_geometry + initialise (LBMGrid.h:152-246) + collision of iteration 0
        int solid = 0;
        check(lbm_group_initialise(ctx_.data(), (int)ctx_.size(), &solid), "lbm_group_initialise");
        invalidate();
        return solid;
    }
    void advance(int nsteps, int output_frequency) {
        upload_f_current();
        check(lbm_group_step(ctx_.data(), (int)ctx_.size(), nsteps, output_frequency), "lbm_group_step");
        invalidate();
    }
    int steps_done() const { return lbm_steps_done(ctx_[0]); }
    // IOManager::record_forces' MPI_Reduce(SUM) over the ranks (LBMIO.h:167-168): each strip sums the links whose fluid
    // end it owns
    void forces_now(double& fx, double& fy) const {
        fx = fy = 0.0;
        for (lbm_ctx* c : ctx_) {
            double a = 0.0, b = 0.0;
            check(lbm_get_forces(c, &a, &b), "lbm_get_forces");
            fx += a; fy += b;
        }
    }
    std::vector<lbm_force_row> drain_force_log() const {
        std::vector<lbm_force_row> sum;
        for (size_t k = 0; k < ctx_.size(); ++k) {
            std::vector<lbm_force_row> rows(4096);
            const int n = lbm_drain_force_log(ctx_[k], rows.data(), (int)rows.size());
            check(n, "lbm_drain_force_log");
            rows.resize((size_t)n);
            if (k == 0) sum = rows;
            else {
                if (rows.size() != sum.size()) throw std::runtime_error("force logs of the strips differ in length");
                for (size_t r = 0; r < rows.size(); ++r) { sum[r].fx += rows[r].fx; sum[r].fy += rows[r].fy; }
            }
        }
        return sum;
    }
    // checkpoint / restart (build-only feature; the reference keeps its state in memory only): one file per strip
    // (`path` itself for a single strip, `path.k` otherwise)
    void save_state(const std::string& path) const {
        for (size_t k = 0; k < ctx_.size(); ++k) check(lbm_save_state(ctx_[k], strip_file(path, k).c_str()), "lbm_save_state");
    }
    void load_state(const std::string& path) {
        for (size_t k = 0; k < ctx_.size(); ++k) check(lbm_load_state(ctx_[k], strip_file(path, k).c_str()), "lbm_load_state");
        check(lbm_group_refresh_halos(ctx_.data(), (int)ctx_.size()), "lbm_group_refresh_halos");
        invalidate();
    }
    const char* plan() const { return lbm_plan(ctx_[0]); }
    lbm_ctx* handle(int k = 0) const { return ctx_[(size_t)k]; }

private:
    static void check(int rc, const char* what) {
        if (rc < 0) throw std::runtime_error(std::string(what) + ": " + lbm_last_error());
    }
    std::string strip_file(const std::string& path, size_t k) const {
        return ctx_.size() == 1 ? path : path + "." + std::to_string(k);
    }
    size_t idx(int x, int y) const { return static_cast<size_t>(y) * nx_ + x; }
    size_t fidx(int gx, int gy, int i) const { return (static_cast<size_t>(gy) * (nx_ + 2) + gx) * Q + i; }
    void invalidate() { macros_ok_ = false; f_ok_[0] = f_ok_[1] = false; fc_dirty_ = false; }
    void ensure_macros() const {   // the gather of Solver::write_vtk_frame (LBMSolver.h:340-357): strips stacked by y_start
        if (macros_ok_) return;
        const size_t n = static_cast<size_t>(nx_) * ny_;
        rho_.resize(n); ux_.resize(n); uy_.resize(n);
        for (size_t k = 0; k < ctx_.size(); ++k) {
            const size_t off = static_cast<size_t>(y0_[k]) * nx_;
            check(lbm_get_macros(ctx_[k], rho_.data() + off, ux_.data() + off, uy_.data() + off), "lbm_get_macros");
        }
        macros_ok_ = true;
    }
    void ensure_f(int which) const {
        if (f_ok_[which]) return;
        auto& v = which == 0 ? fc_ : fn_;
        const size_t row = static_cast<size_t>(nx_ + 2) * Q;
        v.resize(row * (ny_ + 2));
        std::vector<double> part;
        for (size_t k = 0; k < ctx_.size(); ++k) {
            part.resize(row * (nyl_[k] + 2));
            check(lbm_get_populations(ctx_[k], which, part.data()), "lbm_get_populations");
            // interior rows of the strip; the physical ghost rows come from the first / last strip
            const size_t first = (k == 0) ? 0 : 1, last = (k + 1 == ctx_.size()) ? nyl_[k] + 2 : nyl_[k] + 1;
            std::copy(part.begin() + first * row, part.begin() + last * row, v.begin() + (y0_[k] + first) * row);
        }
        f_ok_[which] = true;
    }
    void upload_f_current() {   // values a client wrote through f_current(x,y,i): re-collided into the device state
        if (!fc_dirty_) return;
        const size_t row = static_cast<size_t>(nx_ + 2) * Q;
        for (size_t k = 0; k < ctx_.size(); ++k)
            check(lbm_set_f_current(ctx_[k], fc_.data() + static_cast<size_t>(y0_[k]) * row), "lbm_set_f_current");
        check(lbm_group_refresh_halos(ctx_.data(), (int)ctx_.size()), "lbm_group_refresh_halos");
        fc_dirty_ = false;
    }
    void ensure_solid() const {
        if (!solid_.empty()) return;
        solid_.resize(static_cast<size_t>(nx_) * ny_);
        for (size_t k = 0; k < ctx_.size(); ++k)
            check(lbm_get_solid(ctx_[k], solid_.data() + static_cast<size_t>(y0_[k]) * nx_), "lbm_get_solid");
    }

    int nx_, ny_;
    std::vector<lbm_ctx*> ctx_;
    std::vector<int> y0_, nyl_, devs_;
    mutable std::vector<double> rho_, ux_, uy_, fc_, fn_;
    mutable std::vector<unsigned char> solid_;
    mutable bool macros_ok_ = false;
    mutable bool f_ok_[2] = {false, false};
    bool fc_dirty_ = false;
};

}  // namespace LBM
