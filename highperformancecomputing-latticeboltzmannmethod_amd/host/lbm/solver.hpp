// host/lbm/solver.hpp — LBM::Solver: constructor, initialise(), run(IOManager&), get_grid(), get_params() as in the
// reference (LBMSolver.h:23,31,43,80-81) plus step(t, io) = the loop body (LBMSolver.h:49-75), over the HIP backend.
#pragma once
#include "grid.hpp"
#include "io.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <sys/stat.h>

namespace LBM {

class Solver {
public:
    explicit Solver(const SimulationParams& params, bool enable_vtk = false, const BackendOptions& opt = {})
        : params_(params), opt_(opt), grid_(params, opt), enable_vtk_output_(enable_vtk) {
        if (enable_vtk) mkdir("vtk_output", 0755);   // LBMSolver.h:26-28
    }

    void initialise() {   // LBMSolver.h:31-41
        if (!opt_.quiet)
            std::printf("Cylinder Flow LBM Parameters:\n  Domain: %d×%d\n  tau = %g, nu = %g\n  Inlet velocity = %g\n"
                        "  Reynolds number = %g\n", params_.nx, params_.ny, params_.tau, params_.nu(),
                        params_.inlet_velocity, params_.reynolds());
        const int solid = grid_.setup_and_initialise();
        if (!opt_.quiet) {
            std::printf("  Cylinder: center=(%d,%d), radius=%d cells\n  Solid cells: %d\n  Plan: %s\n",
                        params_.get_cylinder_x(), params_.get_cylinder_y(), params_.get_cylinder_radius_cells(), solid,
                        grid_.plan());
            std::fflush(stdout);
        }
    }

    // One loop body of the reference's run() (LBMSolver.h:49-75). Returns false when iteration t is unstable.
    bool step(int t, IOManager& io) {
        if (t != grid_.steps_done()) throw std::runtime_error("Solver::step: iterations must be taken in order");
        if (t % params_.output_frequency == 0) io.record_forces(t, grid_, params_);
        grid_.advance(1, 0);
        if (!report_stability()) return false;
        after_iteration(t, io);
        return true;
    }

    // Solver::run (LBMSolver.h:43-78). Iterations are queued on the GPU in chunks that end at output iterations;
    // forces come from the device-resident log, the stability word is read once per chunk (the reported timestep
    // is the first unstable iteration, as the reference prints at :62).
    bool run(IOManager& io) {
        if (!opt_.quiet) { std::printf("Starting LBM cylinder flow simulation...\n"); std::fflush(stdout); }
        const int T = params_.num_timesteps, of = std::max(1, params_.output_frequency);
        const auto w0 = std::chrono::steady_clock::now();
        int t = grid_.steps_done();
        const int t_begin = t;
        while (t < T) {
            const int snap = ((std::max(t, 1) + of - 1) / of) * of;      // next iteration > 0 with t % of == 0
            const int stop = std::min(T, snap + 1);                       // run through it: the snapshot follows it
            grid_.advance(stop - t, of);
            const int bad = grid_.first_unstable_step();
            for (const auto& r : grid_.drain_force_log())
                if (bad < 0 || r.timestep <= bad) io.append_force_row(r.timestep, r.fx, r.fy, params_);
            if (bad >= 0) {
                std::fprintf(stderr, "Simulation unstable at timestep %d\n", bad);
                return false;
            }
            t = stop;
            after_iteration(t - 1, io);
        }
        io.finish_async();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        if (!opt_.quiet && T > t_begin)
            std::printf("Time loop: %d steps in %.3f s = %.1f MLUPS (incl. output)\n", T - t_begin, sec,
                        1e-6 * params_.nx * params_.ny * double(T - t_begin) / sec);
        return true;
    }

    // Restart support: continue a run from a state written by save_state (same parameters).
    void load_state(const std::string& path) { grid_.load_state(path); }
    void save_state(const std::string& path) const { grid_.save_state(path); }

    const Grid& get_grid() const { return grid_; }
    Grid& get_grid() { return grid_; }   // (the reference's Solver reaches its Grid's mutable accessors as a member)
    const SimulationParams& get_params() const { return params_; }

private:
    bool report_stability() {
        const int bad = grid_.first_unstable_step();
        if (bad < 0) return true;
        std::fprintf(stderr, "Simulation unstable at timestep %d\n", bad);
        return false;
    }
    // Log line + VTK frame of LBMSolver.h:66-75 for the iteration that has just completed.
    void after_iteration(int t, IOManager& io) {
        if (!(t > 0 && t % params_.output_frequency == 0)) return;
        const double max_vel = grid_.max_velocity();
        if (!opt_.quiet) { std::printf("Timestep %d: max_vel=%.6f\n", t, max_vel); std::fflush(stdout); }
        if (enable_vtk_output_ && t >= params_.vtk_start_step) {
            if (opt_.async_vtk) io.write_vtk_async(grid_.ux_field(), grid_.uy_field(), grid_.rho_field(), params_, t);
            else IOManager::write_vtk_timestep(grid_.ux_field(), grid_.uy_field(), grid_.rho_field(), params_, t);
        }
    }

    SimulationParams params_;
    BackendOptions opt_;
    Grid grid_;
    bool enable_vtk_output_;
};

}  // namespace LBM
