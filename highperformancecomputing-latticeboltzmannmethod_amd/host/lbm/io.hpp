// host/lbm/io.hpp — LBM::IOManager: forces.csv (Cd/Cl), legacy-ASCII VTK frames, final CSVs. File formats follow
// /root/reference/include/LBMIO.h field for field (forces.csv :38-41,:171-185; VTK :55-111; velocity_field.csv
// :302-325; simulation_params.csv :327-365; force statistics :367-413) so that scripts/lift.py and
// scripts/visualise_results.py of the reference run unchanged on these files. Numbers are formatted with
// "%.8f" (== std::fixed << std::setprecision(8)) through one buffered writer; VTK frames can be written by a
// background thread so that a ~300 MB frame at 4096x1024 does not stall the GPU.
#pragma once
#include "grid.hpp"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace LBM {

namespace detail {
// Append-only text sink with a large buffer; fixed-point formatting of doubles at 8 decimals.
class TextFile {
public:
    explicit TextFile(const std::string& path) : fp_(std::fopen(path.c_str(), "w")) {
        if (fp_) std::setvbuf(fp_, nullptr, _IOFBF, 1 << 22);
    }
    ~TextFile() { close(); }
    TextFile(const TextFile&) = delete;
    TextFile& operator=(const TextFile&) = delete;
    bool ok() const { return fp_ != nullptr; }
    void put(const char* s) { std::fputs(s, fp_); }
    void put_int(long v) { std::fprintf(fp_, "%ld", v); }
    void put_f8(double v) { std::fprintf(fp_, "%.8f", v); }
    void flush() { if (fp_) std::fflush(fp_); }
    void close() { if (fp_) { std::fclose(fp_); fp_ = nullptr; } }
private:
    std::FILE* fp_;
};
}  // namespace detail

struct ForceSample { int timestep; double fx, fy, cd, cl; };

class IOManager {
public:
    // IOManager::IOManager (LBMIO.h:35-46): creates forces.csv in the current directory and writes the header.
    IOManager() : forces_(std::make_unique<detail::TextFile>("forces.csv")) {
        if (forces_->ok()) forces_->put("timestep,drag_force,lift_force,drag_coeff,lift_coeff\n");
        else std::fprintf(stderr, "ERROR: Could not open forces.csv\n");
    }
    ~IOManager() { finish_async(); }

    // IOManager::record_forces (LBMIO.h:114-192) for the iteration the grid is about to stream (t == steps_done).
    void record_forces(int timestep, const Grid& grid, const SimulationParams& params) {
        double fx = 0.0, fy = 0.0;
        grid.forces_now(fx, fy);
        append_force_row(timestep, fx, fy, params);
    }
    // Same row from a device-log entry (Solver::run drains the log at output cadence).
    void append_force_row(int timestep, double fx, double fy, const SimulationParams& params) {
        const double d_ref = 2.0 * params.get_cylinder_radius_cells();
        const double q_ref = 0.5 * 1.0 * params.inlet_velocity * params.inlet_velocity * d_ref;
        const double cd = (q_ref > 1e-12) ? fx / q_ref : 0.0;
        const double cl = (q_ref > 1e-12) ? fy / q_ref : 0.0;
        samples_.push_back({timestep, fx, fy, cd, cl});
        if (!forces_->ok()) return;
        forces_->put_int(timestep);
        for (double v : {fx, fy, cd, cl}) { forces_->put(","); forces_->put_f8(v); }
        forces_->put("\n");
        if (timestep % 10000 == 0) forces_->flush();
    }

    // IOManager::write_vtk_timestep (LBMIO.h:55-111): legacy ASCII STRUCTURED_POINTS, vtk_output/lbm_%06d.vtk.
    static void write_vtk_timestep(const std::vector<double>& ux_g, const std::vector<double>& uy_g,
                                   const std::vector<double>& rho_g, const SimulationParams& p, int timestep) {
        char name[256];
        std::snprintf(name, sizeof(name), "vtk_output/lbm_%06d.vtk", timestep);
        detail::TextFile f(name);
        if (!f.ok()) { std::fprintf(stderr, "ERROR: Cannot write %s\n", name); return; }
        const size_t n = static_cast<size_t>(p.nx) * p.ny;
        f.put("# vtk DataFile Version 3.0\nLBM Flow Timestep "); f.put_int(timestep);
        f.put("\nASCII\nDATASET STRUCTURED_POINTS\nDIMENSIONS "); f.put_int(p.nx); f.put(" "); f.put_int(p.ny);
        f.put(" 1\nORIGIN 0 0 0\nSPACING 1 1 1\nPOINT_DATA "); f.put_int(p.nx * p.ny);
        f.put("\nVECTORS velocity double\n");
        for (size_t k = 0; k < n; ++k) { f.put_f8(ux_g[k]); f.put(" "); f.put_f8(uy_g[k]); f.put(" 0.0\n"); }
        f.put("\nSCALARS velocity_magnitude double\nLOOKUP_TABLE default\n");
        for (size_t k = 0; k < n; ++k) { f.put_f8(std::sqrt(ux_g[k] * ux_g[k] + uy_g[k] * uy_g[k])); f.put("\n"); }
        f.put("\nSCALARS density double\nLOOKUP_TABLE default\n");
        for (size_t k = 0; k < n; ++k) { f.put_f8(rho_g[k]); f.put("\n"); }
    }
    // Asynchronous frame: the three fields are moved to a writer thread, at most 2 frames queued.
    void write_vtk_async(std::vector<double> ux, std::vector<double> uy, std::vector<double> rho,
                         const SimulationParams& p, int timestep) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return queue_.size() < 2; });
        queue_.push_back({std::move(ux), std::move(uy), std::move(rho), p, timestep});
        if (!writer_.joinable()) writer_ = std::thread([this] { writer_loop(); });
        cv_.notify_all();
    }
    void finish_async() {
        {
            std::unique_lock<std::mutex> lk(mu_);
            stop_ = true;
            cv_.notify_all();
        }
        if (writer_.joinable()) writer_.join();
        stop_ = false;
    }

    // IOManager::write_final_results (LBMIO.h:194-219): velocity_field.csv, simulation_params.csv, statistics.
    void write_final_results(const Grid& grid, const SimulationParams& params) {
        finish_async();
        std::printf("\nGathering final results...\n");
        const auto& ux = grid.ux_field();
        const auto& uy = grid.uy_field();
        const auto& rho = grid.rho_field();
        write_velocity_field(ux, uy, rho, params);
        write_simulation_params(ux, uy, params);
        forces_->flush();
        print_force_statistics();
        std::printf("Files written: velocity_field.csv, simulation_params.csv, forces.csv\n");
        std::fflush(stdout);
    }
    const std::vector<ForceSample>& samples() const { return samples_; }

private:
    struct Frame { std::vector<double> ux, uy, rho; SimulationParams p; int t; };

    void writer_loop() {
        for (;;) {
            Frame fr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
                if (queue_.empty()) return;
                fr = std::move(queue_.front());
            }
            write_vtk_timestep(fr.ux, fr.uy, fr.rho, fr.p, fr.t);
            {
                std::unique_lock<std::mutex> lk(mu_);
                queue_.pop_front();
                cv_.notify_all();
            }
        }
    }
    static void write_velocity_field(const std::vector<double>& ux, const std::vector<double>& uy,
                                     const std::vector<double>& rho, const SimulationParams& p) {   // LBMIO.h:302-325
        detail::TextFile f("velocity_field.csv");
        if (!f.ok()) { std::fprintf(stderr, "ERROR: Cannot write velocity_field.csv\n"); return; }
        f.put("x,y,ux,uy,rho,velocity_magnitude\n");
        for (int y = 0; y < p.ny; ++y)
            for (int x = 0; x < p.nx; ++x) {
                const size_t k = static_cast<size_t>(y) * p.nx + x;
                f.put_int(x); f.put(","); f.put_int(y);
                for (double v : {ux[k], uy[k], rho[k], std::sqrt(ux[k] * ux[k] + uy[k] * uy[k])}) { f.put(","); f.put_f8(v); }
                f.put("\n");
            }
        std::printf("  velocity_field.csv written\n");
    }
    static void write_simulation_params(const std::vector<double>& ux, const std::vector<double>& uy,
                                        const SimulationParams& p) {   // LBMIO.h:327-365
        detail::TextFile f("simulation_params.csv");
        if (!f.ok()) { std::fprintf(stderr, "ERROR: Cannot write simulation_params.csv\n"); return; }
        double vmax = 0.0, vsum = 0.0;
        for (size_t k = 0; k < ux.size(); ++k) {
            const double v = std::sqrt(ux[k] * ux[k] + uy[k] * uy[k]);
            vmax = std::max(vmax, v);
            vsum += v;
        }
        const double vavg = vsum / (p.nx * p.ny);
        auto row_i = [&](const char* k, long v) { f.put(k); f.put(","); f.put_int(v); f.put("\n"); };
        auto row_d = [&](const char* k, double v) { f.put(k); f.put(","); f.put_f8(v); f.put("\n"); };
        f.put("parameter,value\n");
        row_i("nx", p.nx); row_i("ny", p.ny); row_d("tau", p.tau); row_d("nu", p.nu());
        row_d("inlet_velocity", p.inlet_velocity); row_i("num_timesteps", p.num_timesteps);
        row_d("reynolds_number", p.reynolds()); row_i("cylinder_x", p.get_cylinder_x());
        row_i("cylinder_y", p.get_cylinder_y()); row_i("cylinder_radius", p.get_cylinder_radius_cells());
        row_d("max_velocity", vmax); row_d("avg_velocity", vavg);
        std::printf("  simulation_params.csv written\n");
    }
    // calculate_time_averaged_drag (LBMIO.h:367-413): statistics of the CSV values (8 decimals) for timestep > 1000.
    // (The reference re-reads forces.csv while its own stream is still open and therefore only sees the rows
    // flushed so far; this mirror uses every row it wrote.)
    void print_force_statistics() const {
        auto csv = [](double v) { char b[64]; std::snprintf(b, sizeof(b), "%.8f", v); return std::strtod(b, nullptr); };
        double scd = 0, scl = 0, cdmin = 1e9, cdmax = -1e9, clmin = 1e9, clmax = -1e9;
        int n = 0;
        for (const auto& s : samples_) {
            if (s.timestep <= 1000) continue;
            const double cd = csv(s.cd), cl = csv(s.cl);
            scd += cd; scl += cl; ++n;
            cdmin = std::min(cdmin, cd); cdmax = std::max(cdmax, cd);
            clmin = std::min(clmin, cl); clmax = std::max(clmax, cl);
        }
        if (n == 0) return;
        std::printf("\n=== Time-Averaged Force Coefficients ===\n  Mean C_D = %.6f\n  C_D range: [%.6f, %.6f]\n"
                    "  Mean C_L = %.6f\n  C_L range: [%.6f, %.6f]\n  (Averaged over %d samples)\n",
                    scd / n, cdmin, cdmax, scl / n, clmin, clmax, n);
    }

    std::unique_ptr<detail::TextFile> forces_;
    std::vector<ForceSample> samples_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Frame> queue_;
    std::thread writer_;
    bool stop_ = false;
};

}  // namespace LBM
