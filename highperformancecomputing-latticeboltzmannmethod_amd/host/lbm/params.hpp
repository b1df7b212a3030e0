// host/lbm/params.hpp — lattice constants and run parameters of the solver surface.
// Mirrors /root/reference/include/LBMConfig.h (names, field order, defaults and derived quantities are the public
// surface a caller of the reference touches: LBMConfig.h:9-34 constants, :36-66 SimulationParams).
#pragma once
#include <array>

namespace LBM {

inline constexpr int Q = 9;   // LBMConfig.h:9
inline constexpr int D = 2;   // LBMConfig.h:10

// Direction numbering 0:(0,0) 1:E 2:N 3:W 4:S 5:NE 6:NW 7:SW 8:SE — observable through f_current(x,y,i).
inline constexpr std::array<std::array<int, 2>, Q> VELOCITIES = {{
    {{0, 0}}, {{1, 0}}, {{0, 1}}, {{-1, 0}}, {{0, -1}}, {{1, 1}}, {{-1, 1}}, {{-1, -1}}, {{1, -1}}}};
inline constexpr std::array<double, Q> WEIGHTS = {4.0 / 9.0,  1.0 / 9.0,  1.0 / 9.0,  1.0 / 9.0, 1.0 / 9.0,
                                                 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0};
inline constexpr std::array<int, Q> OPPOSITE = {0, 3, 4, 1, 2, 7, 8, 5, 6};

struct SimulationParams {
    double tau = 0.6;
    double inlet_velocity = 0.01333;
    int nx = 2048;
    int ny = 512;
    int num_timesteps = 120000;
    int output_frequency = 140;
    double cylinder_x = 0.2;        // fraction of nx
    double cylinder_y = 0.5;        // fraction of ny
    double cylinder_radius = 0.05;  // fraction of ny
    int vtk_start_step = 0;

    double nu() const { return (tau - 0.5) / 3.0; }
    double reynolds() const { return inlet_velocity * (2.0 * cylinder_radius * ny) / nu(); }
    int get_cylinder_x() const { return static_cast<int>(cylinder_x * nx); }
    int get_cylinder_y() const { return static_cast<int>(cylinder_y * ny); }
    int get_cylinder_radius_cells() const { return static_cast<int>(cylinder_radius * ny); }
};

// Build-side run options that the reference does not have (it hard-codes everything in main.cpp:11-12).
struct BackendOptions {
    int device = 0;               // first device; strip k runs on device (device + k) % visible devices ...
    int gpus = 1;                 // ... of the `gpus` devices used (clamped to the devices present)
    int strips = 0;               // row strips the lattice is cut into (0: one per GPU); > gpus: several strips share a GPU
    bool rccl = false;            // strip halos through RCCL (one communicator per strip; needs strips == gpus) instead of
                                  // peer copies over xGMI
    bool contracted = false;      // collision arithmetic: FMA-contracted + one reciprocal (lbm_set_option "arith" 1)
    bool fp32 = false;            // single-precision populations (build-only variant)
    bool tune = true;             // measured plan at initialise (lbm_set_option "tune")
    bool async_vtk = true;        // write VTK frames on a writer thread
    bool quiet = false;
};

}  // namespace LBM
