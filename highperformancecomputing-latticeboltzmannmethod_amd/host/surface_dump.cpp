// host/surface_dump.cpp — test helper: drives the C++ mirror exactly as a client of the reference's classes would
// (Solver::initialise, the per-iteration Solver::step, then the Grid read accessors rho/ux/uy/f_current/f_next/is_solid,
// check_stability, max_velocity) and dumps what the accessors return, in the layout of oracle/ref_driver's dump, for
// tests/test_host_cpp.py to compare with the reference's own values (tests/golden/g1_128x32_s100.npz).
#include "compat/LBMConfig.h"
#include "compat/LBMIO.h"
#include "compat/LBMSolver.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: surface_dump nx ny steps output_frequency out.bin [strips [poke_at_step]]\n"); return 2; }
    LBM::SimulationParams p;
    p.nx = std::atoi(argv[1]); p.ny = std::atoi(argv[2]); p.num_timesteps = std::atoi(argv[3]);
    p.output_frequency = std::atoi(argv[4]);
    LBM::BackendOptions opt;
    opt.quiet = true;
    if (argc > 6) opt.strips = std::atoi(argv[6]);         // several row strips behind the one Grid (on the GPUs present)
    const int poke_at = argc > 7 ? std::atoi(argv[7]) : -1;
    LBM::Solver solver(p, false, opt);
    LBM::IOManager io;
    solver.initialise();
    for (int t = 0; t < p.num_timesteps; ++t) {
        if (t == poke_at) {                                // a client writing through Grid::f_current (LBMGrid.h:115)
            LBM::Grid& gw = solver.get_grid();
            const int cx = gw.local_nx() / 2 + 1, cy = gw.local_ny() / 2 + 1;   // ghost-inclusive coordinates
            gw.f_current(cx, cy, 1) += 1e-3;
            gw.f_current_ptr(cx + 3, cy - 2)[5] *= 1.01;
        }
        if (!solver.step(t, io)) return 1;                 // the loop body of the reference's run(), one call each
    }
    const LBM::Grid& g = solver.get_grid();
    std::FILE* fp = std::fopen(argv[5], "wb");
    if (!fp) return 3;
    const int nx = g.local_nx(), ny = g.local_ny(), tnx = g.total_nx(), tny = g.total_ny();
    const int hdr[4] = {nx, ny, p.num_timesteps, g.check_stability() ? 1 : 0};
    std::fwrite(hdr, sizeof(int), 4, fp);
    std::vector<double> v;
    auto flush = [&] { std::fwrite(v.data(), sizeof(double), v.size(), fp); v.clear(); };
    for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) v.push_back(g.rho(x, y));
    flush();
    for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) v.push_back(g.ux(x, y));
    flush();
    for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) v.push_back(g.uy(x, y));
    flush();
    for (int gy = 0; gy < tny; ++gy) for (int gx = 0; gx < tnx; ++gx) for (int i = 0; i < LBM::Q; ++i) v.push_back(g.f_current(gx, gy, i));
    flush();
    for (int gy = 0; gy < tny; ++gy) for (int gx = 0; gx < tnx; ++gx) for (int i = 0; i < LBM::Q; ++i) v.push_back(g.f_next(gx, gy, i));
    flush();
    std::vector<unsigned char> solid;
    for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) solid.push_back(g.is_solid(x, y) ? 1 : 0);
    std::fwrite(solid.data(), 1, solid.size(), fp);
    const double mv = g.max_velocity();
    std::fwrite(&mv, sizeof(double), 1, fp);
    std::fclose(fp);
    return 0;
}
