// oracle/ref_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// A thin driver around the *unmodified* reference headers (compiled from where they lie under
// /root/reference/include by oracle/Makefile; nothing from the reference is copied into this repo).
// It only calls the reference's public API:
//   LBM::SimulationParams (LBMConfig.h:36), LBM::Solver{initialise,run,get_grid} (LBMSolver.h:23-81),
//   LBM::IOManager (LBMIO.h:35) and the LBM::Grid accessors (LBMGrid.h:105-150).
// Purpose: (1) generate the golden fixtures in tests/golden (oracle/make_fixtures.py),
//          (2) optional "reference" CPU baseline timing for bench.py (--time).
// The binary lands in oracle/_ref/ (git-ignored, travels to the GPU box as a built artefact).
#include "LBMConfig.h"
#include "LBMSolver.h"
#include "LBMIO.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mpi.h>

static void write_vec(FILE* fp, const std::vector<double>& v) { fwrite(v.data(), sizeof(double), v.size(), fp); }

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    LBM::SimulationParams p;
    p.num_timesteps = 10;
    std::string dump;
    bool timing = false, vtk = false, final_results = false;
    for (int a = 1; a < argc; ++a) {
        std::string k = argv[a];
        auto val = [&]() -> const char* { return (a + 1 < argc) ? argv[++a] : "0"; };
        if (k == "--nx") p.nx = atoi(val());
        else if (k == "--ny") p.ny = atoi(val());
        else if (k == "--steps") p.num_timesteps = atoi(val());
        else if (k == "--tau") p.tau = atof(val());
        else if (k == "--u") p.inlet_velocity = atof(val());
        else if (k == "--of") p.output_frequency = atoi(val());
        else if (k == "--cylx") p.cylinder_x = atof(val());
        else if (k == "--cyly") p.cylinder_y = atof(val());
        else if (k == "--cylr") p.cylinder_radius = atof(val());
        else if (k == "--dump") dump = val();
        else if (k == "--time") timing = true;
        else if (k == "--vtk") vtk = true;                 // Solver(params, enable_vtk = true), LBMSolver.h:23
        else if (k == "--final") final_results = true;     // IOManager::write_final_results, LBMIO.h:194
        else { fprintf(stderr, "unknown arg %s\n", k.c_str()); return 2; }
    }
    int rc = 0;
    {
        LBM::Solver solver(p, vtk);
        LBM::IOManager io;
        solver.initialise();
        auto t0 = std::chrono::steady_clock::now();
        bool ok = solver.run(io);
        auto t1 = std::chrono::steady_clock::now();
        double sec = std::chrono::duration<double>(t1 - t0).count();
        const LBM::Grid& g = solver.get_grid();
        if (timing && g.mpi_rank() == 0) {
            printf("REFTIME nx=%d ny=%d steps=%d ranks=%d threads=%d seconds=%.6f mlups=%.3f ok=%d\n",
                   p.nx, p.ny, p.num_timesteps, g.mpi_size(), omp_get_max_threads(), sec,
                   (double)p.nx * p.ny * p.num_timesteps / sec / 1e6, (int)ok);
        }
        if (!ok) rc = 1;
        if (ok && final_results) io.write_final_results(solver.get_grid(), solver.get_params());
        if (!dump.empty() && g.mpi_size() == 1) {
            FILE* fp = fopen(dump.c_str(), "wb");
            if (!fp) { perror("dump"); MPI_Finalize(); return 3; }
            int hdr[4] = {p.nx, p.ny, p.num_timesteps, ok ? 1 : 0};
            fwrite(hdr, sizeof(int), 4, fp);
            const int nx = g.local_nx(), ny = g.local_ny(), tnx = g.total_nx(), tny = g.total_ny();
            std::vector<double> rho((size_t)nx * ny), ux(rho.size()), uy(rho.size());
            std::vector<unsigned char> solid(rho.size());
            for (int y = 0; y < ny; ++y)
                for (int x = 0; x < nx; ++x) {
                    size_t i = (size_t)y * nx + x;
                    rho[i] = g.rho(x, y); ux[i] = g.ux(x, y); uy[i] = g.uy(x, y);
                    solid[i] = g.is_solid(x, y) ? 1 : 0;
                }
            write_vec(fp, rho); write_vec(fp, ux); write_vec(fp, uy);
            std::vector<double> fc((size_t)tnx * tny * LBM::Q), fn(fc.size());
            for (int gy = 0; gy < tny; ++gy)
                for (int gx = 0; gx < tnx; ++gx)
                    for (int i = 0; i < LBM::Q; ++i) {
                        size_t k = ((size_t)gy * tnx + gx) * LBM::Q + i;
                        fc[k] = g.f_current(gx, gy, i); fn[k] = g.f_next(gx, gy, i);
                    }
            write_vec(fp, fc); write_vec(fp, fn);
            fwrite(solid.data(), 1, solid.size(), fp);
            double mv = g.max_velocity();
            fwrite(&mv, sizeof(double), 1, fp);
            fclose(fp);
        }
    }
    MPI_Finalize();
    return rc;
}
