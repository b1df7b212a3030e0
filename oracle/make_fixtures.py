#!/usr/bin/env python3
"""oracle/make_fixtures.py — TEST INFRASTRUCTURE ONLY.

Generates tests/golden/*.npz by running the UNMODIFIED reference (oracle/_ref/ref_driver, built by
oracle/Makefile from /root/reference/include) with 1 MPI rank and OMP_NUM_THREADS=1 — the canonical,
race-free configuration. Runs only in the build container (needs /root/reference);
the fixtures it writes are data (inputs + expected outputs), committed, and travel to the GPU box.

Fixture sets:
  g1_128x32_s{1,2,10,100}   default params; rho/ux/uy always, ghost-inclusive f_current/f_next at s2 and s100
  g2_256x64_s{1,100,1000}   default params; rho/ux/uy; forces.csv rows (output_frequency=50) of the 1000-step run
  g3_poiseuille_256x64      cylinder disabled, 20000 steps: ux columns x=128, x=192 (+ rho column)
  g4_1024x256_re100_s3000   tau=0.6 u=0.13020833: every-4th-point rho/ux/uy, six full rows, forces every 100
  g6_inlet_cyl_64x32_s50    cylinder centred ON the inlet column (solid cells in x=0): all arrays
  g7_wall_cyl_64x32_s50     cylinder touching the bottom wall: all arrays
  g9_files_64x32_s1201      the reference's output FILES byte for byte: forces.csv, last VTK frame, velocity_field.csv,
                            simulation_params.csv, stdout (IOManager / write_vtk_frame, LBMIO.h, LBMSolver.h:269-362)
  g8{a,b}_unstable_128x32   tau=0.502 u=0.35 / tau=0.51 u=0.15: the timestep at which the reference reports instability
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref", "ref_driver")
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
FLAG = {"tau": "--tau", "inlet_velocity": "--u", "cylinder_x": "--cylx", "cylinder_y": "--cyly",
        "cylinder_radius": "--cylr"}


def run_ref(nx, ny, steps, of=140, files=False, **kw):
    d = tempfile.mkdtemp(prefix="lbmref_")
    args = [REF, "--nx", str(nx), "--ny", str(ny), "--steps", str(steps), "--of", str(of), "--dump", d + "/d.bin"]
    if files:
        args += ["--vtk", "--final"]
    for k, v in kw.items():
        args += [FLAG[k], repr(float(v))]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    pr = subprocess.run(args, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    raw = open(d + "/d.bin", "rb").read()
    hdr = np.frombuffer(raw[:16], dtype=np.int32)
    off = 16
    n, nf = nx * ny, (nx + 2) * (ny + 2) * 9

    def take(k):
        nonlocal off
        a = np.frombuffer(raw[off:off + 8 * k], dtype=np.float64).copy()
        off += 8 * k
        return a
    out = dict(rho=take(n).reshape(ny, nx), ux=take(n).reshape(ny, nx), uy=take(n).reshape(ny, nx),
               f_current=take(nf).reshape(ny + 2, nx + 2, 9), f_next=take(nf).reshape(ny + 2, nx + 2, 9))
    out["solid"] = np.frombuffer(raw[off:off + n], dtype=np.uint8).reshape(ny, nx).copy()
    off += n
    out["max_velocity"] = take(1)
    out["forces"] = np.loadtxt(d + "/forces.csv", delimiter=",", skiprows=1, ndmin=2)
    out["forces_text"] = np.array(open(d + "/forces.csv").read())
    out["ok"] = np.array(int(hdr[3]))
    if files:   # the reference's output FILES, byte for byte (data: expected outputs of the writers)
        out["velocity_field_csv"] = np.frombuffer(open(d + "/velocity_field.csv", "rb").read(), dtype=np.uint8)
        out["simulation_params_csv"] = np.frombuffer(open(d + "/simulation_params.csv", "rb").read(), dtype=np.uint8)
        out["forces_csv"] = np.frombuffer(open(d + "/forces.csv", "rb").read(), dtype=np.uint8)
        frames = sorted(os.listdir(d + "/vtk_output"))
        out["vtk_names"] = np.array(frames)
        out["vtk_last"] = np.frombuffer(open(d + "/vtk_output/" + frames[-1], "rb").read(), dtype=np.uint8)
        out["stdout"] = np.array(pr.stdout)
    m = re.search(r"unstable at timestep (\d+)", pr.stderr)
    out["unstable_t"] = np.array(int(m.group(1)) if m else -1)
    params = dict(nx=nx, ny=ny, steps=steps, output_frequency=of, tau=0.6, inlet_velocity=0.01333,
                  cylinder_x=0.2, cylinder_y=0.5, cylinder_radius=0.05)
    params.update(kw)
    for k, v in params.items():
        out["p_" + k] = np.array(v)
    return out


def save(name, d, keys):
    keep = {k: v for k, v in d.items() if k in keys or k.startswith("p_")}
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **keep)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_driver missing: run `make -C oracle ref` in the build container")
    os.makedirs(OUT, exist_ok=True)
    macros = {"rho", "ux", "uy", "solid", "max_velocity", "ok"}
    allk = macros | {"f_current", "f_next", "forces", "forces_text"}
    # g9: the reference's output files (forces.csv, vtk_output/lbm_%06d.vtk, velocity_field.csv,
    # simulation_params.csv) of a 64x32 run, 1201 steps, output every 400 (frames at t=400,800,1200)
    d = run_ref(64, 32, 1201, of=400, files=True, inlet_velocity=0.04, cylinder_radius=0.1)
    save("g9_files_64x32_s1201", d, macros | {"velocity_field_csv", "simulation_params_csv", "forces_csv", "vtk_names",
                                              "vtk_last", "stdout"})
    if "--only-g9" in sys.argv:
        return
    for s in (1, 2, 10, 100):
        d = run_ref(128, 32, s, of=50)
        save(f"g1_128x32_s{s}", d, allk if s in (2, 100) else macros | {"forces", "forces_text"})
    for s in (1, 100, 1000):
        d = run_ref(256, 64, s, of=50)
        save(f"g2_256x64_s{s}", d, macros | {"forces", "forces_text"})
    d = run_ref(256, 64, 20000, of=1000, cylinder_x=-1.0, cylinder_radius=0.0)
    d["ux_x128"], d["ux_x192"], d["rho_x128"] = d["ux"][:, 128].copy(), d["ux"][:, 192].copy(), d["rho"][:, 128].copy()
    save("g3_poiseuille_256x64", d, {"ux_x128", "ux_x192", "rho_x128", "max_velocity", "ok"})
    d = run_ref(1024, 256, 3000, of=100, inlet_velocity=0.13020833)
    for k in ("rho", "ux", "uy"):
        d[k + "_ds4"] = d[k][::4, ::4].copy()
        d[k + "_rows"] = d[k][[0, 1, 127, 128, 254, 255], :].copy()
    save("g4_1024x256_re100_s3000", d, {"rho_ds4", "ux_ds4", "uy_ds4", "rho_rows", "ux_rows", "uy_rows",
                                         "forces", "forces_text", "max_velocity", "ok"})
    d = run_ref(64, 32, 50, of=10, cylinder_x=0.0, cylinder_radius=0.1)
    save("g6_inlet_cyl_64x32_s50", d, allk)
    d = run_ref(64, 32, 50, of=10, cylinder_y=0.08, cylinder_radius=0.1)
    save("g7_wall_cyl_64x32_s50", d, allk)
    for tag, tau, u in (("a", 0.502, 0.35), ("b", 0.51, 0.15)):
        d = run_ref(128, 32, 2000, of=50, tau=tau, inlet_velocity=u)
        print("unstable_t =", d["unstable_t"], "ok =", d["ok"])
        save(f"g8{tag}_unstable_128x32", d, {"unstable_t", "ok"})


if __name__ == "__main__":
    main()
