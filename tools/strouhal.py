#!/usr/bin/env python3
"""tools/strouhal.py — Strouhal number of a cylinder run from forces.csv + simulation_params.csv (the files written
by host/lbm_solver, file-compatible with the reference's). Same estimator and parameters as the reference's
post-processing (scripts/lift.py:60-101: peaks of lift_coeff for timestep >= 30000, scipy find_peaks with
prominence 0.5, St = D / (U * mean peak spacing)), without the plotting."""
import sys

import numpy as np
from scipy.signal import find_peaks


def strouhal(forces_csv="forces.csv", params_csv="simulation_params.csv", start=30000, prominence=0.5):
    f = np.loadtxt(forces_csv, delimiter=",", skiprows=1)
    params = dict(line.strip().split(",") for line in open(params_csv).read().splitlines()[1:])
    u, d = float(params["inlet_velocity"]), 2.0 * float(params["cylinder_radius"])
    sel = f[:, 0] >= start
    t, cl = f[sel, 0], f[sel, 4]
    peaks, _ = find_peaks(cl, prominence=prominence)
    if len(peaks) < 2:
        raise RuntimeError(f"only {len(peaks)} peaks after timestep {start}")
    period = float(np.mean(np.diff(t[peaks])))
    return dict(strouhal=d / (u * period), period=period, peaks=int(len(peaks)), U=u, D=d,
                reynolds=float(params["reynolds_number"]), mean_cd=float(np.mean(f[sel, 3])),
                cl_amplitude=float(0.5 * (cl.max() - cl.min())))


if __name__ == "__main__":
    r = strouhal(*sys.argv[1:3])
    print(f"Re = {r['reynolds']:.1f}  U = {r['U']:.4f}  D = {r['D']:.0f}  peaks = {r['peaks']}  period = {r['period']:.2f} steps")
    print(f"Strouhal number St = f*D/U = {r['strouhal']:.4f}   mean Cd = {r['mean_cd']:.4f}   Cl amplitude = {r['cl_amplitude']:.4f}")
