"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_params(g):
    """kwargs for oracle.make_params / the HIP Params from a fixture's p_* entries."""
    return dict(nx=int(g["p_nx"]), ny=int(g["p_ny"]), tau=float(g["p_tau"]),
                inlet_velocity=float(g["p_inlet_velocity"]), cylinder_x=float(g["p_cylinder_x"]),
                cylinder_y=float(g["p_cylinder_y"]), cylinder_radius=float(g["p_cylinder_radius"]))


def linf_rel(a, b, scale=None):
    """L-inf(a-b) / L-inf(b): the 'relative L∞/L∞' norm of BASELINE.md §3. `scale` overrides the
    denominator (velocity components are normalised by max|u|, not by their own possibly-zero maximum)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b)) if scale is None else scale
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def macro_errors(rho, ux, uy, g_rho, g_ux, g_uy):
    """(err_rho, err_u): rho relative to max|rho|; both velocity components relative to max|u| of the expected field."""
    uscale = float(np.max(np.sqrt(np.asarray(g_ux) ** 2 + np.asarray(g_uy) ** 2)))
    return linf_rel(rho, g_rho), max(linf_rel(ux, g_ux, uscale), linf_rel(uy, g_uy, uscale))


def record(name, **values):
    """Append one measured-error line to gpurun_out/parity_measured.jsonl (merged back from the GPU box; the round's copy is
    committed under profiles/). Best effort: a read-only tree must not fail a test."""
    import json
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=name, **values)) + "\n")
    except OSError:
        pass
