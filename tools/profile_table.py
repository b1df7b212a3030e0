#!/usr/bin/env python3
"""Markdown table of a round's per-configuration profile directories (profiles/rNN/<config>/bench_line.json joined with
profiles/traffic.json): python tools/profile_table.py profiles/r02"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def num(v):
    return f"{v:,.0f}".replace(",", " ")


def main():
    rdir = sys.argv[1]
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    rows = []
    for f in sorted(glob.glob(os.path.join(rdir, "*", "bench_line.json")), key=lambda p: (os.path.basename(os.path.dirname(p)).replace("alt_", ""), "alt_" in p)):
        name = os.path.basename(os.path.dirname(f))
        try:
            l = json.loads(open(f).read())
        except Exception:
            continue
        r = l["roofline"]
        key = f"{l['config']['nx']}x{l['config']['rows_per_gpu']}_{l['dtype']}"
        pl = l["config"]["plan"]
        lay = "row-interleaved" if pl.startswith("fixed by options") else pl.split("/")[0]
        ents = [e for e in tj.get(key, []) if e["kernel"].replace(" ", "") == l["config"]["kernel"].replace(" ", "")]
        ents.sort(key=lambda e: (e.get("source", "").split("/")[-2] != name, e.get("layout", "") != lay))
        e0 = ents[0] if ents else None
        cells = l["config"]["nx"] * l["config"]["rows_per_gpu"]
        ipl = r["iterations_per_launch"]
        secs = r["kernel_ms"] * 1e-3
        base = f"| `{name}` | {l['config']['nx']}×{l['config']['ny']} {l['dtype']} | `{l['config']['kernel']}` ({lay}) | {num(l['value'])} | {r['kernel_ms'] * 1e3:.1f} | {ipl:.2f} | "
        if e0:
            t = e0["hbm_bytes_per_launch"] * ipl / e0["iterations_per_launch"]
            gbs = t / secs / 1e9
            rate = {"f64": 256 * 4 * 16 * 2.4e9, "f32": 256 * 4 * 32 * 2.4e9}[l["dtype"]]
            lane = (e0.get("valu_insts_per_launch") or 0) * 64.0 / (cells * e0["iterations_per_launch"])
            fv = (cells * ipl / secs) / (rate / lane) if lane else None
            ldsi = (e0.get("lds_insts_per_launch") or 0) * 64.0 / (cells * e0["iterations_per_launch"])
            rows.append(base + f"{e0['fetch_bytes_corrected'] / 1e6:.0f} + {e0['write_bytes'] / 1e6:.0f} | {t / (cells * ipl):.1f} | {gbs / 8000:.2f} | "
                        + (f"{lane:.0f} | {fv:.2f} | {ldsi:.1f} | " if lane else "- | - | - | ") + f"{r.get('frac_144B', r.get('equiv_144B_frac', 0)):.2f} | {cells * ipl / secs / 1e6 / gbs:.1f} |")
        else:
            rows.append(base + f"- | - | - | - | - | - | {r.get('frac_144B', r.get('equiv_144B_frac', 0)):.2f} | - |")
    print("| config (`alt_*`: plan forced with `--set`) | grid | kernel (layout) | MLUPS | µs / launch (live HIP events) | iterations / launch | HBM MB / launch: fetch + write | HBM B / update | `frac_hbm_measured` | VALU lane-instr / update | `frac_valu` | LDS lane-instr / update | `frac_144B` | MLUPS per GB/s |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
