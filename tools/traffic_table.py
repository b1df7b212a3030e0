#!/usr/bin/env python3
"""Markdown table of profiles/traffic.json (tools/collect_live_traffic.py): one row per (grid, candidate kernel) with its unprofiled
rate, the counter bytes and the fractions bench.py derives from them.   python3 tools/traffic_table.py [profiles/traffic.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATE = {"f64": 256 * 4 * 16 * 2.4e9, "f32": 256 * 4 * 32 * 2.4e9}


def main():
    tj = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "traffic.json")))
    print("| grid | kernel (layout, arithmetic) | MLUPS (unprofiled) | µs / launch | it / launch | HBM MB / launch: fetch + write | B / update | `frac_hbm_measured` | VALU lane-instr / update | `frac_valu` | MLUPS per GB/s | build |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for key, ents in tj.items():
        nx, ny = (int(v) for v in key.split("_")[0].split("x"))
        prec = key.split("_")[1]
        for e in sorted(ents, key=lambda e: -(e.get("mlups_unprofiled") or 0)):
            ipl = e["iterations_per_launch"]
            us = (e.get("us_per_iteration_unprofiled") or 0) * ipl
            bpu = e["hbm_bytes_per_launch"] / (nx * ny * ipl)
            frac = e["hbm_bytes_per_launch"] / (us * 1e-6) / 8e12 if us else float("nan")
            lane = (e.get("valu_insts_per_launch") or 0) * 64 / (nx * ny * ipl)
            fv = (e.get("mlups_unprofiled") or 0) * 1e6 / (RATE[prec] / lane) if lane else float("nan")
            print(f"| {key} | `{e['kernel']}` ({e.get('layout', '')}, {e.get('arithmetic', '')}) | {e.get('mlups_unprofiled') or 0:,.0f} | {us:.1f} | {ipl:.2f} | "
                  f"{e['fetch_bytes_corrected'] / 1e6:.0f} + {e['write_bytes'] / 1e6:.0f} | {bpu:.1f} | {frac:.2f} | {lane:.0f} | {fv:.2f} | {1000 / bpu:.1f} | `{e.get('build_id', '')[:8]}` |".replace(",", " "))


if __name__ == "__main__":
    main()
