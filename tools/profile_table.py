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
        t = ents[0]["hbm_bytes_per_launch"] if ents else None
        fr = t / (r["kernel_ms"] * 1e-3) / 8e12 * (r["iterations_per_launch"] / ents[0]["iterations_per_launch"]) if t else None
        rows.append(f"| `{name}` | {l['config']['nx']}×{l['config']['ny']} {l['dtype']} | `{l['config']['kernel']}` ({lay}) | {num(l['value'])} | "
                    f"{r['kernel_ms'] * 1e3:.1f} | {r['iterations_per_launch']:.2f} | {t / 1e6:.1f} | {fr:.2f} | {r['equiv_144B_frac']:.2f} |" if t else
                    f"| `{name}` | {l['config']['nx']}×{l['config']['ny']} {l['dtype']} | `{l['config']['kernel']}` ({lay}) | {num(l['value'])} | "
                    f"{r['kernel_ms'] * 1e3:.1f} | {r['iterations_per_launch']:.2f} | - | - | {r['equiv_144B_frac']:.2f} |")
    print("| config (`alt_*`: plan forced with `--set`) | grid | kernel (layout) | MLUPS | µs / launch (live HIP events) | iterations / launch | HBM MB / launch of that kernel | `frac` | `equiv_144B_frac` |")
    print("|---|---|---|---|---|---|---|---|---|")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
