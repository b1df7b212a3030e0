#!/usr/bin/env python3
"""Markdown summary of a tools/profile_round4.sh record directory (profiles/rNN): the two bench lines, and per BASELINE.json grid the
bench line of the measured plan beside the rocprofv3 --kernel-trace --stats mean of the same command.
    python3 tools/record_table.py profiles/r04"""
import csv
import json
import os
import sys


def line_of(path):
    for ln in reversed(open(path).read().strip().splitlines()):
        if ln.startswith("{") and '"metric"' in ln:
            return json.loads(ln)
    raise SystemExit(f"no bench line in {path}")


def describe(d):
    r, sp, oa, cb = d["roofline"], d.get("single_precision_variant") or {}, d.get("other_arithmetic") or {}, d.get("cpu_baseline") or {}
    t = r.get("traffic")
    s = (f"**{d['value']:,.0f} MLUPS** ({d['steps']}-step window), sustained {d['sustained']['value']:,.0f}; kernel `{r['kernel']}` {r['kernel_ms'] * 1e3:.1f} µs per launch of "
         f"{r['iterations_per_launch']:.4f} iterations; `frac` (144 B) {r['frac']:.4f}")
    if t:
        s += (f"; live passes: {t / 1e6:.0f} MB per launch = {r['hbm_bytes_per_update']:.2f} B per update, `frac_hbm_measured` {r['frac_hbm_measured']:.4f}, "
              f"`frac_valu` {r['frac_valu']:.4f}, bound {r['bound']}, {r['mlups_per_gbs']:.2f} MLUPS per GB/s; `traffic_build_id` {r.get('traffic_build_id')} "
              f"(`config.build_id` {d['config']['build_id']}), stale {r.get('stale')}")
    s += f"; plan: {d['config']['plan'].split(' (')[0]}"
    if oa:
        s += f"; strict arithmetic {oa['value']:,.0f} (`{oa['kernel']}`)"
    if sp:
        s += (f"; 16384×4096 fp32 {sp['value']:,.0f} (`{sp.get('kernel')}`) with **{sp.get('mlups_per_gbs')} MLUPS per GB/s**, "
              f"`frac_hbm_measured` {sp.get('frac_hbm_measured')}")
    if cb:
        s += f"; CPU reference {cb['value']} MLUPS ({cb['cores']} cores)"
    return s.replace(",", " ")


def main():
    root = sys.argv[1]
    print("| file | what |\n|---|---|")
    for f, cmd in (("bench_default.json", "python bench.py"), ("bench_driver_style.json", "python bench.py --gpus 1 --steps 20 --warmup 5")):
        print(f"| `{f}` | `{cmd}`: {describe(line_of(os.path.join(root, f)))} |")
    print("\n| config | grid | kernel | plan | MLUPS (window) | sustained | µs / launch (HIP events) | `kernel_stats.csv` mean µs |\n|---|---|---|---|---|---|---|---|")
    for name in sorted(os.listdir(root)):
        p = os.path.join(root, name, "bench_line.json")
        if not os.path.isfile(p):
            continue
        d = line_of(p)
        k = d["roofline"]["kernel"]
        mean = calls = None
        for row in csv.DictReader(open(os.path.join(root, name, "kernel_stats.csv"))):
            if row["Name"].replace("lbmk::", "").replace(" ", "").startswith("void" + k.replace(" ", "")) or k.replace(" ", "") in row["Name"].replace(" ", ""):
                mean, calls = float(row["AverageNs"]) / 1e3, int(row["Calls"])
                break
        c = d["config"]
        print(f"| `{name}` | {c['nx']}×{c['ny']} {d['dtype']} | `{k}` | {c['plan'].split(' (')[0]} | {d['value']:,.0f} | {d['sustained']['value']:,.0f} | "
              f"{d['roofline']['kernel_ms'] * 1e3:.1f} | {mean:.2f} ({calls} calls) |".replace(",", " "))


if __name__ == "__main__":
    main()
