#!/usr/bin/env python3
"""Builds profiles/traffic.json from the PMC passes of tools/pmc_traffic.sh / tools/pmc_passes.sh.

    python tools/collect_traffic.py ROUND_DIR...   (e.g. profiles/r02 profiles/r03: every sub-directory with a
                                                    pmc_summary.csv and a bench_line.json contributes the fused step kernel
                                                    of that run; later directories win)

HBM bytes per launch = FETCH_SIZE [KiB] x 1024 x 2 + WRITE_SIZE [KiB] x 1024: gfx950 tallies the 128-byte requests of a
coalesced read stream at 64 B (MI355X_MICROARCH.md, HBM section; calibrated in round 1 on the single-iteration kernels,
whose known 302 MB read stream is reported as 150-154 MiB); WRITE_SIZE reads exactly."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = {}
    # every round directory given contributes; a later one replaces an earlier one's entry for the same (grid, kernel)
    for summ in [f for rdir in sys.argv[1:] for f in sorted(glob.glob(os.path.join(rdir, "*", "pmc_summary.csv")))]:
        d = os.path.dirname(summ)
        try:
            line = json.loads(open(os.path.join(d, "bench_line.json")).read())
        except Exception:
            continue
        kernel = line["config"]["kernel"]
        vals = {}
        with open(summ, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["kernel"].replace(" ", "") == kernel.replace(" ", ""):
                    vals[r["counter"]] = (float(r["mean"]), int(r["launches"]))
        if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
            continue
        fetch = vals["FETCH_SIZE"][0] * 1024 * 2
        write = vals["WRITE_SIZE"][0] * 1024
        key = f"{line['config']['nx']}x{line['config']['rows_per_gpu']}_{line['dtype']}"
        out[key] = [e for e in out.get(key, []) if e["kernel"].replace(" ", "") != kernel.replace(" ", "")]
        out[key].append({
            "kernel": kernel, "hbm_bytes_per_launch": int(fetch + write), "fetch_bytes_corrected": int(fetch),
            "write_bytes": int(write), "launches_sampled": vals["FETCH_SIZE"][1],
            "iterations_per_launch": round(line["roofline"]["iterations_per_launch"]),
            "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
            "valu_insts_per_launch": vals.get("SQ_INSTS_VALU", (None,))[0], "lds_insts_per_launch": vals.get("SQ_INSTS_LDS", (None,))[0],
            "lds_bank_conflict_cycles": vals.get("SQ_LDS_BANK_CONFLICT", (None,))[0], "wait_inst_lds": vals.get("SQ_WAIT_INST_LDS", (None,))[0],
            "active_inst_valu": vals.get("SQ_ACTIVE_INST_VALU", (None,))[0], "busy_cycles": vals.get("SQ_BUSY_CYCLES", (None,))[0],
            "wave_cycles": vals.get("SQ_WAVE_CYCLES", (None,))[0], "wait_any": vals.get("SQ_WAIT_ANY", (None,))[0],
            "arithmetic": line["config"].get("arithmetic", "")[:20], "layout": (lambda pl: "row-interleaved" if pl.startswith("fixed by options") else pl.split("/")[0])(line["config"].get("plan", "")),   # (the forced alt_* runs use --set layout=1)
            "source": os.path.relpath(summ, ROOT),
            "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters, separate passes over `python3 bench.py ...` "
                      "(tools/pmc_traffic.sh); KiB -> B; FETCH_SIZE x2 (gfx950 correction)"})
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    for k, v in out.items():
        for e in v:
            print(k, e["kernel"], f"{e['hbm_bytes_per_launch'] / 1e6:.1f} MB/launch", f"(fetch {e['fetch_bytes_corrected'] / 1e6:.1f} + write {e['write_bytes'] / 1e6:.1f})")


if __name__ == "__main__":
    main()
