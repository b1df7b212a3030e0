#!/usr/bin/env python3
"""Summarises rocprofv3 counter-collection CSVs: mean counter value per kernel over its dispatches.

    python tools/pmc_summary.py DIR [--match substring] [--csv out.csv]

DIR is searched recursively for *counter_collection.csv (one rocprofv3 --pmc pass each) and *kernel_stats.csv."""
import argparse
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("lbmk::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--match", default="k_step")
    ap.add_argument("--csv", default=None)
    a = ap.parse_args()
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                if a.match not in k:
                    continue
                e = acc[(k, r["Counter_Name"])]
                e[0] += float(r["Counter_Value"])
                e[1] += 1
    rows = [(k, c, n, s / n) for (k, c), (s, n) in sorted(acc.items())]
    out = open(a.csv, "w") if a.csv else sys.stdout
    out.write("kernel,counter,launches,mean\n")
    for k, c, n, m in rows:
        out.write(f"\"{k}\",{c},{n},{m:.6g}\n")
    if a.csv:
        out.close()
    for f in glob.glob(os.path.join(a.dir, "**", "*kernel_stats.csv"), recursive=True):
        print("==", f)
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if a.match in r.get("Name", ""):
                    print(f"  {short(r['Name'])}: calls {r['Calls']}, mean {float(r['AverageNs']) / 1e3:.2f} us, "
                          f"min {float(r['MinNs']) / 1e3:.2f}, max {float(r['MaxNs']) / 1e3:.2f}")


if __name__ == "__main__":
    main()
