#!/bin/bash
# Resource usage of the kernels matching $2 in a hipcc remarks file $1 (-Rpass-analysis=kernel-resource-usage).
python3 - "$1" "${2:-k_stepc_col}" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
blocks = re.split(r"remark: Function Name: ", txt)[1:]
keys = [("VGPR", "VGPRs"), ("spill", "VGPRs Spill"), ("SGPR", "SGPRs"), ("sspill", "SGPRs Spill"),
        ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"), ("LDS", r"LDS Size \[bytes/block\]")]
for b in blocks:
    name = b.split()[0]
    if pat not in name: continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("lbmk::", "").split("(")[0].replace("void ", "")
    out = []
    for label, k in keys:
        m = re.search(k + r": (\d+)", b)
        out.append("%s %3s" % (label, m.group(1) if m else "?"))
    print("%-58s %s" % (dem, "  ".join(out)))
PY
