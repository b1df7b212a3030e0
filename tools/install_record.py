#!/usr/bin/env python3
"""Installs a tools/profile_round4.sh record directory (scratch, e.g. gpurun_out/r04/record3) as the committed record of the round:
copies the bench lines, per-grid summaries and the traffic table into profiles/rNN + profiles/traffic.json and regenerates the
generated rows of profiles/rNN/README.md (section 4: tools/record_table.py, section 5: tools/traffic_table.py).
    python3 tools/install_record.py gpurun_out/r04/record3 profiles/r04"""
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    for name in os.listdir(src):
        p = os.path.join(src, name)
        if os.path.isdir(p):
            shutil.copytree(p, os.path.join(dst, name), dirs_exist_ok=True)
        elif name == "traffic.json":
            shutil.copy(p, os.path.join(ROOT, "profiles", "traffic.json"))
        elif not name.endswith(".err"):
            shutil.copy(p, os.path.join(dst, name))
    build = json.loads(open(os.path.join(dst, "bench_default.json")).read().strip().splitlines()[-1])["config"]["build_id"]
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "record_table.py"), dst], capture_output=True, text=True, check=True).stdout.splitlines()
    rows = {ln.split("|")[1].strip(): ln for ln in gen if ln.startswith("| `")}
    tab = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_table.py"), os.path.join(ROOT, "profiles", "traffic.json")],
                         capture_output=True, text=True, check=True).stdout.rstrip("\n").splitlines()
    readme = os.path.join(dst, "README.md")
    L = open(readme).read().splitlines()
    out, sec = [], None
    for ln in L:
        m = re.match(r"## (\d+)\.", ln)
        if m:
            sec = int(m.group(1))
            if sec == 4:
                ln = re.sub(r"build `[0-9a-f]{16}`", f"build `{build}`", ln)
        if sec == 4 and ln.startswith("| `") and ln.split("|")[1].strip() in rows:
            ln = rows[ln.split("|")[1].strip()]
        out.append(ln)
    i5 = next(i for i, ln in enumerate(out) if ln.startswith("## 5."))
    ts = next(i for i in range(i5, len(out)) if out[i].startswith("| grid |"))
    te = ts
    while te < len(out) and out[te].startswith("|"):
        te += 1
    out[ts:te] = tab
    open(readme, "w").write("\n".join(out) + "\n")
    print(f"installed build {build}: {len(rows)} generated rows in section 4, {len(tab) - 2} table rows in section 5")


if __name__ == "__main__":
    main()
