#!/bin/bash
# Resource usage (VGPR/SGPR/spills/LDS/occupancy) of the kernels matching $1 (default: k_step_slide), from hipcc remarks.
PAT=${1:-k_step_slide}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -S --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage /root/repo/highperformancecomputing-latticeboltzmannmethod_amd/csrc/lbm_hip.hip -o /tmp/lbm.s 2>&1 \
  | grep -A9 "Function Name: .*$PAT" | grep -E "Name|VGPRs:|Spill|SGPRs:|Occ|Scratch|LDS" | sed 's/.*remark: *//; s/ \[-Rpass.*//' \
  | awk '/Function Name/{if (l) print l; l=$3} !/Function Name/{l=l" | "$0} END{print l}' | c++filt | sed 's/lbmk:://g; s/KArgs<[a-z]*>, K2Extra<[a-z]*>, SlideArgs//'
