#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/validate_round.sh OUTDIR
# The reference's published case (its SimulationParams defaults: 2048x512, tau 0.6, u_in 0.1333 -> Re 204.7, 120 000 steps, forces every
# 140) through host/lbm_solver on the binary in the tree: contracted arithmetic on one strip, strict arithmetic on two strips; the Strouhal
# number by the reference's own estimator (tools/strouhal.py); and the fp32 variant (no reference counterpart) on the same case. Writes OUTDIR/{run.txt,forces.csv,simulation_params.csv,strouhal.txt} and
# OUTDIR/strips/{run.txt,strouhal.txt}; every step is logged at once to OUTDIR/progress.log.
set -e
OUT=$PWD/$1
S=$PWD/highperformancecomputing-latticeboltzmannmethod_amd/host/lbm_solver
mkdir -p "$OUT/strips" "$OUT/work1" "$OUT/work2"
say() { echo "$(date +%T) $*" >> "$OUT/progress.log"; }
say "contracted, one strip"
t0=$(date +%s.%N); ( cd "$OUT/work1" && "$S" --no-vtk --inlet-velocity 0.1333 --contracted > "$OUT/run.txt" 2>&1 ); python3 -c "import sys,time; print(f'wall: {time.time() - float(sys.argv[1]):.1f} s')" $t0 >> "$OUT/run.txt"
cp "$OUT/work1/forces.csv" "$OUT/work1/simulation_params.csv" "$OUT/"
python3 tools/strouhal.py "$OUT/forces.csv" "$OUT/simulation_params.csv" > "$OUT/strouhal.txt"
say "strict, two strips"
t0=$(date +%s.%N); ( cd "$OUT/work2" && "$S" --no-vtk --inlet-velocity 0.1333 --strips 2 > "$OUT/strips/run.txt" 2>&1 ); python3 -c "import sys,time; print(f'wall: {time.time() - float(sys.argv[1]):.1f} s')" $t0 >> "$OUT/strips/run.txt"
python3 tools/strouhal.py "$OUT/work2/forces.csv" "$OUT/work2/simulation_params.csv" > "$OUT/strips/strouhal.txt"
cmp "$OUT/work1/forces.csv" "$OUT/work2/forces.csv" > /dev/null 2>&1 && echo "forces.csv of the two runs: byte-identical" >> "$OUT/strips/strouhal.txt" || echo "forces.csv of the two runs differ (strict against contracted arithmetic: expected beyond the printed digits only if a digit flips)" >> "$OUT/strips/strouhal.txt"
say "fp32 variant, contracted, one strip"
mkdir -p "$OUT/fp32" "$OUT/work3"
t0=$(date +%s.%N); ( cd "$OUT/work3" && "$S" --no-vtk --inlet-velocity 0.1333 --fp32 --contracted > "$OUT/fp32/run.txt" 2>&1 ); python3 -c "import sys,time; print(f'wall: {time.time() - float(sys.argv[1]):.1f} s')" $t0 >> "$OUT/fp32/run.txt"
python3 tools/strouhal.py "$OUT/work3/forces.csv" "$OUT/work3/simulation_params.csv" > "$OUT/fp32/strouhal.txt"
cp "$OUT/work3/forces.csv" "$OUT/fp32/forces.csv"
rm -rf "$OUT/work1" "$OUT/work2" "$OUT/work3"
say done
cat "$OUT/strouhal.txt" "$OUT/strips/strouhal.txt" "$OUT/fp32/strouhal.txt"
