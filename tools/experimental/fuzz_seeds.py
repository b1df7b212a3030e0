#!/usr/bin/env python3
"""One-off fuzz: the random sweep of tests/test_gpu_random.py under other seeds (the committed suite runs seed 20260104).
    python3 tools/experimental/fuzz_seeds.py 1 2 3 4 5"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _progress import Progress      # noqa: E402
PROGRESS = Progress("fuzz_seeds", "_".join(sys.argv[1:]) or "1")      # (sets LBM_TRACE before the library loads)
import tests.test_gpu_random as t   # noqa: E402

bad = total = 0
for seed in [int(v) for v in sys.argv[1:]] or [1]:
    for case in t.cases(seed=seed):
        total += 1
        PROGRESS.start(f"seed {seed} case {case}")
        try:
            t.test_random_case_matches_oracle(case)
            PROGRESS.done()
        except Exception as e:      # noqa: BLE001
            msg = str(e)
            PROGRESS.done(f"FAILED {msg[:200]}")
            if "at least 12 rows" in msg or "needs at least" in msg:
                continue            # (a draw the generator of another seed does not clamp)
            bad += 1
            print(f"seed {seed} case {case}\n{traceback.format_exc(limit=3)}", flush=True)
    print(f"seed {seed}: {total} cases so far, {bad} failing", flush=True)
sys.exit(1 if bad else 0)
