#!/usr/bin/env python3
"""Replays a bench window of the HIP path with a PINNED plan, for a rocprofv3 counter pass around it:

    rocprofv3 --pmc FETCH_SIZE -d DIR -o p --output-format csv -- python3 tools/pmc_probe.py --nx 4096 --ny 1024 \
        --precision f64 --arith 1 --plan "layout=1 nt=0 alternate=1 pair_ty=12 xcd=1 fuse=6 deep=7" --steps 20 --reps 12

(hardware counters cannot be read inside a timed run: bench.py starts this program as a CHILD under rocprofv3, one pass per
counter group, after its timed region — so the bytes in its `roofline` belong to the binary and the plan it has just timed.)
The plan comes from lbm_plan_options() of the benchmarked context; tune=0 pins it. Each repetition is one lbm_step(steps) call,
i.e. the bench's own launch sequence (20 = 7+7+6 on the six-iteration register plan). Prints one JSON line: what ran."""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "highperformancecomputing-latticeboltzmannmethod_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, required=True)
    ap.add_argument("--ny", type=int, required=True)
    ap.add_argument("--re", type=float, default=200.0)
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--arith", type=int, default=1)
    ap.add_argument("--plan", default="", help="lbm_plan_options() of the context to reproduce ('' = let this process measure)")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--warm", type=int, default=2, help="untimed repetitions first (their dispatches are counted too: same kernels)")
    a = ap.parse_args()
    lbm = importlib.import_module(PKG)
    u_in = a.re * ((0.6 - 0.5) / 3.0) / (2.0 * 0.05 * a.ny)
    opts = {"arith": a.arith, "trailing_pair": 1}
    if a.plan:
        opts["tune"] = 0
        for kv in a.plan.split():
            k, v = kv.split("=")
            opts[k] = int(v)
    with lbm.Context(a.nx, a.ny, tau=0.6, inlet_velocity=u_in, precision=a.precision, options=opts) as c:
        c.initialise()
        c.set_option("timing", 1)
        launches = iters = 0
        ms = 0.0
        for k in range(a.warm + a.reps):
            c.step(a.steps, 0)
            c.sync()
            t, nl, ni = c.last_step_stats()
            launches += nl
            iters += ni
            if k >= a.warm:
                ms += t
        bad = c.first_unstable_step()
        print(json.dumps({"probe": True, "kernel": c.kernel_name(), "plan": c.plan(), "plan_options": c.plan_options(),
                          "build_id": lbm.build_id(), "calls": a.warm + a.reps, "steps_per_call": a.steps,
                          "launches": launches, "iterations": iters, "unstable": bad,
                          # HIP-event time of the counted calls (meaningful only when this program runs WITHOUT a counter pass around it)
                          "ms_per_iteration": ms / max(1, a.reps * a.steps)}), flush=True)
    return 0 if bad == -1 else 1


if __name__ == "__main__":
    sys.exit(main())
