#!/usr/bin/env python3
"""Reads the in-kernel phase records written by `tools/colbench_prof --prof DIR` (lbm_kernel_col.hpp, LBM_COL_PROF) and
prints, per kernel variant: the clock calibration, the distribution of every phase over the blocks (load issue -> level 1
done, then one entry per level / step), how many blocks a CU holds at once and how their phases overlap.
usage: tools/prof_phases.py DIR [--cu N]"""
import csv, glob, os, sys
import numpy as np

def load(path):
    with open(path) as f:
        head = f.readline().strip()
        rows = list(csv.reader(f))
    cols = rows[0]
    data = np.array([[int(v) for v in r] for r in rows[1:]], dtype=np.uint64)
    launch_us = float(head.split("launch_us=")[1])
    return head, launch_us, cols, data

def describe(path, show_cu=None):
    head, launch_us, cols, d = load(path)
    t = d[:, 5:].astype(np.float64)
    used = (t > 0)
    nmarks = int(used.sum(axis=1).max())
    # marks are s_memrealtime samples: 100 MHz, one counter for the whole chip
    us = (t - t[used].min()) * 0.01
    rates = [100.0]
    t0 = 0.0
    span = us[used].max() - us[used].min()
    print(f"== {head[2:]}\n   {len(d)} wave records, {len(set(d[:,0]))} blocks, marks/wave up to {nmarks}; s_memtime {np.mean(rates):.0f} ticks/us "
          f"(per XCD {min(rates):.0f}..{max(rates):.0f}); first mark -> last mark {span:.1f} us, HIP events {launch_us:.1f} us")
    cyc_per_us = 1.0
    t = us
    w0 = d[:, 1] == 0
    tw = t[w0]
    uw = used[w0]
    # phases of wave 0 of every block
    names = []
    ph = []
    for k in range(1, nmarks):
        ok = uw[:, k] & uw[:, k - 1]
        if ok.sum() == 0: continue
        dur = tw[ok, k] - tw[ok, k - 1]
        ph.append((k, dur))
    print("   phase (mark k-1 -> k) of wave 0, us: median / p10 / p90 / max   [blocks]")
    for k, dur in ph:
        print(f"     {k:3d}: {np.median(dur):7.2f} {np.percentile(dur,10):7.2f} {np.percentile(dur,90):7.2f} {dur.max():7.2f}   [{len(dur)}]")
    # per wave index of a block: when the wave starts (relative to its block's first wave) and how long its level 1 takes
    nwv = int(d[:, 1].max()) + 1
    blocks = sorted(set(d[:, 0].tolist()))
    bix = {b: i for i, b in enumerate(blocks)}
    W0 = np.full((len(blocks), nwv), np.nan); W1 = np.full((len(blocks), nwv), np.nan)
    for r in range(len(d)):
        if used[r, 0] and used[r, 1]:
            W0[bix[d[r, 0]], int(d[r, 1])] = t[r, 0]; W1[bix[d[r, 0]], int(d[r, 1])] = t[r, 1]
    b0 = np.nanmin(W0, axis=1)
    print("   per wave index: start after the block's first wave / its level 1 (mark 0 -> 1), us (medians): "
          + " ".join(f"{np.nanmedian(W0[:, k] - b0):.2f}/{np.nanmedian(W1[:, k] - W0[:, k]):.2f}" for k in range(nwv)))
    print(f"   a block's first wave start -> its last wave's level 1 done: median {np.nanmedian(np.nanmax(W1, axis=1) - b0):.2f} us")
    first = tw[:, 0]
    last = np.array([tw[i, uw[i]].max() for i in range(len(tw))])
    life = last - first
    print(f"   block lifetime us: median {np.median(life):.2f}, p10 {np.percentile(life,10):.2f}, p90 {np.percentile(life,90):.2f}; "
          f"first starts: {np.sort(first)[:3].round(2)}, last start {first.max():.2f}, last end {last.max():.2f}")
    # residency per CU
    hw = d[w0, 3]; xcc = d[w0, 2] & 0xf
    cu = ((hw >> 8) & 0xf); sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    key = (xcc * 8 + se) * 32 + sh * 16 + cu
    keys = sorted(set(key.tolist()))
    print(f"   distinct CUs seen: {len(keys)}; blocks per CU: min {min((key==k).sum() for k in keys)}, max {max((key==k).sum() for k in keys)}")
    # time-average number of resident blocks per CU, and of blocks in the load phase (mark 0 -> 1)
    res = []; both_load = []; both_comp = []
    for kk in keys:
        idx = np.nonzero(key == kk)[0]
        ev = []
        for i in idx:
            ev.append((first[i], +1)); ev.append((last[i], -1))
        ev.sort()
        area = 0.0; cur = 0; prev = ev[0][0]
        for tt, s in ev:
            area += cur * (tt - prev); prev = tt; cur += s
        res.append(area / (ev[-1][0] - ev[0][0]))
    print(f"   time-averaged resident blocks per CU: mean {np.mean(res):.2f} (min {np.min(res):.2f}, max {np.max(res):.2f})")
    if show_cu is not None:
        kk = keys[show_cu % len(keys)]
        idx = np.nonzero(key == kk)[0]
        idx = idx[np.argsort(first[idx])]
        print(f"   timeline of CU key {kk} (xcc {kk//256}, se {(kk//32)%8}, cu {kk%32}): block: marks in us")
        for i in idx:
            print("     b%-5d " % d[w0][i, 0] + " ".join(f"{v:7.2f}" for v in tw[i, uw[i]]))
    return cyc_per_us

if __name__ == "__main__":
    dirn = sys.argv[1]
    show = int(sys.argv[sys.argv.index("--cu") + 1]) if "--cu" in sys.argv else None
    for p in sorted(glob.glob(os.path.join(dirn, "prof_*.csv"))):
        describe(p, show)
