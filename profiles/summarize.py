#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (profiles/collect.sh) into per-kernel tables: launch count, average
duration, and per-launch PMC sums for the MLP kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, suffix):
    hits = glob.glob(os.path.join(out, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("ibl::(anonymous namespace)::", "").replace("void ", "")
    return name[:70]


st = find("kt", "kernel_stats.csv")
if st:
    print("== kernel-trace stats (%s)" % st)
    rows = list(csv.DictReader(open(st)))
    for r in rows[:14]:
        print("%-72s calls=%6s total_ms=%10.3f avg_ms=%9.4f pct=%6s" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
for sub in ("pmc_mfma", "pmc_sq", "pmc_fetch", "pmc_write"):
    f = find(sub, "counter_collection.csv")
    if not f:
        continue
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    print("== %s (per-launch averages)" % sub)
    for k in agg:
        n = max(len(cnt[k]), 1)
        print("%-72s launches=%4d " % (k, n) + " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(agg[k].items())))
