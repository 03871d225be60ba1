#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (profiles/collect.sh) into per-kernel tables: launch count, average
duration, per-launch PMC sums, and derived figures for the MLP kernel (clock, MFMA utilisation, HBM
traffic).  Also writes <out>/pmc.json, which bench.py reads for roofline.traffic.

Counter notes (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE is reported once per XCD (8 rows per dispatch) —
busy cycles = sum / 8; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (busy cycles * 1024 SIMDs); FETCH_SIZE and
WRITE_SIZE are in KiB and FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (corrected here).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
N_XCD, N_SIMD = 8, 1024


def find(sub, suffix):
    hits = glob.glob(os.path.join(out, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.replace("ibl::(anonymous namespace)::", "").replace("void ", "")[:70]


stats = {}
st = find("kt", "kernel_stats.csv")
if st:
    print("== kernel-trace stats (%s)" % st)
    for r in list(csv.DictReader(open(st)))[:14]:
        k = short(r["Name"])
        stats[k] = dict(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6, pct=float(r["Percentage"]))
        print("%-72s calls=%6s total_ms=%10.3f avg_ms=%9.4f pct=%6s" % (
            k, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
pmc = defaultdict(dict)
for sub in ("pmc_mfma", "pmc_sq", "pmc_fetch", "pmc_write"):
    f = find(sub, "counter_collection.csv")
    if not f:
        continue
    agg, cnt, dur = defaultdict(lambda: defaultdict(float)), defaultdict(set), defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print("== %s (per-launch averages)" % sub)
    for k in agg:
        n = max(len(cnt[k]), 1)
        for c, v in agg[k].items():
            pmc[k][c] = v / n
        pmc[k]["_ms_" + sub] = sum(dur[k].values()) / n
        print("%-72s launches=%4d " % (k, n) + " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(agg[k].items())))
print("== derived (MLP kernels)")
derived = {}
for k, v in pmc.items():
    if "mlp_kernel" not in k and "k_trunk_fp32" not in k:
        continue
    d = {}
    if "GRBM_GUI_ACTIVE" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / N_XCD
        d["busy_cycles"] = cyc
        d["clock_ghz"] = cyc / (v["_ms_pmc_mfma"] * 1e6)
        d["mfma_util"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * N_SIMD)
    if "FETCH_SIZE" in v:
        d["hbm_read_bytes"] = 2.0 * v["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in v:
        d["hbm_write_bytes"] = v["WRITE_SIZE"] * 1024
    if k in stats:
        d["avg_ms"] = stats[k]["avg_ms"]
        d["calls"] = stats[k]["calls"]
    derived[k] = d
    print("%-40s " % k[:40] + " ".join("%s=%.4g" % kv for kv in d.items()))
print("== derived (per-ray kernels: achieved HBM GB/s = (2 x FETCH_SIZE + WRITE_SIZE) KiB / launch time)")
for k, v in pmc.items():
    if "mlp_kernel" in k or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v or k not in stats:
        continue
    byts = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    gbs = byts / (stats[k]["avg_ms"] * 1e-3) / 1e9
    derived[k] = dict(hbm_bytes=byts, avg_ms=stats[k]["avg_ms"], hbm_gb_s=gbs, calls=stats[k]["calls"])
    print("%-40s bytes/launch=%.3g avg_ms=%.4f -> %.0f GB/s" % (k[:40], byts, stats[k]["avg_ms"], gbs))
json.dump(dict(kernel_stats=stats, pmc=pmc, derived=derived), open(os.path.join(out, "pmc.json"), "w"), indent=1)
