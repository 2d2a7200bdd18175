#!/usr/bin/env python
"""Per-kernel table of a scripts/pmc_cmd.sh visit: mean duration, HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of
MI355X_MICROARCH.md), SQ counters per launch.  Also writes gpurun_out/<tag>/pmc_table.json.  usage: pmc_table.py <tag> [name filter]"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "bds::"
src = os.path.join("gpurun_out", tag)


def short(n):
    return n.split("(")[0].replace("void ", "").replace("bds::", "")[:60]


tab = collections.defaultdict(dict)
f = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
if f:
    for r in csv.DictReader(open(f[0])):
        if flt in r["Name"]:
            tab[short(r["Name"])].update(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3)
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    f = glob.glob(os.path.join(src, d, "*counter_collection.csv"))
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if flt in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            tab[k][c] = sum(v) / len(v)
for k, t in tab.items():
    if "FETCH_SIZE" in t or "WRITE_SIZE" in t:
        t["hbm_MB"] = (2.0 * t.get("FETCH_SIZE", 0.0) + t.get("WRITE_SIZE", 0.0)) * 1024.0 / 1e6
    if "SQ_BUSY_CYCLES" in t and "SQ_ACTIVE_INST_VALU" in t and t.get("SQ_WAVE_CYCLES"):
        t["valu_busy_frac_of_wave_cycles"] = 4.0 * t["SQ_ACTIVE_INST_VALU"] / t["SQ_WAVE_CYCLES"] if t["SQ_WAVE_CYCLES"] else None
json.dump(tab, open(os.path.join(src, "pmc_table.json"), "w"), indent=1, sort_keys=True)
cols = ["calls", "avg_us", "hbm_MB", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES",
        "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"]
for k, t in sorted(tab.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 1)):
    print(k)
    print("   " + "  ".join(f"{c}={t[c]:.4g}" for c in cols if c in t))
