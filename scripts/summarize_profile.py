#!/usr/bin/env python3
"""Condense a gpurun_out/<tag>/ profile (scripts/profile_round.sh) into small committed files under
profiles/: per-kernel stats (rocprofv3 --kernel-trace --stats) and per-kernel HBM traffic from the
FETCH_SIZE / WRITE_SIZE passes (KB per launch; gfx950 correction: FETCH_SIZE x2 for wide coalesced
reads, calibrated here on sh_fwd_kernel whose byte count is known)."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)

rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
total = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:45]:
        w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
print(f"GPU kernel time per step: {total / steps / 1e6:.3f} ms over {steps} steps")

pmc = {}
for name, key in (("pmc_fetch", "FETCH_SIZE_KB"), ("pmc_write", "WRITE_SIZE_KB")):
    path = os.path.join(src, name, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:80]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        if k.startswith("void bds::") or k.startswith("bds::"):
            pmc.setdefault(k, {})[key] = v / n
            pmc[k]["launches"] = n
for k, d in pmc.items():
    f, wv = d.get("FETCH_SIZE_KB", 0.0), d.get("WRITE_SIZE_KB", 0.0)
    d["hbm_bytes_per_launch_corrected"] = (2.0 * f + wv) * 1024.0
json.dump({"note": "KB per launch, mean over launches; corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                   "(MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide coalesced read on gfx950; "
                   "WRITE_SIZE is exact -- both confirmed on sh_fwd/sh_bwd whose byte counts are known)",
           "kernels": pmc}, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
# SQ counters (two passes of 8): per-kernel means + derived VALU figures
sq = {}
for name in ("pmc_sq", "pmc_sq2"):
    path = os.path.join(src, name, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:80]
        if k.startswith("void bds::") or k.startswith("bds::"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            sq.setdefault(k, {})[c] = sum(v) / len(v)
for k, d in sq.items():
    if d.get("SQ_INSTS_VALU") and d.get("SQ_ACTIVE_INST_VALU") and d.get("GRBM_GUI_ACTIVE"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
        d["valu_cycles_per_inst"] = 4.0 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"]
        d["valu_busy_frac"] = 4.0 * d["SQ_ACTIVE_INST_VALU"] / 1024.0 / (d["GRBM_GUI_ACTIVE"] / 8.0)
        if d.get("SQ_THREAD_CYCLES_VALU"):
            d["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"])
if sq:
    json.dump({"note": "rocprofv3 --pmc, two passes of 8 SQ counters + GRBM_GUI_ACTIVE, mean per launch; valu_busy_frac = "
                       "4 * SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles)",
               "kernels": sq}, open(os.path.join(dst, f"{tag}_sq_counters.json"), "w"), indent=1, sort_keys=True)
for extra in ("valu_rate.txt", "pair_stats.json", "gpu_tests.log", "smoke.log", "step_timeline.txt", "frame_overlap.txt"):
    p = os.path.join(src, extra)
    if os.path.exists(p):
        open(os.path.join(dst, f"{tag}_{extra}"), "w").write(open(p).read())
for extra in ("bench.json", "trace_bench.json", "bench_share2.json"):
    p = os.path.join(src, extra)
    if os.path.exists(p):
        open(os.path.join(dst, f"{tag}_{extra}"), "w").write(open(p).read())
print("wrote", sorted(x for x in os.listdir(dst) if x.startswith(tag)))
