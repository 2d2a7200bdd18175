#!/usr/bin/env python3
"""Summary of tests/test_gpu_25_gs_random_sweep.py's log (gpurun_out/gs_parity_sweep.json): worst errors of the HIP path against the
float64 oracle over all cases, next to the float32 oracle's on the same cases, and how many cases meet north_star's bounds outright.

    python scripts/summarize_parity_sweep.py gpurun_out/gs_parity_sweep.json > profiles/<tag>_gs_parity_sweep_summary.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
ran = [c for c in d if "skipped" not in c and "failed" not in c]
out = {"cases": len(d), "passed": len(ran), "skipped": sum("skipped" in c for c in d), "failed": sum("failed" in c for c in d),
       "empty_or_offscreen": sum(c.get("isects", 0) == 0 for c in ran), "modes": {}, "anisotropy": {}}
for c in ran:
    out["modes"][c["mode"]] = out["modes"].get(c["mode"], 0) + 1
    out["anisotropy"][str(c["anisotropy"])] = out["anisotropy"].get(str(c["anisotropy"]), 0) + 1
mx = lambda key: max((c.get(key, 0.0) for c in ran), default=0.0)
out["radii_differing_by_1"] = sum(c.get("radii_differ", 0) for c in ran)
out["visible_total"] = sum(c.get("visible", 0) for c in ran)
out["isects_total"] = sum(c.get("isects", 0) for c in ran)
out["stable_frac_min"] = min((c["stable_frac"] for c in ran if "stable_frac" in c), default=None)
out["projection"] = {"means2d_err_per_100px_max": mx("means2d_err_per_100px"), "depth_rel_err_max": mx("depth_rel_err"),
                     "conic_rel_err_max": mx("conic_rel_err"), "oracle_fp32_conic_rel_err_max": mx("oracle_fp32_conic_rel_err")}
out["image"] = {"err_max": mx("image_err"), "oracle_fp32_err_max": mx("oracle_fp32_image_err"), "alpha_err_max": mx("alpha_err"),
                "oracle_fp32_alpha_err_max": mx("oracle_fp32_alpha_err"),
                "cases_within_1e-4_outright": sum(c.get("image_err", 0) < 1e-4 and c.get("alpha_err", 0) < 1e-4 for c in ran)}
out["absgrad_norm_rel_max"] = mx("absgrad_norm_rel")
g = {}
for c in ran:
    for k, v in c.get("grads", {}).items():
        e = g.setdefault(k, {"cases": 0, "norm_rel_max": 0.0, "oracle_fp32_norm_rel_max": 0.0, "elem_worst_max": 0.0, "oracle_fp32_elem_worst_max": 0.0,
                             "elem_p99_max": 0.0, "oracle_fp32_elem_p99_max": 0.0, "cases_norm_rel_below_1e-3": 0, "cases_elem_worst_below_2e-3": 0,
                             "cases_hip_norm_rel_le_fp32_oracle": 0})
        e["cases"] += 1
        for a, b in (("norm_rel_max", "norm_rel"), ("oracle_fp32_norm_rel_max", "oracle_fp32_norm_rel"), ("elem_worst_max", "elem_worst"),
                     ("oracle_fp32_elem_worst_max", "oracle_fp32_elem_worst"), ("elem_p99_max", "elem_p99"), ("oracle_fp32_elem_p99_max", "oracle_fp32_elem_p99")):
            e[a] = max(e[a], v[b])
        e["cases_norm_rel_below_1e-3"] += v["norm_rel"] < 1e-3
        e["cases_elem_worst_below_2e-3"] += v["elem_worst"] < 2e-3
        e["cases_hip_norm_rel_le_fp32_oracle"] += v["norm_rel"] <= v["oracle_fp32_norm_rel"]
out["gradients"] = g
print(json.dumps(out, indent=1))
