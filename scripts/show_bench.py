import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "phases", d.get("phases_s"))
print("svd_stats", d["svd_stats"])
for e in d.get("untimed_sweeps", []): print("  ", e)
print({k: d.get(k) for k in ("sv_max_rel_err", "sv_max_rel_err_individual", "svd_isometry_defect", "matvec_max_rel_err", "E0_rel_err", "energy_err")})
print("svd ms/call", d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], "frac", d["roofline"]["frac"], "gemm frac", d["roofline_gemm"]["frac"])
