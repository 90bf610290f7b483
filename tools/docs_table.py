#!/usr/bin/env python3
"""The configuration table of DESIGN.md section 4 / README.md out of a bench line (so that the documents quote the committed file):
    python tools/docs_table.py profiles/r6_bench_default.json"""
import json
import sys

b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o = b["other_workloads"]
cb = b["cpu_baseline"]
seeds = b["config"]["seeds"]["ms_per_step_by_seed"]
sw = b["scaling_workload"]


def frac(r):
    return r["frac"] if "bound" in r else max(v["frac"] for v in r.values())


rows = [("C2' 9298 x 256, k = 4 (`value`; seeds 0 / 1 / 2: %s ms)" % " / ".join("%.3f" % seeds[k] for k in sorted(seeds)),
         "**%.3f ms**, %d launches, GPU busy %.2f" % (b["ms_per_step"], b["library_launches_per_step"], (b.get("round_gaps") or {}).get("gpu_busy_frac", float("nan"))),
         "**%.1f M**" % (b["value"] / 1e6), "`qmc_main<4>` %.3f FP64 VALU, issue %.2f" % (b["roofline"]["frac"], b["roofline"].get("valu_issue_frac") or float("nan")),
         "%.0f" % cb["value"], "%.0fx" % b["speedup_vs_cpu_baseline"])]
for label, name, kern in (("C3' 25 000 x 512, k = 8", "ital_k8_25000x512", "`qmc_main<8>`"), ("C4' 50 000 x 2048, k = 8", "ital_k8_50000x2048", "`qmc_main<8>`"),
                          ("C5' share 125 000 x 512, k = 16, mc = 1", "ital_k16_mc1_125000x512", "`gen_main<3..16>` (whole steps)"),
                          ("C5' whole 1 000 000 x 512, k = 16, mc = 1, ONE GPU", "ital_k16_mc1_1Mx512", "`gen_main<3..16>`"),
                          ("noisy user 9298 x 256, k = 4 (lp 0.5, mp 0.25)", "ital_general_user_k4", "`gen_main<4>`")):
    w = o[name]
    ms = w["ms_per_round"]
    rows.append((label, "%.3f s" % (ms / 1e3) if ms > 1000 else "%.1f ms" % ms, "%.0f" % w["candidates_per_s"], "%s %.3f" % (kern, frac(w["roofline"])),
                 "%.1f" % w["cpu_baseline"]["value"], "%.0fx" % w["speedup_vs_cpu_baseline"]))
rows.append(("C5' 1 000 000 x 512, k = 4 (the scaling curve's workload, N = 1)", "%.1f ms (fetch %.1f, update %.1f; host %.2f ms between rounds)" % (
    sw["ms_per_round"], sw["fetch_ms_per_round"], sw["update_ms_per_round"], sw["host_ms_per_round"][0]), "%.1f M" % (sw["candidates_per_s"] / 1e6),
    "t = 4: %.1f ms" % [v for k, v in sw["kernel_ms"].items() if k.endswith("_t4")][0], "--", "--"))
print("| configuration (BASELINE.json) | round (fetch + update) | scored candidates/s | dominant kernel vs bound | CPU baseline | ratio |")
print("|---|---|---|---|---|---|")
for r in rows:
    print("| " + " | ".join(r) + " |")
print()
print("hbm: %.3f ms, %.2f TB/s, frac %.3f, traffic %.4f GB" % (b["roofline_hbm"]["avg_launch_ms"], b["roofline_hbm"]["achieved"] / 1e3, b["roofline_hbm"]["frac"], (b["roofline_hbm"].get("traffic") or 0) / 1e9))
print("cov_block: %.2f ms, %.1f TF, frac %.3f" % (o["cov_block_20000x512"]["ms_per_launch"], o["cov_block_20000x512"]["roofline"]["achieved"], o["cov_block_20000x512"]["roofline"]["frac"]))
m = o["mcmi_min_subsample1000_k4"]
print("mcmi: %.2f ms/round k4, %.2f k6; score<4> frac %.3f issue %s" % (m["ms_per_round"], o["mcmi_min_subsample1000_k6"]["ms_per_round"],
      m["roofline"]["mcmi_score_kernel<4>"]["frac"], m["roofline"]["mcmi_score_kernel<4>"].get("valu_issue_frac")))
print("ce subset: %.1f ms (150x4), %.1f ms (9298x256)" % (o["ital_ce_subset5_iris_shaped_150x4"]["ms_per_round"], o["ital_ce_subset5_9298x256"]["ms_per_round"]))
print("qmc4: %.3f ms frac %.3f issue %s traffic %s" % (b["roofline"]["avg_launch_ms"], b["roofline"]["frac"], b["roofline"].get("valu_issue_frac"), b["roofline"].get("traffic")))
print("fit_ms", b["fit_ms"]["construct_ms"], b["fit_ms"]["first_update_ms"], "wall", b["summary"].get("bench_wall_s"))
print("cpu samples", cb.get("samples_candidates_per_s"), cb["cores"])
