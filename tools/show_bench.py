#!/usr/bin/env python3
"""Prints the figures of a bench.py JSON line that the round's notes quote.  usage: show_bench.py <file with the line>"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f MB/s  %.3f ms/step  frac %.4f  launch %.4f ms  traffic %s corrected %s" % (
    d["value"], d["ms_per_step"], r["frac"], r["kernel_ms_avg"], r.get("traffic"), r.get("traffic_corrected")))
for k in ("headline_host_paths", "merge_loop_only", "oov_sensitivity", "configs_1_small_batch"):
    if k in d:
        print(k, {a: b for a, b in d[k].items() if a not in ("what", "workload")})
if "configs_2_roofline_run" in d:
    print("configs_2", d["configs_2_roofline_run"]["timings"]["kernels_ms"], d["configs_2_roofline_run"]["timings"]["device_e2e_ms"],
          d["configs_2_roofline_run"]["timings"]["python_e2e_ms"], d["configs_2_roofline_run"]["roofline"]["frac"],
          "unpadded", d["configs_2_roofline_run"].get("unpadded_run", {}).get("kernel_ms_avg"))
if "configs_4_long_docs" in d:
    print("configs_4", d["configs_4_long_docs"]["kernel_ms_avg"], d["configs_4_long_docs"]["roofline"]["frac"],
          "unpadded", d["configs_4_long_docs"].get("unpadded_run", {}).get("kernel_ms_avg"))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("all_cores", {}).get("value"))
if "next_rows" in d:
    print("next", d["next_rows"])
