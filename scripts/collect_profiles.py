#!/usr/bin/env python3
"""Copies what scripts/r02_final.sh left under gpurun_out/<tag>/ into profiles/ and rebuilds profiles/traffic.json from
the FETCH_SIZE / WRITE_SIZE passes (per kernel, per launch; bench.py reads the walk's figure for its roofline object).
usage: collect_profiles.py [tag]      (here, after the gpurun call has merged gpurun_out/)"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for f in ("bench_line.json", "bench_line_under_rocprof.json", "kernel_stats.csv", "pmc_traffic.txt", "pmc_sq.txt", "pmc_tcp.txt"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (tag, f)))
with open(os.path.join(dst, tag + "_other_workloads.txt"), "w") as out:
    for f in ("configs", "robust", "tiny", "big_stages"):
        out.write("".join(open(os.path.join(src, f + ".txt")).readlines()[-4:]))

vals, cur = {}, None
for line in open(os.path.join(dst, tag + "_pmc_traffic.txt")):
    if line.strip() and not line.startswith(" "):
        cur = line.strip()
    else:
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)", line)
        if m:
            vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
walk = next(v for k, v in vals.items() if "k_spec_both" in k)
total = sum(sum(v.values()) for v in vals.values()) * 1024
path = os.path.join(dst, "traffic.json")
j = json.load(open(path))
b_alg = json.loads(open(os.path.join(dst, tag + "_bench_line.json")).read().strip().splitlines()[-1])["roofline"]["algorithmic_bytes"]
j.update({"fetch_size_kb_per_launch": walk["FETCH_SIZE"], "write_size_kb_per_launch": walk["WRITE_SIZE"],
          "walk_kernel_hbm_bytes": int((walk["FETCH_SIZE"] + walk["WRITE_SIZE"]) * 1024),
          "pipeline_hbm_bytes_per_batch": int(total), "algorithmic_bytes": b_alg,
          "walk_over_b_alg": round((walk["FETCH_SIZE"] + walk["WRITE_SIZE"]) * 1024 / b_alg, 3),
          "pipeline_over_b_alg": round(total / b_alg, 3),
          "per_kernel_kb": {k.replace("void ", "").split("(")[0]: {"fetch": v.get("FETCH_SIZE"), "write": v.get("WRITE_SIZE")}
                            for k, v in vals.items()}})
json.dump(j, open(path, "w"), indent=1)
print("walk %.1f MB = %.2f x B_alg, pipeline %.1f MB = %.2f x" % (j["walk_kernel_hbm_bytes"] / 1e6, j["walk_over_b_alg"],
                                                                 total / 1e6, j["pipeline_over_b_alg"]))
