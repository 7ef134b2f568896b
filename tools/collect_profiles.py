#!/usr/bin/env python3
"""tools/collect_profiles.py <round>  -- copy what tools/profile_all.sh left under gpurun_out/ into profiles/<round>/ (tracked)
and rebuild profiles/traffic.json (PMC-measured HBM bytes per launch of the dominant kernel, keyed by workload the way
bench.py looks it up, each entry tagged with the profile it came from)."""
import json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
dst = os.path.join(root, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
keys = {"sorted": "hits/sorted/q1000000", "shuffled": "hits/shuffled/q1000000", "v500": "v/sorted/q1000000",
        "dense": "hits/sorted/q12500000", "slab8": "hits/sorted/q12500000/slab0of8", "exact": "hits/sorted/q1000000/exact", "auto": None}
try:
    traffic = json.load(open(os.path.join(root, "profiles", "traffic.json")))      # entries of earlier rounds stay until re-measured
except Exception:
    traffic = {}
for tag, key in keys.items():
    src = os.path.join(root, "gpurun_out", "profile_" + tag)
    if not os.path.isdir(src):
        continue
    d = os.path.join(dst, tag)
    os.makedirs(d, exist_ok=True)
    for f in ("bench.json", "kernel_stats.csv", "pmc_fetch.csv", "pmc_write.csv", "traffic.json"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(d, f))
    try:
        t = json.load(open(os.path.join(src, "traffic.json")))
        if key and t.get("hbm_bytes_per_launch"):
            t["source"] = "profiles/%s/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % (rnd, tag)
            traffic[key] = t
    except Exception:
        pass
for tag in ("sorted", "dense", "slab8"):                  # instruction mix of the scan kernel / of k_query_bounds (tools/pmc_any.sh)
    for kern in ("scan", "qb", "direct"):
        src = os.path.join(root, "gpurun_out", "pmc_%s_%s" % (kern, tag), "summary.txt")
        if os.path.exists(src):
            os.makedirs(os.path.join(dst, "pmc"), exist_ok=True)
            shutil.copy(src, os.path.join(dst, "pmc", "instruction_mix_%s_%s.txt" % (kern, tag)))
for f in ("misc.json",):
    if os.path.exists(os.path.join(root, "gpurun_out", f)):
        shutil.copy(os.path.join(root, "gpurun_out", f), os.path.join(dst, f))
if traffic:
    json.dump(traffic, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print("collected", sorted(os.listdir(dst)), "traffic keys", sorted(traffic))
