#!/bin/bash
# tools/kstats_cmd.sh <python script> [args] -- GPU box: per-kernel average times (rocprofv3 --kernel-trace --stats) of one python command
python tools/prep.py > /dev/null 2>&1
root=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $root/"$@" 2>&1 | grep -v amdgpu.ids | tail -2
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/ks/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 10 and float(r["AverageNs"]) > 2000: print("%-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
