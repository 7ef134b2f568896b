#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_grouping.py -q -x -k "every_form" 2>&1 | tail -3 > $O/long_probe.txt
root=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $root/tools/length_one.py 100000 200000 100000 30 >> $root/$O/long_probe.txt 2>&1
python3 - >> $root/$O/long_probe.txt <<'PY'
import csv, glob
for f in glob.glob("/tmp/ks/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 30: print("%-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
