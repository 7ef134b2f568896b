#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_long.py tests/test_gpu_stress.py tests/test_gpu_direct.py tests/test_gpu_grouping.py -q -x 2>&1 | tail -3 > $O/long_probe.txt
root=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $root/tools/length_one.py 100000 200000 100000 30 2>&1 | grep "^len" >> $root/$O/long_probe.txt
python3 - >> $root/$O/long_probe.txt <<'PY'
import csv, glob
for f in glob.glob("/tmp/ks/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 30: print("%-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cd $root
python3 tools/length_one.py 100000 200000 1000000 10 2>&1 | grep "^len" >> $O/long_probe.txt
python3 tools/length_one.py 20000 60000 1000000 10 2>&1 | grep "^len" >> $O/long_probe.txt
