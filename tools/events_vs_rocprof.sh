#!/bin/bash
# tools/events_vs_rocprof.sh -- GPU box: the scan kernel's time by HIP events (bench line) and by rocprofv3 IN THE SAME PROCESS, then un-profiled twice
python tools/prep.py > /dev/null 2>&1
root=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/evr
rocprofv3 --kernel-trace --output-format csv -d /tmp/evr -- python3 $root/bench.py --no-cpu --no-extra --no-cold > /tmp/evr.json 2>/dev/null
python3 - <<'PY'
import csv, glob, json, statistics
d = json.load(open("/tmp/evr.json"))
t = []
for f in glob.glob("/tmp/evr/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "igd_scan_sorted" in r["Kernel_Name"]: t.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("profiled process : events %.2f us over %d launches | rocprof all %d launches avg %.2f median %.2f, last 200 avg %.2f | step %.1f us" % (
    d["roofline"]["kernel_ms"] * 1e3, d["roofline"]["launches_timed"], len(t), sum(t) / len(t), statistics.median(t), sum(t[-200:]) / 200, d["ms_per_step"] * 1e3))
PY
cd $root
for i in 1 2; do python3 bench.py --no-cpu --no-extra --no-cold | python3 -c "import json,sys; d=json.load(sys.stdin); print('un-profiled run   : events %.2f us | step %.1f us' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3))"; done
