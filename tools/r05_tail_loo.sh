#!/bin/bash
# leave-one-out builds of the last launch (WRONG counts): 10^5 queries of 100-200 kbp without the coverage sums / without the exact walks
O=gpurun_out/r05; mkdir -p $O; : > $O/tail_loo.txt
python tools/prep.py > /dev/null 2>&1
for d in igd_amd/lib igd_amd/libv_nocov igd_amd/libv_nowalk; do
  echo "== $(basename $d)" >> $O/tail_loo.txt
  root=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
  IGD_HIP_ALLOW_EXP_BUILD=1 IGD_AMD_LIBDIR=$root/$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $root/tools/length_one.py 100000 200000 100000 30 2>&1 | grep "^len" >> $root/$O/tail_loo.txt
  python3 - >> $root/$O/tail_loo.txt <<'PY'
import csv, glob
for f in glob.glob("/tmp/ks/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 30 and "aos" not in r["Name"]: print("   %-60s avg %8.1f us" % (r["Name"][:60], float(r["AverageNs"]) / 1e3))
PY
  cd $root
done
