#!/bin/bash
# tools/kstats_lib.sh <libdir> [bench args] -- tools/kstats.sh for one library directory (variant builds)
lib=$1; shift
python tools/prep.py > /dev/null 2>&1
root=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
IGD_HIP_ALLOW_EXP_BUILD=1 IGD_AMD_LIBDIR=$root/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $root/bench.py --no-cpu --no-extra --no-cold --steps 20 --warmup 3 "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/ks/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 20: print("%-44s calls %4s avg %8.1f us" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
