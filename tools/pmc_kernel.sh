#!/bin/bash
# tools/pmc_kernel.sh <kernel name substring> [bench args] -- instruction counters of one kernel (GPU box), one rocprofv3 --pmc pass per pair
pat=$1; shift
root=$PWD; out=$root/gpurun_out/pmc_kernel; rm -rf $out; mkdir -p $out
python tools/prep.py > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$tag -- python3 $root/bench.py --no-cpu --no-extra --no-cold --steps 6 --warmup 2 "$@" > $out/$tag.log 2>&1 || true
done
cd $root
python3 - $out "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-24s avg/launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
