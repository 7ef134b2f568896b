#!/bin/bash
# tools/pmc_few.sh <out tag> <kernel patterns, comma separated> [bench args] -- instruction mix of several kernels from the same few rocprofv3 --pmc passes
tag0=$1; pats=$2; shift 2
root=$PWD; out=$root/gpurun_out/$tag0; rm -rf $out; mkdir -p $out
python tools/prep.py > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$tag -- python3 $root/bench.py --no-cpu --no-extra --no-cold --steps 6 --warmup 2 "$@" > $out/$tag.log 2>&1 || true
done
cd $root
python3 - $out "$pats" <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
out, pats = sys.argv[1], sys.argv[2].split(",")
for pat in pats:
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", pat)
    for k in sorted(acc):
        v = acc[k]
        print("%-28s avg/launch %.5g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
rm -rf $out/*/
