#!/bin/bash
# tools/pmc_scan.sh -- instruction mix of igd_scan_tiles (run on the GPU box): one rocprofv3 --pmc pass per group
root=$PWD; out=$root/gpurun_out/pmc_scan; rm -rf $out; mkdir -p $out   # (a fresh directory: the summary below averages every file it finds)
python tools/prep.py > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$tag -- python3 $root/bench.py --no-cpu --no-extra --no-cold --steps 10 --warmup 2 "$@" > $out/$tag.log 2>&1 || true
done
cd $root
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "igd_scan_sorted" in r["Kernel_Name"] or "igd_scan_tiles<true" in r["Kernel_Name"] or ("igd_scan_tiles<false" in r["Kernel_Name"] and "--shuffled" in sys.argv):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-24s avg/launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
