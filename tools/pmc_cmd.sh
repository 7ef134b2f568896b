#!/bin/bash
# tools/pmc_cmd.sh <out tag> <kernel name substring> <python script> [args] -- GPU box: counters of one kernel of any python command,
# one rocprofv3 --pmc pass per group (with --kernel-trace only -- the kernel names; never with the sys / runtime / hip / hsa / memory-copy trace domains)
tag0=$1; pat=$2; shift 2
root=$PWD; out=$root/gpurun_out/$tag0; rm -rf $out; mkdir -p $out
python tools/prep.py > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_FLAT" "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum" "TCC_ATOMIC_sum TCC_WRITE_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVES SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$tag -- python3 $root/"$@" > $out/$tag.log 2>&1 || true
done
cd $root
python3 - $out "$pat" <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-40s avg/launch %.5g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
rm -rf $out/*/
