#!/bin/bash
# instruction mix of the batch's last launch (k_reduce_slabs) for 10^5 queries of 100-200 kbp
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
root=$PWD; out=$root/gpurun_out/pmc_tail; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$tag -- python3 $root/tools/length_one.py 100000 200000 100000 6 > $out/$tag.log 2>&1 || true
done
cd $root
python3 - $out <<'PY' | tee $O/tail_pmc.txt
import csv, glob, sys, collections
out = sys.argv[1]
for pat in ("k_reduce_slabs", "igd_scan_sorted"):
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
