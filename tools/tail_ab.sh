#!/bin/bash
# tools/tail_ab.sh <out tag> -- GPU box: the batch's last launch (k_reduce_slabs) in every library build: headline batch and two long-query batches
out=gpurun_out/$1; mkdir -p $out
for d in igd_amd/lib igd_amd/libv_*; do
  [ -f $d/libigd_hip.so ] || continue
  for a in "100 1999 1000000" "20000 60000 1000000" "100000 200000 100000"; do
    echo "== $(basename $d) $a" | tee -a $out/tail_ab.txt
    IGD_AMD_LIBDIR=$PWD/$d bash tools/kstats_cmd.sh tools/length_one.py $a 2>&1 | grep "k_reduce\|k_query\|igd_scan" | tee -a $out/tail_ab.txt
  done
done
