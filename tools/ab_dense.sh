#!/bin/bash
# tools/ab_dense.sh <out tag> -- GPU box: ab4.sh's dense and slab-of-8 batches, each build three times in turn (box drift shows)
out=gpurun_out/$1; mkdir -p $out
export IGD_HIP_ALLOW_EXP_BUILD=1
python tools/prep.py > /dev/null 2>&1
for rep in 1 2 3; do
for cfg in "dense:--queries 12500000" "slab8:--slab-of 8" "headline:"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  for d in igd_amd/lib igd_amd/libv_*; do
    [ -f $d/libigd_hip.so ] || continue
    echo "== $tag $(basename $d) $(bash tools/kstats_lib.sh $d $args 2>&1 | grep "igd_scan_sorted" | sed 's/.*avg//')" | tee -a $out/ab.txt
  done
done
done
