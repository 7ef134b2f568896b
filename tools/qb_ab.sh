#!/bin/bash
# tools/qb_ab.sh <out tag> -- GPU box: k_query_bounds' time (rocprofv3 --stats) in every library build (IGD_EXP section variants give wrong counts)
out=gpurun_out/$1; mkdir -p $out
export IGD_HIP_ALLOW_EXP_BUILD=1
python tools/prep.py > /dev/null 2>&1
for cfg in "headline:" "dense:--queries 12500000" "slab8:--slab-of 8"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  for d in igd_amd/lib igd_amd/libv_*; do
    [ -f $d/libigd_hip.so ] || continue
    echo "== $tag $(basename $d)" | tee -a $out/ab.txt
    bash tools/kstats_lib.sh $d $args 2>&1 | grep "k_query_bounds\|igd_scan_sorted\|k_reduce" | tee -a $out/ab.txt
  done
done
