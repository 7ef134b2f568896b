#!/bin/bash
# A/B of library variants (igd_amd/lib and igd_amd/libv_*) on config 4's per-rank batches: per-kernel times by rocprofv3, 2 rounds
O=gpurun_out/r05; mkdir -p $O; : > $O/ab.txt
python tools/prep.py > /dev/null 2>&1
for rep in 1 2; do
for cfg in "dense:--queries 12500000" "slab8:--slab-of 8"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  for d in igd_amd/lib igd_amd/libv_*; do
    [ -f $d/libigd_hip.so ] || continue
    echo "== $tag $(basename $d) $(bash tools/kstats_lib.sh $d $args 2>&1 | grep -E "igd_scan|k_query_bounds" | sed 's/calls.*avg//' | sed 's/(.*)//' | tr '\n' ' ')" >> $O/ab.txt
  done
done
done
