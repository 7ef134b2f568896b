#!/bin/bash
# tools/ab4.sh <out tag> -- GPU box: per-kernel times (rocprofv3 --stats) of every library build (igd_amd/lib, igd_amd/libv_*) on the SAME box
# for the three batches that matter: headline (10^6), config 4's per-GPU share (1.25e7) and slab 0 of 8
out=gpurun_out/$1; mkdir -p $out
python tools/prep.py > /dev/null 2>&1
for cfg in "headline:" "dense:--queries 12500000" "slab8:--slab-of 8" "small:--queries 100000"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  for d in igd_amd/lib igd_amd/libv_*; do
    [ -f $d/libigd_hip.so ] || continue
    echo "== $tag $(basename $d)" | tee -a $out/ab.txt
    bash tools/kstats_lib.sh $d $args 2>&1 | grep -v "copyBuffer\|aos_to_soa\|k_pack\|fill\|elementwise" | tee -a $out/ab.txt
  done
done
