#!/bin/bash
# A/B of library variants: k_query_bounds on the headline batch, slab 0 of 8 and the dense share (rocprofv3 kernel averages)
O=gpurun_out/r05; mkdir -p $O; : > $O/qb_ab.txt
python tools/prep.py > /dev/null 2>&1
for d in igd_amd/lib igd_amd/libv_*; do
IGD_AMD_LIBDIR=$PWD/$d timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_runs.py tests/test_gpu_grouping.py tests/test_gpu_rank.py -q -x 2>&1 | tail -1 >> $O/qb_ab.txt
done
for rep in 1 2; do
for cfg in "headline:" "slab8:--slab-of 8" "dense:--queries 12500000"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  for d in igd_amd/lib igd_amd/libv_*; do
    echo "== $tag $(basename $d) $(bash tools/kstats_lib.sh $d $args 2>&1 | grep -E "k_query_bounds|igd_scan" | sed 's/calls.*avg//' | tr '\n' ' ')" >> $O/qb_ab.txt
  done
done
done
