#!/bin/bash
O=gpurun_out/r05; mkdir -p $O; : > $O/layout.txt
python tools/prep.py > /dev/null 2>&1
B="--no-cpu --no-extra --no-cold --steps 30 --warmup 3"
for cfg in "dense:--queries 12500000" "slab8:--slab-of 8"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  for lay in ichr runs; do
   for w in 0 1; do
    if [ $w = 1 ]; then export IGD_HIP_BONLY_WIDE=1; else unset IGD_HIP_BONLY_WIDE; fi
    echo "== $tag $lay wide=$w $(bash tools/kstats_cmd.sh bench.py $B $args --query-layout $lay 2>&1 | grep -E "igd_scan|k_query_bounds|k_reduce" | sed 's/calls.*avg//' | tr '\n' ' ')" >> $O/layout.txt
   done
  done
done
