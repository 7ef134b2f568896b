#!/bin/bash
# A/B of library variants on the headline batch (and -v 500, 10^5): scan kernel by rocprofv3, three rounds
O=gpurun_out/r05; mkdir -p $O; : > $O/ab_head.txt
python tools/prep.py > /dev/null 2>&1
for rep in 1 2 3; do
for cfg in "headline:" "v500:--v 500" "q1e5:--queries 100000"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  for d in igd_amd/lib igd_amd/libv_*; do
    [ -f $d/libigd_hip.so ] || continue
    echo "== $tag $(basename $d) $(bash tools/kstats_lib.sh $d $args 2>&1 | grep -E "igd_scan" | sed 's/calls.*avg//' | tr '\n' ' ')" >> $O/ab_head.txt
  done
done
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -x 2>&1 | tail -3 >> $O/ab_head.txt
