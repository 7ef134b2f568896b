#!/bin/bash
O=gpurun_out/r05; mkdir -p $O; : > $O/shape_ab.txt
python tools/prep.py > /dev/null 2>&1
for rep in 1 2; do
for d in igd_amd/lib igd_amd/libv_*; do
  [ -f $d/libigd_hip.so ] || continue
  IGD_AMD_LIBDIR=$PWD/$d python tools/shape_ab.py 2>/dev/null >> $O/shape_ab.txt
done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_rank.py tests/test_gpu_skew.py tests/test_gpu_limits.py -q -x 2>&1 | tail -3 >> $O/shape_ab.txt
