#!/bin/bash
O=gpurun_out/r05; mkdir -p $O; : > $O/skew_ab.txt
python tools/prep.py > /dev/null 2>&1
for rep in 1 2; do
for d in igd_amd/lib igd_amd/libv_*; do
  echo "== $(basename $d)" >> $O/skew_ab.txt
  IGD_AMD_LIBDIR=$PWD/$d python3 tools/skew_probe.py 2>&1 | grep -v "^W\|^E\|amdgpu" >> $O/skew_ab.txt
done
done
