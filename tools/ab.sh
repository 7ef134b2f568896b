#!/bin/bash
# tools/ab.sh <out tag> [bench args] -- GPU box: per-kernel times (rocprofv3 --stats) of every library build igd_amd/lib, igd_amd/libv_*
# on the SAME box (A/B runs: box-to-box spread is larger than most single changes)
out=gpurun_out/$1; shift; mkdir -p $out
python tools/prep.py > /dev/null 2>&1
for d in igd_amd/lib igd_amd/libv_*; do
  [ -f $d/libigd_hip.so ] || continue
  echo "== $(basename $d) $*" | tee -a $out/ab.txt
  bash tools/kstats_lib.sh $d "$@" 2>&1 | grep -v "copyBuffer\|aos_to_soa" | tee -a $out/ab.txt
done
