#!/bin/bash
# tools/ab_step.sh [bench args] -- GPU box: ms_per_step of bench.py (no profiler) for every library build igd_amd/lib, igd_amd/libv_* on
# the SAME box, three runs each, alternating
python tools/prep.py > /dev/null 2>&1
for rep in 1 2 3; do
  for d in igd_amd/lib igd_amd/libv_*; do
    [ -f $d/libigd_hip.so ] || continue
    IGD_AMD_LIBDIR=$PWD/$d python bench.py --no-cpu --no-extra --no-cold "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-24s step %.2f us  kernel %.2f us' % ('$(basename $d)', 1e3*d['ms_per_step'], 1e3*d['roofline']['kernel_ms']))"
  done
done
