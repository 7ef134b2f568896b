#!/bin/bash
# quick kernel A/B on the GPU box: tools/kb.sh <libdir-suffix> <grouping> [extra bench args]
v=$1; g=$2; shift 2
IGD_AMD_LIBDIR=$PWD/igd_amd/lib$v python bench.py --steps 30 --warmup 5 --no-cpu --grouping $g "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s %-7s %.3g q/s  step %.3f ms  scan %.4f ms  pipe %.4f ms  frac %.3f  total %s' % ('$v' or 'base', d['config']['grouping'], d['value'], d['ms_per_step'], r['kernel_ms'], r['pipeline_ms'], r['frac'], d['hits_per_step_total']))"
