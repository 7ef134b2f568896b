#!/bin/bash
# tools/direct_ab.sh -- GPU box: the DIRECT step's two forms side by side on config 4's shapes (one box, same process order):
# IGD_HIP_CHUNKS=1 igd_scan_chunks (query-partitioned, no pre-pass), IGD_HIP_CHUNKS=0 round 5's bounds pass + igd_scan_direct.
# Prints ms per step, the scan kernel's HIP-event time and the oracle check of every run.
for shape in "--slab-of 8" "--slab-of 4" "--slab-of 2" "--queries 12500000"; do
  for c in 1 0 1 0; do
    IGD_HIP_CHUNKS=$c python bench.py --no-cpu --no-extra --no-cold --steps 30 --warmup 5 $shape "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s chunks=$c  step %7.1f us  kernel %-16s %7.1f us  frac %.3f  oracle %s' % ('$shape', d['ms_per_step'] * 1e3, d['roofline']['kernel'], d['roofline']['kernel_ms'] * 1e3, d['roofline']['frac'], d.get('matches_oracle')))"
  done
done
