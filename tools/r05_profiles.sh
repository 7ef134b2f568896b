#!/bin/bash
# round 5: every summary kept under profiles/r05/ (tools/collect_profiles.py r05 copies them there)
python tools/prep.py > gpurun_out/prep.log 2>&1
bash tools/profile.sh sorted > gpurun_out/p_sorted.log 2>&1
bash tools/profile.sh v500 --v 500 > gpurun_out/p_v500.log 2>&1
bash tools/profile.sh shuffled --shuffled > gpurun_out/p_shuffled.log 2>&1
bash tools/profile.sh dense --queries 12500000 --steps 10 --warmup 2 --no-cpu > gpurun_out/p_dense.log 2>&1
bash tools/profile.sh slab8 --slab-of 8 --steps 10 --warmup 2 --no-cpu > gpurun_out/p_slab8.log 2>&1
bash tools/pmc_any.sh pmc_scan_sorted igd_scan_sorted > /dev/null 2>&1
bash tools/pmc_any.sh pmc_direct_dense igd_scan_direct --queries 12500000 > /dev/null 2>&1
bash tools/pmc_any.sh pmc_direct_slab8 igd_scan_direct --slab-of 8 > /dev/null 2>&1
bash tools/pmc_any.sh pmc_qb_slab8 k_query_bounds --slab-of 8 > /dev/null 2>&1
ls gpurun_out
