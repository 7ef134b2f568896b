#!/bin/bash
# tools/profile_all.sh -- run ON THE GPU BOX (via gpurun): refreshes every summary kept under profiles/rNN/ (tools/collect_profiles.py copies them there)
python tools/prep.py > gpurun_out/prep.log 2>&1
bash tools/profile.sh sorted > gpurun_out/p_sorted.log 2>&1
bash tools/profile.sh auto --grouping auto > gpurun_out/p_auto.log 2>&1
bash tools/profile.sh shuffled --shuffled > gpurun_out/p_shuffled.log 2>&1
bash tools/profile.sh v500 --v 500 > gpurun_out/p_v500.log 2>&1
bash tools/profile.sh exact --exact-arrays > gpurun_out/p_exact.log 2>&1
bash tools/profile.sh dense --queries 12500000 --steps 10 --warmup 2 --no-cpu > gpurun_out/p_dense.log 2>&1
bash tools/profile.sh slab8 --slab-of 8 --steps 10 --warmup 2 --no-cpu > gpurun_out/p_slab8.log 2>&1
bash tools/pmc_any.sh pmc_scan_sorted igd_scan_sorted > /dev/null 2>&1
rm -rf gpurun_out/pmc_direct_dense gpurun_out/pmc_direct_slab8 gpurun_out/pmc_qb_slab8      # (never a stale summary of an earlier round)
bash tools/pmc_any.sh pmc_direct_dense igd_scan_direct --queries 12500000 > /dev/null 2>&1
bash tools/pmc_any.sh pmc_direct_slab8 igd_scan_direct --slab-of 8 > /dev/null 2>&1
bash tools/pmc_any.sh pmc_qb_slab8 k_query_bounds --slab-of 8 > /dev/null 2>&1
bash tools/pmc_any.sh pmc_qb_dense k_query_bounds --queries 12500000 > /dev/null 2>&1
python tools/measure_misc.py > gpurun_out/misc.json 2> gpurun_out/misc.err
bash tools/profile_create.sh > gpurun_out/p_create.log 2>&1
bash tools/seqpare_bench.sh 10000 100000 300000 1000000 > /dev/null 2>&1
ls gpurun_out
