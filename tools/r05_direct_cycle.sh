#!/bin/bash
# one cycle of the DIRECT step's development: its parity tests, then step / kernel times on config 4's per-rank batches
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_runs.py -q -x 2>&1 | tail -30 > $O/direct_tests.txt
grep -q passed $O/direct_tests.txt || exit 0
grep -q failed $O/direct_tests.txt && exit 0
python tools/prep.py > /dev/null 2>&1
B="--no-cpu --no-extra --no-cold --steps 30 --warmup 3"
for tag in slab8 dense; do
  if [ $tag = slab8 ]; then W="--slab-of 8"; else W="--queries 12500000"; fi
  python bench.py $B $W > $O/perf_${tag}_direct.json 2> $O/perf_${tag}_direct.err
  bash tools/kstats_cmd.sh bench.py $B $W > $O/kstats_${tag}_direct.txt 2>&1
done
if [ "$1" = pmc ]; then
  bash tools/pmc_any.sh r05/pmc_direct_dense igd_scan_direct --queries 12500000 > /dev/null 2>&1
  bash tools/pmc_any.sh r05/pmc_direct_slab8 igd_scan_direct --slab-of 8 > /dev/null 2>&1
fi
