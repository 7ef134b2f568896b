#!/bin/bash
# DIRECT step vs the ordinary one on config 4's per-rank batches: step times (bench lines) and per-kernel times (rocprofv3)
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
B="--no-cpu --no-extra --no-cold --steps 30 --warmup 3"
for tag in slab8 dense; do
  if [ $tag = slab8 ]; then W="--slab-of 8"; else W="--queries 12500000"; fi
  for ab in direct ordinary; do
    X=""; [ $ab = ordinary ] && X="--long-queries"
    python bench.py $B $W $X > $O/perf_${tag}_${ab}.json 2> $O/perf_${tag}_${ab}.err
    bash tools/kstats_cmd.sh bench.py $B $W $X > $O/kstats_${tag}_${ab}.txt 2>&1
  done
done
