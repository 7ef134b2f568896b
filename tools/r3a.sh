#!/bin/bash
out=gpurun_out/r3a; mkdir -p $out
python tools/prep.py > /dev/null 2>&1
for cfg in "default:" "dense:--queries 12500000" "slab8:--slab-of 8"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  echo "== $tag" >> $out/kstats.txt
  bash tools/kstats.sh --no-extra $args >> $out/kstats.txt 2>&1
done
for cfg in "dense:--queries 12500000" "slab8:--slab-of 8"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  IGD_AMD_LIBDIR=$PWD/igd_amd/libv_sect python bench.py --no-cpu --no-extra --steps 30 --warmup 3 $args > $out/sect_$tag.json 2> $out/sect_$tag.err
done
python bench.py --steps 200 --warmup 10 > $out/bench_default.json 2> $out/bench_default.err
