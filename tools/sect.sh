#!/bin/bash
# tools/sect.sh <out tag> -- GPU box: section timers of the rank method (igd_amd/libv_sect: -DIGD_EXP=1024) on the dense share and slab 0 of 8
out=gpurun_out/$1; mkdir -p $out
python tools/prep.py > /dev/null 2>&1
for cfg in "dense:--queries 12500000" "slab8:--slab-of 8"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  IGD_AMD_LIBDIR=$PWD/igd_amd/libv_sect python bench.py --no-cpu --no-extra --steps 30 --warmup 3 $args > $out/sect_$tag.json 2> $out/sect_$tag.err
  echo "== $tag"; grep "igd sect" $out/sect_$tag.err
done
