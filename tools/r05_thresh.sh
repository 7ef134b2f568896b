#!/bin/bash
O=gpurun_out/r05; mkdir -p $O; : > $O/thresh.txt
python tools/prep.py > /dev/null 2>&1
B="--no-cpu --no-extra --no-cold --steps 30 --warmup 3"
for cfg in "slab2:--slab-of 2" "slab4:--slab-of 4" "slab8:--slab-of 8" "dense:--queries 12500000"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  for ab in direct ordinary; do
    X=""; [ $ab = ordinary ] && X="--long-queries"
    python bench.py $B $args $X 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$tag $ab step %.1f us kernel %s %.1f us matches %s' % (j['ms_per_step']*1e3, r['kernel'], r['kernel_ms']*1e3, j.get('matches_oracle')))" >> $O/thresh.txt
  done
done
