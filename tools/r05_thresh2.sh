#!/bin/bash
# where the DIRECT step starts to pay: unsharded sorted batches of 30 .. 66 queries per tile, both steps
O=gpurun_out/r05; mkdir -p $O; : > $O/thresh2.txt
python tools/prep.py > /dev/null 2>&1
B="--no-cpu --no-extra --no-cold --steps 30 --warmup 3"
for nq in 5700000 7500000 9400000; do
  for ab in direct ordinary; do
    X=""; [ $ab = ordinary ] && X="--long-queries"
    python bench.py $B --queries $nq $X 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('nq $nq $ab step %.1f us kernel %s %.1f us' % (j['ms_per_step']*1e3, r['kernel'], r['kernel_ms']*1e3))" >> $O/thresh2.txt
  done
done
