#!/bin/bash
# A/B of the grouping kernels on the shuffled batch (kernel times by rocprofv3, step by bench.py): library variants
O=gpurun_out/r05; mkdir -p $O; : > $O/split_ab.txt
python tools/prep.py > /dev/null 2>&1
for d in igd_amd/lib igd_amd/libv_*; do
IGD_AMD_LIBDIR=$PWD/$d timeout 1200 python -m pytest tests/test_gpu_grouping.py tests/test_gpu_skew.py -q -x 2>&1 | tail -1 >> $O/split_ab.txt
done
for rep in 1 2; do
  for d in igd_amd/lib igd_amd/libv_*; do
    echo "== bucket $d" >> $O/split_ab.txt
    bash tools/kstats_lib.sh $d --shuffled 2>&1 | grep -E "k_split|igd_scan|k_reduce|k_query" >> $O/split_ab.txt
  done
done
for d in igd_amd/lib igd_amd/libv_*; do
  IGD_AMD_LIBDIR=$PWD/$d python bench.py --no-cpu --no-extra --no-cold --shuffled --steps 200 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step', '$d', d['ms_per_step']*1e3, d.get('matches_oracle'))" >> $O/split_ab.txt
done
