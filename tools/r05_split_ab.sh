#!/bin/bash
# A/B of the grouping kernels on the shuffled batch: staged fine split vs IGD_HIP_SPLIT_NOSTAGE, tile bits in LDS vs IGD_HIP_SPLIT_NOBITS, flags 2 vs auto
O=gpurun_out/r05; mkdir -p $O; : > $O/split_ab.txt
python tools/prep.py > /dev/null 2>&1
timeout 1200 python -m pytest tests/test_gpu_grouping.py tests/test_gpu_skew.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_tilewidths.py tests/test_gpu_long.py -q -x 2>&1 | tail -3 >> $O/split_ab.txt
for rep in 1 2; do
  echo "== bucket lib" >> $O/split_ab.txt
  bash tools/kstats_lib.sh igd_amd/lib --shuffled 2>&1 | grep -E "k_split|igd_scan|k_reduce|k_query" >> $O/split_ab.txt
  echo "== bucket lib FINEB_GRID=96" >> $O/split_ab.txt
  IGD_HIP_FINEB_GRID=96 bash tools/kstats_lib.sh igd_amd/lib --shuffled 2>&1 | grep -E "k_split|igd_scan|k_reduce|k_query" >> $O/split_ab.txt
done
for g in bucket auto; do
  python bench.py --no-cpu --no-extra --no-cold --shuffled --grouping $g --steps 200 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step', '$g', d['ms_per_step']*1e3, d.get('matches_oracle'))" >> $O/split_ab.txt
done
bash tools/r05_sp_stamps.sh
