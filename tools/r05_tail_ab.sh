#!/bin/bash
# A/B of library variants: the batch's last launch at other occupancies -- long queries' step, and the headline step it is part of
O=gpurun_out/r05; mkdir -p $O; : > $O/tail_ab.txt
python tools/prep.py > /dev/null 2>&1
for rep in 1 2; do
for d in igd_amd/lib igd_amd/libv_*; do
  [ -f $d/libigd_hip.so ] || continue
  echo "== $(basename $d)" >> $O/tail_ab.txt
  IGD_AMD_LIBDIR=$PWD/$d python3 tools/length_one.py 100000 200000 100000 30 2>&1 | grep "^len" >> $O/tail_ab.txt
  IGD_AMD_LIBDIR=$PWD/$d python3 tools/length_one.py 100000 200000 1000000 10 2>&1 | grep "^len" >> $O/tail_ab.txt
  IGD_AMD_LIBDIR=$PWD/$d python bench.py --no-cpu --no-extra --no-cold --steps 200 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('headline step', d['ms_per_step']*1e3, d.get('matches_oracle'))" >> $O/tail_ab.txt
done
done
IGD_AMD_LIBDIR=$PWD/igd_amd/libv_o5w640 timeout 900 python -m pytest tests/test_gpu_long.py tests/test_gpu_stress.py tests/test_gpu_skew.py -q -x 2>&1 | tail -2 >> $O/tail_ab.txt
