#!/bin/bash
# the round's last GPU call: full GPU suite, kernel profiles of the five step shapes on the final build, the bench line
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > gpurun_out/prep.log 2>&1
(time python -m pytest tests -m gpu -q) > $O/tests_final.txt 2>&1
bash tools/profile.sh sorted > gpurun_out/p_sorted.log 2>&1
bash tools/profile.sh v500 --v 500 > gpurun_out/p_v500.log 2>&1
bash tools/profile.sh shuffled --shuffled > gpurun_out/p_shuffled.log 2>&1
bash tools/profile.sh dense --queries 12500000 --steps 10 --warmup 2 --no-cpu > gpurun_out/p_dense.log 2>&1
bash tools/profile.sh slab8 --slab-of 8 --steps 10 --warmup 2 --no-cpu > gpurun_out/p_slab8.log 2>&1
bash tools/r05_bench.sh
