#!/bin/bash
# the round's last GPU call (second half of the round): full GPU suite, profiles of the shuffled step, the bench line
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > gpurun_out/prep.log 2>&1
(time python -m pytest tests -m gpu -q) > $O/tests_final.txt 2>&1
bash tools/profile.sh shuffled --shuffled > gpurun_out/p_shuffled.log 2>&1
bash tools/profile.sh sorted > gpurun_out/p_sorted.log 2>&1
bash tools/r05_bench.sh
