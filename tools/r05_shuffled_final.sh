#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
bash tools/pmc_few.sh pmc_split k_split_local,k_split_fine_a,igd_scan_tiles --shuffled > /dev/null 2>&1
bash tools/profile.sh shuffled --shuffled > gpurun_out/p_shuffled.log 2>&1
(time python -m pytest tests -m gpu -q -x) > $O/tests_split.txt 2>&1
tail -3 $O/tests_split.txt
