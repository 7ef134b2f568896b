#!/bin/bash
# the round's last GPU call: engine fuzz on the final build, profiles of the headline kernels, the bench line
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > gpurun_out/prep.log 2>&1
timeout 1200 python tools/fuzz_engine.py 100 11000 > $O/fuzz_engine3.txt 2>&1
bash tools/profile.sh sorted > gpurun_out/p_sorted.log 2>&1
bash tools/profile.sh v500 --v 500 > gpurun_out/p_v500.log 2>&1
bash tools/profile.sh shuffled --shuffled > gpurun_out/p_shuffled.log 2>&1
bash tools/pmc_any.sh pmc_scan_sorted igd_scan_sorted > /dev/null 2>&1
bash tools/r05_bench.sh
