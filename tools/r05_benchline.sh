#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > gpurun_out/prep.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_bench_line.py -q -x 2>&1 | tail -2 > $O/benchline_test.txt
bash tools/r05_bench.sh
