#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_create.py tests/test_gpu_seqpare.py -q -x 2>&1 | tail -2 > $O/create_tests.txt
bash tools/create_bench.sh > /dev/null 2>&1
cp gpurun_out/create_bench.txt $O/create_bench.txt
