#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_runs.py -q -x 2>&1 | tail -60 > $O/direct_tests.txt
