#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
IGD_HOST_MAX_QUERIES=0 timeout 700 python tools/fuzz_gpu.py 80 2100 > $O/fuzz_gpu6.txt 2>&1
tail -2 $O/fuzz_gpu6.txt
timeout 500 python tools/fuzz_gpu.py 40 3100 > $O/fuzz_gpu6b.txt 2>&1
tail -2 $O/fuzz_gpu6b.txt
