#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python tools/fuzz_engine.py 120 13000 > $O/fuzz_engine3.txt 2>&1
tail -2 $O/fuzz_engine3.txt
IGD_HOST_MAX_QUERIES=0 timeout 600 python tools/fuzz_gpu.py 40 900 > $O/fuzz_gpu3.txt 2>&1
tail -2 $O/fuzz_gpu3.txt
