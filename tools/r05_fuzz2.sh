#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python tools/fuzz_engine.py 200 9000 > $O/fuzz_engine2.txt 2>&1
tail -2 $O/fuzz_engine2.txt
IGD_HOST_MAX_QUERIES=0 timeout 900 python tools/fuzz_gpu.py 60 400 > $O/fuzz_gpu.txt 2>&1
tail -2 $O/fuzz_gpu.txt
