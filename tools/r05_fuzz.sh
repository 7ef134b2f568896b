#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python tools/fuzz_engine.py ${1:-60} ${2:-5000} > $O/fuzz_engine.txt 2>&1
tail -3 $O/fuzz_engine.txt
