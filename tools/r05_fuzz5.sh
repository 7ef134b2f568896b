#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 1000 python tools/fuzz_engine.py 80 17000 > $O/fuzz_engine5.txt 2>&1
tail -2 $O/fuzz_engine5.txt
