#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
IGD_FUZZ_B=14,15 IGD_FUZZ_BUILDS= timeout 800 python tools/fuzz_engine.py 60 15000 > $O/fuzz_engine4.txt 2>&1
tail -2 $O/fuzz_engine4.txt
