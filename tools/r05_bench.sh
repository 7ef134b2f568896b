#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
( time python bench.py --steps 20 > $O/bench1.json 2> $O/bench1.err ) 2> $O/bench1.time
