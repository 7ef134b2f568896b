#!/bin/bash
# round-5 baseline on one GPU box: the gpu tests, one bench line, and the CLI's phase table on the headline files
set -u
O=gpurun_out/r05
mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/tests0.txt 2>&1
python bench.py --steps 20 > $O/bench0.json 2> $O/bench0.err
DB=/tmp/igdb/rm1900x26316.igd
Q=/tmp/igdb/q1000000_sorted.bed
for i in 1 2 3 4 5 6; do
  /usr/bin/time -f "wall %e s" env IGD_TIMING=1 bin/igd search $DB -q $Q > /dev/null 2> $O/cli_timing_$i.txt
done
for i in 1 2 3; do
  /usr/bin/time -f "wall %e s" oracle/_ref/igd search $DB -q $Q > /dev/null 2> $O/ref_timing_$i.txt
done
nproc > $O/nproc.txt; grep -m1 "model name" /proc/cpuinfo >> $O/nproc.txt; free -g >> $O/nproc.txt
