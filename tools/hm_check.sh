#!/bin/bash
# full-scale `-m`: this CLI vs the reference CLI (same .igd), byte comparison + wall times
python tools/prep.py >/dev/null
cd /tmp
for i in 1 2; do s=$(date +%s.%N); $GRAFT_REPO_ROOT/bin/igd search /tmp/igdb/rm1900x26316.igd -m -o /tmp/hm_gpu.txt > /tmp/hm.stdout; e=$(date +%s.%N); echo "gpu cli -m: $(echo "$e - $s" | bc) s"; done
md5sum /tmp/hm_gpu.txt
if [ -x $GRAFT_REPO_ROOT/oracle/_ref/igd ]; then
  s=$(date +%s.%N); $GRAFT_REPO_ROOT/oracle/_ref/igd search /tmp/igdb/rm1900x26316.igd -m -o /tmp/hm_ref.txt > /tmp/hmr.stdout; e=$(date +%s.%N); echo "ref cli -m: $(echo "$e - $s" | bc) s"
  md5sum /tmp/hm_ref.txt
  cmp /tmp/hm_gpu.txt /tmp/hm_ref.txt && cmp /tmp/hm.stdout /tmp/hmr.stdout && echo "IDENTICAL matrix file and stdout"
fi
