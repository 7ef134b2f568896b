#!/bin/bash
# `igd create` at roadmap scale on the GPU box: tools/create_bench.sh [files] [per-file]
# synthetic BED files -> bin/igd create (GPU) and the CPU oracle (igd_oracle create); the two .igd
# files are compared byte for byte.  Writes gpurun_out/create_bench.txt.
F=${1:-1900}; N=${2:-26316}
D=/tmp/cb; rm -rf $D; mkdir -p $D/in gpurun_out
bin/igd_synth beds $D/in --files $F --per-file $N > /dev/null
make -s -C oracle > /dev/null 2>&1
{
echo "== inputs: $F files x $N lines, $(du -sh $D/in | cut -f1)"
echo "== GPU: bin/igd create (first run includes HIP start-up)"
TIMEFORMAT="wall %R s  user %U s  sys %S s"
for i in 1 2; do rm -rf $D/g; time (IGD_TIMING=1 bin/igd create $D/in/ $D/g/ db 2>&1 | cut -c1-120); done
echo "== CPU oracle (port of the reference algorithm, in memory, 1 thread)"
rm -rf $D/o; time (oracle/_build/igd_oracle create $D/in/ $D/o/ db 2>&1 | tail -3 | cut -c1-120)
cmp $D/g/db.igd $D/o/db.igd && echo "db.igd: byte-identical (GPU vs oracle), $(stat -c %s $D/g/db.igd) bytes"
cmp $D/g/db_index.tsv $D/o/db_index.tsv && echo "db_index.tsv: byte-identical"
} 2>&1 | tee gpurun_out/create_bench.txt
