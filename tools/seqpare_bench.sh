#!/bin/bash
# Seqpare at scale on the GPU box: tools/seqpare_bench.sh [queries...]   (roadmap-scale DB from tools/prep.py)
# bin/igd search -s (GPU) vs the CPU oracle (port of the reference algorithm); outputs compared.
mkdir -p gpurun_out; D=/tmp/igdb
python tools/prep.py > /dev/null 2>&1
make -s -C oracle > /dev/null 2>&1
DB=$(ls $D/*.igd | head -1)
TIMEFORMAT="%R s"
{
echo "== database $DB"
for Q in "$@"; do
  bin/igd_synth queries /tmp/sq_$Q.bed --n $Q --seed 11 > /dev/null
  echo "-- $Q queries"
  echo -n "GPU  bin/igd search -s : "; time (bin/igd search $DB -q /tmp/sq_$Q.bed -s > /tmp/sq_gpu_$Q.txt)
  if [ $Q -le 200000 ]; then
    echo -n "CPU  oracle search -s  : "; time (oracle/_build/igd_oracle search $DB -q /tmp/sq_$Q.bed -s > /tmp/sq_orc_$Q.txt)
    cmp /tmp/sq_gpu_$Q.txt /tmp/sq_orc_$Q.txt && echo "outputs identical ($(wc -l < /tmp/sq_gpu_$Q.txt) lines)"
  fi
done
} 2>&1 | tee gpurun_out/seqpare_bench.txt
