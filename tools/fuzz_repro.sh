#!/bin/bash
# tools/fuzz_repro.sh <seed> -- re-run one case of tools/fuzz_gpu.py and show how `search -q -f` differs (GPU box)
seed=$1
python tools/fuzz_gpu.py 1 $seed | tail -2
d=/tmp/igz_bad_$seed
[ -d $d ] || exit 0
bin/igd search $d/gpu/db.igd -q $d/q.bed -f > gpurun_out/f_gpu.txt 2>gpurun_out/f_gpu.err
oracle/_build/igd_oracle search $d/gpu/db.igd -q $d/q.bed -f > gpurun_out/f_orc.txt
wc -l gpurun_out/f_gpu.txt gpurun_out/f_orc.txt
diff gpurun_out/f_gpu.txt gpurun_out/f_orc.txt | head -20
wc -l $d/q.bed; cp $d/q.bed gpurun_out/q_bad.bed; cp $d/gpu/db.igd gpurun_out/db_bad.igd; cp $d/gpu/db_index.tsv gpurun_out/ 2>/dev/null
for i in 1 2 3; do bin/igd search $d/gpu/db.igd -q $d/q.bed -f | md5sum; done
IGD_ENUM_CHUNK_HITS=64 bin/igd search $d/gpu/db.igd -q $d/q.bed -f | md5sum
oracle/_build/igd_oracle search $d/gpu/db.igd -q $d/q.bed -f | md5sum
