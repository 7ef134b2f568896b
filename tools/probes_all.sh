#!/bin/bash
# tools/probes_all.sh -- GPU box: every shape probe on the current build, one text file (profiles/rNN/probes.txt quotes it)
out=gpurun_out/probes.txt; : > $out
for p in "length_probe.py" "skew_probe.py" "skew_bucket_probe.py 1" "skew_bucket_probe.py 10" "skew_bucket_probe.py 1000" "fewfiles_probe.py 1,2,4,8,16,64" \
         "manyfiles_probe.py 4000,12000,15360,16000,30000,60000" "tilesize_probe.py 18,16,15,14,12,11,10" "clustered_probe.py" "density_probe.py"; do
  echo "== tools/$p" >> $out
  timeout 900 python tools/$p 2>&1 | grep -v "amdgpu.ids" >> $out
done
tail -5 $out
