#!/bin/bash
# tools/r3k.sh <tag> -- GPU box: per-kernel times (rocprofv3 --stats) of the headline, dense share and slab-of-8 batches
out=gpurun_out/$1; mkdir -p $out
python tools/prep.py > /dev/null 2>&1
for cfg in "default:" "dense:--queries 12500000" "slab8:--slab-of 8"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  echo "== $tag" >> $out/kstats.txt
  bash tools/kstats.sh --no-extra $args >> $out/kstats.txt 2>&1
  python bench.py --no-cpu --no-extra --steps 100 --warmup 5 $args > $out/bench_$tag.json 2> $out/bench_$tag.err
  python3 - $out/bench_$tag.json $tag >> $out/kstats.txt <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); r = j["roofline"]
    print("%-10s step %7.1f us  scan %7.1f us  pipeline %7.1f us  frac %.3f  hits %d oracle %s" % (sys.argv[2], 1e3 * j["ms_per_step"], 1e3 * r["kernel_ms"], 1e3 * r["pipeline_ms"], r["frac"], j["hits_per_step_total"], j.get("matches_oracle")))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
done
cat $out/kstats.txt
