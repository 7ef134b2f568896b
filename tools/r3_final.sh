#!/bin/bash
# tools/r3_final.sh -- GPU box: the bench lines quoted in DESIGN.md section 9 (round 3)
out=gpurun_out/r3final; mkdir -p $out
python tools/prep.py > /dev/null 2>&1
python bench.py > $out/default.json 2> $out/default.err
for cfg in "auto:--grouping auto" "shuffled:--shuffled" "v500:--v 500" "exact:--exact-arrays" "dense:--queries 12500000 --steps 50" "slab8:--slab-of 8 --steps 50" "slab4:--slab-of 4 --steps 50" "slab2:--slab-of 2 --steps 50" "q1e3:--queries 1000" "q1e4:--queries 10000" "q1e5:--queries 100000"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu --no-extra --no-cold $args > $out/$tag.json 2> $out/$tag.err
done
python3 - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        j = json.load(open(f)); r = j["roofline"]
        print("%-10s value %.3g step %7.1f us scan %7.1f pipeline %7.1f frac %.3f bytes %.1f MB oracle %s" % (os.path.basename(f)[:-5], j["value"], 1e3 * j["ms_per_step"], 1e3 * r["kernel_ms"], 1e3 * r["pipeline_ms"], r["frac"], r["bytes_per_launch"] / 1e6, j.get("matches_oracle")))
    except Exception as e:
        print(f, "failed", e)
PY
