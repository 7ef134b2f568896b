#!/bin/bash
# tools/variants.sh <out_dir> <bench args...> -- on the GPU box: run bench.py once per kernel-variant library directory
# igd_amd/libv_* (built here with `make LIB=igd_amd/libv_<tag> EXTRA=-D...`), print scan kernel / step times.
out=$1; shift
mkdir -p $out
python tools/prep.py > /dev/null 2>&1
for d in igd_amd/lib igd_amd/libv_*; do
  [ -f $d/libigd_hip.so ] || continue
  tag=$(basename $d)
  IGD_AMD_LIBDIR=$PWD/$d python bench.py --no-cpu --no-extra "$@" > $out/$tag.json 2> $out/$tag.err
  python3 - $out/$tag.json $tag <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    r = j["roofline"]
    print("%-16s step %7.1f us  scan %7.1f us  pipeline %7.1f us  frac %.3f  hits %d" % (sys.argv[2], 1e3 * j["ms_per_step"], 1e3 * r["kernel_ms"], 1e3 * r["pipeline_ms"], r["frac"], j["hits_per_step_total"]))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
done
