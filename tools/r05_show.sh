#!/bin/bash
cd gpurun_out/r05; tail -3 direct_tests.txt; for f in perf_*_direct.json; do python - $f <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j['roofline']
    print(sys.argv[1], "step %.1f us kernel %s %.1f us frac %.3f step_frac %.3f matches %s bytes %.1f MB" % (j['ms_per_step']*1e3, r['kernel'], r['kernel_ms']*1e3, r['frac'], r['step_frac'], j.get('matches_oracle'), r['bytes_per_launch']/1e6))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done; for f in kstats_*_direct.txt; do echo "== $f"; grep -E "igd_scan|k_tile|k_query|k_reduce" $f; done
