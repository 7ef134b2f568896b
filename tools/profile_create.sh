#!/bin/bash
# tools/profile_create.sh [files] [per-file] -- run ON THE GPU BOX from the repo root (via gpurun).
# rocprofv3 kernel stats + FETCH_SIZE/WRITE_SIZE of `bin/igd create` at roadmap scale
# -> gpurun_out/profile_create/{kernel_stats.csv,pmc_fetch.csv,pmc_write.csv,traffic.json,create_bench.txt}
F=${1:-1900}; N=${2:-26316}
root=$PWD; out=$root/gpurun_out/profile_create; mkdir -p $out
bash tools/create_bench.sh $F $N > /dev/null 2>&1; cp gpurun_out/create_bench.txt $out/
D=/tmp/cb
cd /tmp && export TMPDIR=/tmp
rm -rf $D/p; rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $root/bin/igd create $D/in/ $D/p/ db > $out/stats.log 2>&1 || true
rm -rf $D/p; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $root/bin/igd create $D/in/ $D/p/ db > $out/fetch.log 2>&1 || true
rm -rf $D/p; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $root/bin/igd create $D/in/ $D/p/ db > $out/write.log 2>&1 || true
cd $root
cp $out/stats/*/*_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || true
cp $out/fetch/*/*_counter_collection.csv $out/pmc_fetch.csv 2>/dev/null || true
cp $out/write/*/*_counter_collection.csv $out/pmc_write.csv 2>/dev/null || true
python3 - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
def tot(path, counter):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]] += float(r["Counter_Value"])
    return acc
try:
    f, w = tot(out + "/pmc_fetch.csv", "FETCH_SIZE"), tot(out + "/pmc_write.csv", "WRITE_SIZE")
    res = {k: {"FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w.get(k, 0.0), "hbm_bytes_total": int((2 * f[k] + w.get(k, 0.0)) * 1024)} for k in f}
    res["_note"] = "summed over all launches of one create; 2*FETCH_SIZE + WRITE_SIZE, KB->bytes (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)"
    json.dump(res, open(out + "/traffic.json", "w"), indent=1)
except Exception as e:
    print("traffic: failed:", e)
PY
rm -rf $out/stats $out/fetch $out/write $out/pmc_fetch.csv $out/pmc_write.csv
ls $out
