#!/bin/bash
# tools/profile.sh <tag> [bench args...]   -- run ON THE GPU BOX (via gpurun), from the repo root.
# Writes, under gpurun_out/profile_<tag>/ (copy what you want judged into profiles/):
#   bench.json          the bench line of the same command, un-profiled
#   kernel_stats.csv    rocprofv3 --kernel-trace --stats   (per-kernel calls / average ns)
#   pmc_fetch.csv       rocprofv3 --pmc FETCH_SIZE         (separate pass)
#   pmc_write.csv       rocprofv3 --pmc WRITE_SIZE         (separate pass)
#   traffic.json        HBM bytes per launch of igd_scan_tiles: 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes);
#                       the factor 2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section)
# rocprofv3 gets the program itself after `--` (python3 bench.py ...), never a shell or env wrapper.
set -e
tag=$1; shift
root=$PWD
out=$root/gpurun_out/profile_$tag
mkdir -p $out
python3 bench.py --no-extra "$@" > $out/bench.json 2> $out/bench.err || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu --no-extra --no-cold "$@" > $out/stats.log 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --no-cpu --no-extra --no-cold --steps 10 --warmup 2 "$@" > $out/fetch.log 2>&1 || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --no-cpu --no-extra --no-cold --steps 10 --warmup 2 "$@" > $out/write.log 2>&1 || true
cd $root
cp $out/stats/*/*_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || true
cp $out/fetch/*/*_counter_collection.csv $out/pmc_fetch.csv 2>/dev/null || true
cp $out/write/*/*_counter_collection.csv $out/pmc_write.csv 2>/dev/null || true
python3 - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
def avg(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}
try:
    (f, nf), (w, _) = avg(out + "/pmc_fetch.csv", "FETCH_SIZE"), avg(out + "/pmc_write.csv", "WRITE_SIZE")
    res = {}
    best = -1
    for k in f:
        # the variant of the timed steps: most launches (set-up code may run another variant once; the gated twin reads ~nothing)
        if ("igd_scan_sorted" in k or "igd_scan_tiles" in k or "igd_scan_direct" in k) and f[k] > 1000 and nf[k] > best:
            best = nf[k]
            res = {"kernel": k, "launches_sampled": nf[k], "FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w.get(k, 0.0),
                   "hbm_bytes_per_launch": int((2 * f[k] + w.get(k, 0.0)) * 1024),
                   "note": "2*FETCH_SIZE + WRITE_SIZE, KB->bytes; factor 2 = gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM)"}
    json.dump(res, open(out + "/traffic.json", "w"), indent=1)
    print(json.dumps(res))
except Exception as e:
    print("traffic: failed:", e)
PY
rm -rf $out/stats $out/fetch $out/write
ls $out
