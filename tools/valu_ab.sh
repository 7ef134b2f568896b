#!/bin/bash
# tools/valu_ab.sh <out tag> [bench args] -- GPU box: SQ_INSTS_VALU / SALU / LDS and duration of the scan kernel for every library build (IGD_EXP section variants)
out=$PWD/gpurun_out/$1; shift; mkdir -p $out; root=$PWD
python tools/prep.py > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for d in $root/igd_amd/lib $root/igd_amd/libv_*; do
  [ -f $d/libigd_hip.so ] || continue
  tag=$(basename $d)
  export IGD_HIP_ALLOW_EXP_BUILD=1   # the section variants give wrong counts on purpose
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVES"; do
    g=$(echo $grp | tr ' ' '_')
    IGD_AMD_LIBDIR=$d rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$tag/$g -- python3 $root/bench.py --no-cpu --no-extra --no-cold --steps 6 --warmup 2 "$@" > /dev/null 2>&1
  done
  python3 - $out/$tag $tag <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list); dur = []
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "igd_scan_sorted" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/*/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "igd_scan_sorted" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-12s" % sys.argv[2], " ".join("%s %.4g" % (k.replace("SQ_INSTS_", ""), sum(v) / len(v)) for k, v in sorted(acc.items())), "us(profiled) %.1f" % (sum(dur) / max(1, len(dur))))
PY
done
