#!/bin/bash
# tools/regs.sh [file.hip] [name filter] -- registers, spills and occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage)
src=${1:-igd_amd/csrc/igd_hip.hip}; filt=${2:-.}
mkdir -p /tmp/regs && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude $EXTRA -Rpass-analysis=kernel-resource-usage -c $src -o /tmp/regs/x.o 2>&1 |
python3 -c '
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name": cur = v; rows[cur] = {}
    elif cur: rows[cur][k] = v
import subprocess
for n, r in rows.items():
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    print("%-74s sgpr %3s vgpr %3s  spill s %3s v %3s  scratch %4s  occ %s" % (d[:74], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
' | grep -E "$filt"
