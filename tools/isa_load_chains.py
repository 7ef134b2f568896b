#!/usr/bin/env python3
"""tools/isa_load_chains.py [file.hip] -- compile for gfx950 (-S) and list, per kernel, the places where a vector-memory load is
followed within three instructions by `s_waitcnt vmcnt(0)` and another load comes soon after: loads the compiler put one round
trip after the other.  Real chains (a table entry that names the next address) look the same; independent loads behind a divergent
branch -- `if (i < n) x = p[i];` with a default in x -- are the ones to fix: clamp the index or use a bounds-checked buffer load
(LABNOTES R5-12: the exact walks' five round trips per walk, k_split_fine's eight)."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "igd_amd/csrc/igd_hip.hip")
out = "/tmp/isa_chains.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"),
                "-I" + os.path.join(root, "igd_amd/csrc"), "-S", "--cuda-device-only", src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
lab = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", s, re.M)]
isload = lambda l: re.match(r"(global|buffer|flat)_load", l) is not None
rows = []
for n, (pos, name) in enumerate(lab):
    body = s[pos:lab[n + 1][0] if n + 1 < len(lab) else len(s)]
    if "s_endpgm" not in body: continue
    ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    chains = 0
    for k in range(len(ins)):
        if not isload(ins[k]): continue
        w = next((k + d for d in range(1, 4) if k + d < len(ins) and ins[k + d].startswith("s_waitcnt vmcnt(0)")), None)
        if w is not None and any(isload(ins[w + d]) for d in range(1, 13) if w + d < len(ins)): chains += 1
    rows.append((chains, len(ins), name))
rows.sort(reverse=True)
print("chains  instructions  kernel")
for c, n, name in rows:
    if c == 0: break
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print("%6d  %12d  %s" % (c, n, d[:130]))
