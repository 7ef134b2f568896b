#!/usr/bin/env python3
"""tools/density_probe.py -- GPU box: lean vs full build of igd_scan_sorted by queries per tile (roadmap database)."""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for nq in (1000000, 1500000, 2000000, 3000000, 4000000, 6000000):
    row = []
    for rank in ("0", "1"):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu", "--no-extra", "--no-cold", "--steps", "40", "--queries", str(nq)],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, IGD_HIP_RANK=rank))
        j = json.loads(p.stdout.decode().strip().splitlines()[-1])
        row.append("%s: step %6.1f scan %6.1f us" % ("lean" if rank == "0" else "full", 1e3 * j["ms_per_step"], 1e3 * j["roofline"]["kernel_ms"]))
    print("nq %8d (%.1f per tile)  %s" % (nq, nq / 188505.0, "   ".join(row)), flush=True)
