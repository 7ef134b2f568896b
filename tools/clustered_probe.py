#!/usr/bin/env python3
"""tools/clustered_probe.py -- GPU box: the clustered stress database of bench.py's extra_configs on its own (for rocprofv3 / counters):
10^6 position-sorted queries per step, merge join."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from igd_amd import Database, synth
import bench
p = "/tmp/igdb/cl300x40000.igd"
if not os.path.exists(p + ".done"):
    os.makedirs("/tmp/igdb", exist_ok=True)
    synth.make_db(p, files=300, per_file=40000, seed=77, genome=synth.HG38, clustered=True)
    open(p + ".done", "w").write("ok")
db = Database(p)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
q = synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=True)
job = bench.Job(db, dev, st.cuda_stream, *q, 0, 1)
el, prof = job.run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, 3)
rl = job.roofline(prof)
cnt = np.array([c for ct in range(db.nctg) for c in []])
print("records %d tiles %d units? step %.1f us scan %.1f us frac %.3f bytes %.1f MB breakdown %s" % (db.nrecords, db.ntiles, 1e6 * el / 30, 1e3 * prof["scan_ms"], rl["frac"], rl["bytes_per_launch"] / 1e6, rl["bytes_breakdown"]))
