#!/usr/bin/env python3
"""tools/length_probe.py -- GPU box: 10^6 position-sorted queries of several length ranges against the roadmap database
(later-tile entries per query, far units, the exact walk of queries longer than four tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igd_amd import Database, synth
import bench
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
PATH = "/tmp/igdb/rm1900x26316.igd"
if not os.path.exists(PATH + ".done"):
    os.makedirs(os.path.dirname(PATH), exist_ok=True)
    synth.make_db(PATH, files=1900, per_file=26316, seed=1000, nbp_log=14, genome=synth.HG38)
    open(PATH + ".done", "w").write("ok")
db = Database(PATH)
for lo, hi in ((100, 1999), (5000, 20000), (20000, 60000), (60000, 70000), (100000, 200000)):
    for nq in (1000000, 100000):
        q = synth.make_queries(nq, seed=7, genome=synth.HG38, min_len=lo, max_len=hi, sorted_=True)
        for flags, name in ((1, "sorted"), (2, "bucket")):
            job = bench.Job(db, dev, st.cuda_stream, *q, 0, flags)
            el, prof = job.run(10, 2)
            print("len %6d..%6d nq %7d | %-6s | step %9.1f us scan %8.1f us | hits/step %d" % (lo, hi, nq, name, 1e6 * el / 10, 1e3 * prof["scan_ms"], int(job.d_hits.sum().item()) // 10), flush=True)
            del job
