#!/usr/bin/env python3
"""tools/skew_bucket_probe.py [tiles] -- GPU box: 10^6 UNORDERED queries inside `tiles` tiles of chr1, bucket path, 12 batches (run
under rocprofv3 --kernel-trace --stats to see which kernel the time goes to)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from igd_amd import Database, synth
import bench
span = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
PATH = "/tmp/igdb/rm1900x26316.igd"
if not os.path.exists(PATH + ".done"):
    os.makedirs(os.path.dirname(PATH), exist_ok=True)
    synth.make_db(PATH, files=1900, per_file=26316, seed=1000, nbp_log=14, genome=synth.HG38)
    open(PATH + ".done", "w").write("ok")
db = Database(PATH)
rng = np.random.default_rng(5)
Q = 1000000
qs = (50000000 + rng.integers(0, 16384 * span, Q)).astype(np.int32)
q = (np.zeros(Q, np.int32), qs, (qs + rng.integers(100, 2000, Q)).astype(np.int32))
job = bench.Job(db, dev, st.cuda_stream, *q, 0, 2)
el, prof = job.run(10, 2)
print("10^6 unordered in %d tiles | step %9.1f us" % (span, 1e5 * el))
