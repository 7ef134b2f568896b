#!/usr/bin/env python3
"""tools/dense_one.py [steps] -- GPU box: config 4's per-GPU share (1.25e7 position-sorted queries) as one job, the handle closed at the
end (diagnostic builds dump their stamps there)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igd_amd import Database, synth
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
db = Database("/tmp/igdb/rm1900x26316.igd")
P = bench.CONFIG4_PER_GPU
q = synth.make_queries_slab(P, 0, P, seed=7, genome=synth.HG38)
job = bench.Job(db, dev, st.cuda_stream, *q, 0, 1)
el, prof = job.run(steps, 3)
print("dense share: step %.1f us scan %.1f us" % (1e6 * el / steps, 1e3 * prof["scan_ms"]))
del job; db.close()
