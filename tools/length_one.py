#!/usr/bin/env python3
"""tools/length_one.py <min_len> <max_len> <nq> [steps] -- GPU box: one position-sorted workload of the length probe, for
rocprofv3 --kernel-trace --stats (which kernel of the step the time of long queries goes to)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igd_amd import Database, synth
import bench
lo, hi, nq = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
db = Database("/tmp/igdb/rm1900x26316.igd")
q = synth.make_queries(nq, seed=7, genome=synth.HG38, min_len=lo, max_len=hi, sorted_=True)
job = bench.Job(db, dev, st.cuda_stream, *q, 0, 1)
el, prof = job.run(steps, 3)
print("len %d..%d nq %d: step %.1f us scan %.1f us" % (lo, hi, nq, 1e6 * el / steps, 1e3 * prof["scan_ms"]))
del job; db.close()
