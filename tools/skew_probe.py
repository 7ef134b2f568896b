#!/usr/bin/env python3
"""tools/skew_probe.py -- GPU box: step time of 10^6 position-sorted queries piled up in 1 / 10 / 100 / 1000 tiles of chr1 (lean
build: heavy_sorted_body + far_units_body), and of a dense batch (1.25e7 queries, full build) with 10^6 more in one tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
rng = np.random.default_rng(5)
Q = 1000000
def pile(span, n=Q):
    ps = np.sort((50000000 + rng.integers(0, 16384 * span, n)).astype(np.int32))
    return np.zeros(n, np.int32), ps, (ps + rng.integers(100, 2000, n)).astype(np.int32)
cases = [("10^6 in %d tiles" % s, pile(s)) for s in (1, 10, 100, 1000)]
d = synth.make_queries_slab(bench.CONFIG4_PER_GPU, 0, bench.CONFIG4_PER_GPU, seed=7, genome=synth.HG38)
p = pile(1)
ichr = np.concatenate([d[0], p[0]]); qs = np.concatenate([d[1], p[1]]); qe = np.concatenate([d[2], p[2]])
o = np.lexsort((qs, ichr))
cases.append(("dense 1.25e7 + 10^6 in 1 tile", (ichr[o], qs[o], qe[o])))
if len(sys.argv) > 1:                                   # `skew_probe.py 1`: only the cases whose name holds "in 1 tiles" (for rocprofv3 / counters)
    cases = [c for c in cases if ("in %s tiles" % sys.argv[1]) in c[0] and not c[0].startswith("dense")]
for name, q in cases:
    job = bench.Job(db, dev, st.cuda_stream, *q, 0, 1)
    el, prof = job.run(10, 2)
    print("%-32s | step %9.1f us scan %8.1f us | hits/step %d" % (name, 1e5 * el, 1e3 * prof["scan_ms"], int(job.d_hits.sum().item()) // 10), flush=True)
    del job
