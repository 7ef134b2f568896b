#!/usr/bin/env python3
"""tools/shape_ab.py -- GPU box: scan-kernel time (HIP events on the kernel's own dispatch) and step time of the headline's 10^6 sorted
queries against databases of different shapes -- uniform roadmap scale, the same with real-data clustering, the small clustered and
the sparse one -- for the library build named by IGD_AMD_LIBDIR.  One line per database."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igd_amd import Database, synth
D = "/tmp/igdb"
dbs = [("uniform  rm1900x26316", "rm1900x26316", dict()),
       ("clustered roadmap     ", "clrm1900x26316", dict(files=1900, per_file=26316, seed=1000, genome=synth.HG38, clustered=True)),
       ("clustered 300x40000   ", "cl300x40000", dict(files=300, per_file=40000, seed=77, genome=synth.HG38, clustered=True)),
       ("sparse 100x1000       ", "sparse100x1000", dict(files=100, per_file=1000, seed=31, genome=synth.HG38))]
q = synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=True)
dev = torch.device("cuda", 0)
t = [torch.from_numpy(x).to(dev) for x in q]
for name, tag, kw in dbs:
    p = os.path.join(D, tag + ".igd")
    if not os.path.exists(p + ".done"):
        synth.make_db(p, **kw); open(p + ".done", "w").write("ok")
    db = Database(p)
    hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
    for v in (0, 500):
        for k in range(5): db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(q[1]), hits.data_ptr(), None, v=v, flags=1)
        db.sync()
        best = None
        for rep in range(3):
            db.profile_begin(40, every=1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(40): db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(q[1]), hits.data_ptr(), None, v=v, flags=1)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            db.sync(); prof = db.profile_end()
            if best is None or prof["scan_ms"] < best[1]: best = (el / 40 * 1e6, prof["scan_ms"])
        print("%s v=%3d  step %6.1f us  kernel %6.1f us  (%s, %s)" % (name, v, best[0], best[1] * 1e3, db.last_scan_kernel(), os.environ.get("IGD_AMD_LIBDIR", "lib")), flush=True)
    db.close()
