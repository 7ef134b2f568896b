#!/usr/bin/env python3
"""tools/hotwin_probe.py -- GPU box: a hot region (90 % of 10^5 queries inside 1 or 30 tiles) against databases of 1900 and 20 000 files
(windows of files), sorted, with and without -v."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from igd_amd import Database, synth
import bench
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
os.makedirs("/tmp/igdb", exist_ok=True)
for files in (1900, 20000):
    path = "/tmp/igdb/hw%d.igd" % files
    if not os.path.exists(path + ".done"):
        synth.make_db(path, files=files, per_file=40000000 // files, seed=5, nbp_log=15, genome=synth.HG38)
        open(path + ".done", "w").write("ok")
    db = Database(path)
    for w in (1, 30):
        for n in (100000, 1000000):
            rng = np.random.default_rng(3)
            ichr, qs, qe = synth.make_queries(n, seed=9, genome=synth.HG38, min_len=100, max_len=1999, sorted_=False)
            m = int(n * 0.9); t0 = 4000 << 15
            qs[:m] = t0 + rng.integers(0, w << 15, m); qe[:m] = qs[:m] + rng.integers(100, 2000, m); ichr[:m] = 0
            o = np.lexsort((qs, ichr))
            for v in (0, 500):
                for flags, q, name in ((1, (ichr[o], qs[o], qe[o]), "sorted"), (0, (ichr, qs, qe), "unordered")):
                    job = bench.Job(db, dev, st.cuda_stream, *q, v, flags)
                    el, prof = job.run(5, 2)
                    print("files %5d | 90%% of %7d queries in %2d tiles v=%3d | %-9s | step %8.1f us scan %7.1f us | hits %d" %
                          (files, n, w, v, name, 2e5 * el, 1e3 * prof["scan_ms"], int(job.d_hits.sum().item()) // 5), flush=True)
                    del job
    db.close()
