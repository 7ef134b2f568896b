#!/usr/bin/env python3
"""tools/tilesize_probe.py -- GPU box: the roadmap-scale files bucketed with small tiles (-b 10, 11, 12): most queries then cover
several tiles, queries of 5-20 kbp more than four (coverage difference arrays + exact walk of the last tile)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igd_amd import Database, synth
import bench
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
for b in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "14,12,11,10".split(","))]:
    path = "/tmp/igdb/rm1900x26316_b%d.igd" % b
    os.makedirs("/tmp/igdb", exist_ok=True)
    if not os.path.exists(path + ".done"):
        t = time.time()
        synth.make_db(path, files=1900, per_file=26316, seed=1000, nbp_log=b, genome=synth.HG38)
        open(path + ".done", "w").write("ok")
    db = Database(path)
    for lo, hi in ((100, 1999), (5000, 20000)):
        q = synth.make_queries(1000000, seed=7, genome=synth.HG38, min_len=lo, max_len=hi, sorted_=True)
        for flags, name in ((1, "sorted"), (2, "bucket")):
            job = bench.Job(db, dev, st.cuda_stream, *q, 0, flags)
            el, prof = job.run(10, 2)
            print("-b %2d (%8d tile records) len %5d..%5d | %-6s | step %9.1f us scan %8.1f us | hits/step %d" %
                  (b, db.nrecords, lo, hi, name, 1e5 * el, 1e3 * prof["scan_ms"], int(job.d_hits.sum().item()) // 10), flush=True)
            del job
    db.close()
