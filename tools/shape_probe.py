#!/usr/bin/env python3
"""tools/shape_probe.py -- GPU box: 10^6 position-sorted queries per step against databases of several shapes (files x intervals,
tile size): is any of them served much worse than its bytes explain?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igd_amd import Database, synth
import bench
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
q = synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=True)
os.makedirs("/tmp/igdb", exist_ok=True)
for files, per_file, nbp_log, clustered in ((8, 10000, 14, False), (40, 30000, 14, False), (300, 40000, 14, False), (300, 40000, 14, True),
                                            (1900, 2632, 14, False), (1900, 26316, 12, False), (1900, 26316, 15, False), (12, 1000000, 14, False)):
    p = "/tmp/igdb/sh_%d_%d_%d_%d.igd" % (files, per_file, nbp_log, clustered)
    synth.make_db(p, files=files, per_file=per_file, seed=77, nbp_log=nbp_log, genome=synth.HG38, clustered=clustered)
    db = Database(p)
    for flags, name in ((1, "sorted"), (2, "bucket")):
        job = bench.Job(db, dev, st.cuda_stream, *q, 0, flags)
        el, prof = job.run(20, 3)
        rl = job.roofline(prof)
        print("files %5d x %7d nbp 2^%d %s | %-6s | records %9d tiles %7d | step %7.1f us scan %7.1f us | %6.1f MB frac %.3f | kernel %s" % (
            files, per_file, nbp_log, "clustered" if clustered else "uniform  ", name, db.nrecords, db.ntiles, 1e6 * el / 20, 1e3 * prof["scan_ms"],
            rl["bytes_per_launch"] / 1e6, rl["frac"], rl["kernel"]), flush=True)
        del job
    db.close()
    os.unlink(p)
