#!/usr/bin/env python3
"""tools/manyfiles_probe.py [files,..] -- GPU box: databases of many files (LDS counters up to 15 360 files, global atomics beyond),
5 x 10^7 intervals in all, 10^6 queries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igd_amd import Database, synth
import bench
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
os.makedirs("/tmp/igdb", exist_ok=True)
q = synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=True)
qu = synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=False)
qd = synth.make_queries_slab(bench.CONFIG4_PER_GPU, 0, bench.CONFIG4_PER_GPU, seed=7, genome=synth.HG38)
for files in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4000,12000,15360,16000,30000,60000").split(",")]:
    path = "/tmp/igdb/many%d.igd" % files
    if not os.path.exists(path + ".done"):
        synth.make_db(path, files=files, per_file=50000000 // files, seed=1000, nbp_log=14, genome=synth.HG38)
        open(path + ".done", "w").write("ok")
    db = Database(path)
    for qq, flags, name in ((q, 1, "sorted"), (qu, 2, "bucket"), (qd, 1, "dense")):
        job = bench.Job(db, dev, st.cuda_stream, *qq, 0, flags)
        el, prof = job.run(10, 2)
        print("files %5d (%8d tile records) | %-6s | step %8.1f us scan %8.1f us | hits/step %d" %
              (files, db.nrecords, name, 1e5 * el, 1e3 * prof["scan_ms"], int(job.d_hits.sum().item()) // 10), flush=True)
        del job
    db.close()
    os.remove(path); os.remove(path + ".done")
