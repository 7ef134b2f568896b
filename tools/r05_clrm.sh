#!/bin/bash
# the clustered roadmap-scale database: bench rows + per-kernel times
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
python - <<'PY' > gpurun_out/r05/clrm.txt 2>&1
import sys, os, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from igd_amd import Database, synth
p = "/tmp/igdb/clrm1900x26316.igd"
if not os.path.exists(p + ".done"):
    synth.make_db(p, files=1900, per_file=26316, seed=1000, genome=synth.HG38, clustered=True); open(p + ".done", "w").write("ok")
db = Database(p)
print("records", db.nrecords, "tiles", db.ntiles)
q = synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=True)
dev = torch.device("cuda", 0)
t = [torch.from_numpy(x).to(dev) for x in q]
hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
for v in (0, 500):
    for k in range(5): db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(q[1]), hits.data_ptr(), None, v=v, flags=1)
    db.sync()
    db.profile_begin(30, every=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(30): db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(q[1]), hits.data_ptr(), None, v=v, flags=1)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    db.sync(); prof = db.profile_end()
    tr = db.batch_traffic(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(q[1]), v=v, flags=1)
    print("v", v, "step %.1f us kernel %.1f us bytes %.1f MB frac %.3f" % (el / 30 * 1e6, prof["scan_ms"] * 1e3, tr["total"] / 1e6, tr["total"] / (prof["scan_ms"] * 1e-3) / 8e12), db.last_scan_kernel())
PY
