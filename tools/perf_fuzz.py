#!/usr/bin/env python3
"""tools/perf_fuzz.py [cases] [seed0] -- GPU box: step time of random (database, batch) shapes, to FIND shapes that are far slower than
their size explains (the round's probes each found one).  Per case: records, queries, hits, step time sorted and unordered, and
the time per (query + hit/32 + record/64) "work unit" -- outliers of that column are worth a probe."""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from igd_amd import Database, synth
import bench
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = []
for ci in range(cases):
    rng = np.random.default_rng(seed0 + ci)
    d = tempfile.mkdtemp(prefix="igp", dir="/tmp")
    try:
        b = int(rng.choice([12, 13, 14, 14, 15])); files = int(rng.choice([1, 3, 12, 100, 1900, 6000, 20000]))
        total = int(rng.choice([2e6, 1e7, 4e7])); per = max(1, total // files); cl = bool(rng.random() < 0.3); gt = int(rng.choice([0, 1, 1]))
        path = os.path.join(d, "p.igd")
        synth.make_db(path, files=files, per_file=per, seed=int(rng.integers(1, 1 << 30)), nbp_log=b, genome=synth.HG38, clustered=cl, gtype=gt)
        db = Database(path)
        n = int(rng.choice([1e4, 1e5, 1e6, 4e6]))
        lo, hi = [(100, 1999), (1, 200), (2000, 30000), (30000, 300000)][int(rng.integers(0, 4))]
        ichr, qs, qe = synth.make_queries(n, seed=int(rng.integers(1, 1 << 30)), genome=synth.HG38, min_len=lo, max_len=hi, sorted_=False)
        hotf = float(rng.choice([0, 0, 0.1, 0.9]))
        if hotf > 0:
            m = int(n * hotf); t0 = int(rng.integers(1000, 9000)) * (1 << b); w = int(rng.choice([1, 30])) << b
            qs[:m] = t0 + rng.integers(0, w, m); qe[:m] = qs[:m] + rng.integers(lo, hi + 1, m); ichr[:m] = 0
        v = int(rng.choice([0, 0, 500]))
        o = np.lexsort((qs, ichr))
        res = []
        for q, flags in (((ichr[o], qs[o], qe[o]), 1), ((ichr, qs, qe), 0)):
            job = bench.Job(db, dev, st.cuda_stream, *q, v, flags)
            el, prof = job.run(4, 1)
            res.append((1e6 * el / 4, int(job.d_hits.sum().item()) // 4))
            del job
        hits = res[0][1]
        work = n + hits / 32 + db.nrecords / 64
        rows.append((res[0][0] * 1e3 / work, "b=%2d files=%5d rec=%8d %s gT%d | n=%7d len %6d..%6d hot %.1f v=%3d | hits %10d | sorted %9.1f us  unordered %9.1f us | %6.2f ns/work" %
                     (b, files, db.nrecords, "cl" if cl else "un", gt, n, lo, hi, hotf, v, hits, res[0][0], res[1][0], res[0][0] * 1e3 / work)))
        print(rows[-1][1], flush=True)
        db.close()
    except Exception as e:
        print("case %d failed: %s" % (seed0 + ci, e), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
print("---- slowest per unit of work")
for r in sorted(rows, reverse=True)[:8]: print(r[1])
