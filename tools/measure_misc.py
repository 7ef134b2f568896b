#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (run on the GPU box): PCIe-inclusive host-API rate,
-f enumeration rate, end-to-end CLI wall times next to the reference CLI, database open time."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from igd_amd import Database, synth

DIR = "/tmp/igdb"
path = os.path.join(DIR, "rm1900x26316.igd")
if not os.path.exists(path + ".done"):
    os.makedirs(DIR, exist_ok=True)
    synth.make_db(path)
    open(path + ".done", "w").write("ok")
out = {}
t = time.perf_counter(); db = Database(path); out["open_s"] = time.perf_counter() - t
out["resident_GB"] = db.resident_bytes / 1e9
Q = 1000000
ichr, qs, qe = synth.make_queries(Q, seed=7, genome=synth.HG38)
db.search(ichr, qs, qe)
best = 1e9
for _ in range(5):
    t = time.perf_counter(); h, tot = db.search(ichr, qs, qe, flags=1); best = min(best, time.perf_counter() - t)
out["host_api_hits_qps"] = Q / best
out["host_api_hits_ms"] = best * 1e3
best = 1e9
for _ in range(3):
    t = time.perf_counter(); qoff, rec = db.enumerate(ichr, qs, qe); best = min(best, time.perf_counter() - t)
out["enumerate_qps"] = Q / best
out["enumerate_records"] = int(len(rec))
out["enumerate_records_per_s"] = len(rec) / best
out["enumerate_ms"] = best * 1e3
db.close()
bed = os.path.join(DIR, "m_q.bed"); synth.write_bed(bed, synth.HG38, ichr, qs, qe)
sh = synth.make_queries(Q, seed=7, genome=synth.HG38, sorted_=False)
bedsh = os.path.join(DIR, "m_qs.bed"); synth.write_bed(bedsh, synth.HG38, *sh)
exe = os.path.join(ROOT, "bin", "igd"); ref = os.path.join(ROOT, "oracle", "_ref", "igd")
def wall(cmd, n=3):
    b = 1e9
    for _ in range(n):
        t = time.perf_counter(); subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); b = min(b, time.perf_counter() - t)
    return b
for tag, q in (("sorted", bed), ("shuffled", bedsh)):
    out["cli_gpu_%s_s" % tag] = wall([exe, "search", path, "-q", q])
    if os.path.exists(ref):
        out["cli_ref_%s_s" % tag] = wall([ref, "search", path, "-q", q])
out["cli_gpu_v500_s"] = wall([exe, "search", path, "-q", bed, "-v", "500"])
out["cli_gpu_f_s"] = wall([exe, "search", path, "-q", bed, "-f"], 2)
if os.path.exists(ref):
    out["cli_ref_v500_s"] = wall([ref, "search", path, "-q", bed, "-v", "500"])
    out["cli_ref_f_s"] = wall([ref, "search", path, "-q", bed, "-f"], 1)
print(json.dumps(out, indent=1))
