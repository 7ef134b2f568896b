#!/usr/bin/env python3
"""tools/enum_chunk_probe.py -- GPU box: config 5 through the C API (igd_hip_enumerate_stream8 / _stream) for the chunk buffer size in
IGD_ENUM_CHUNK_HITS (16-byte overlaps per buffer; round 6: 262144 .. 4194304 -> 7.8 / 7.2 / 7.1 / 6.9 (default) / 6.9 ms in 8 bytes per overlap)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igd_amd import Database, synth
p = "/tmp/igdb/rm1900x26316.igd"
if not os.path.exists(p + ".done"):
    os.makedirs("/tmp/igdb", exist_ok=True)
    synth.make_db(p, files=1900, per_file=26316, seed=1000, nbp_log=14, genome=synth.HG38); open(p + ".done", "w").write("ok")
q = synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=True)
db = Database(p)
for fn, name in ((db.enumerate_stream8, "8 B"), (db.enumerate_stream, "16 B")):
    fn(*q)
    best = None
    for _ in range(4):
        t = time.perf_counter(); _, tot = fn(*q); dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    print(os.environ.get("IGD_ENUM_CHUNK_HITS", "default"), name, "%.2f ms" % (best * 1e3), tot, flush=True)
db.close()
