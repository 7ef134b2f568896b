import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from igd_amd import Database, synth
db = Database("/tmp/igdb/rm1900x26316.igd")
ichr, qs, qe = synth.make_queries(1000000, seed=7, genome=synth.HG38)
for _ in range(3):
    t=time.perf_counter(); qoff, rec = db.enumerate(ichr, qs, qe); print("enumerate %.1f ms" % ((time.perf_counter()-t)*1e3), len(rec))
