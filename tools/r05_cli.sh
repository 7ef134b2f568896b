#!/bin/bash
# the command line tool on the headline's files: phase table (IGD_TIMING), the host path at larger sizes, the reference beside it
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
python - <<'PY' > gpurun_out/r05/cli_walls.txt 2>&1
import os, subprocess, time, sys
sys.path.insert(0, '.')
from igd_amd import synth
DB = "/tmp/igdb/rm1900x26316.igd"
def wall(cmd, env=None, n=5):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env); ts.append(time.perf_counter() - t)
    return min(ts), max(ts), p.stdout
for nq in (300000, 1000000, 2000000, 4000000):
    q = "/tmp/igdb/w%d.bed" % nq
    if not os.path.exists(q):
        synth.write_bed(q, synth.HG38, *synth.make_queries(nq, seed=7, genome=synth.HG38, sorted_=True))
    ref = wall(["oracle/_ref/igd", "search", DB, "-q", q], n=3)
    eng = wall(["bin/igd", "search", DB, "-q", q], env=dict(os.environ, IGD_HOST_MAX_QUERIES="0"))
    host = wall(["bin/igd", "search", DB, "-q", q], env=dict(os.environ, IGD_HOST_MAX_QUERIES="100000000"))
    host8 = wall(["bin/igd", "search", DB, "-q", q], env=dict(os.environ, IGD_HOST_MAX_QUERIES="100000000", IGD_HOST_THREADS="8"))
    print("nq %8d  reference %.3f-%.3f  engine %.3f-%.3f  host %.3f-%.3f  host(8 thr) %.3f-%.3f  same %s" % (nq, ref[0], ref[1], eng[0], eng[1], host[0], host[1], host8[0], host8[1], ref[2] == eng[2] == host[2]), flush=True)
PY
