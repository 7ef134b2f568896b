#!/usr/bin/env python3
"""tools/cli_small_probe.py [--gpu] -- wall time of `bin/igd search <roadmap-scale db> -q <n queries>` next to the reference
binary's (oracle/_ref/igd), n = 10^3 .. 10^6, best of 5, stdout compared byte for byte.  Without --gpu only the sizes the
host path takes (igd_hostpath.c) are run through the product; with --gpu every size additionally with IGD_HOST_MAX_QUERIES=0
(everything on the engine) -- the table that places the crossover."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
D = "/tmp/igdb"
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "prep.py")], stdout=subprocess.DEVNULL)
db = os.path.join(D, "rm1900x26316.igd")
IGD, REF, SYN = os.path.join(ROOT, "bin", "igd"), os.path.join(ROOT, "oracle", "_ref", "igd"), os.path.join(ROOT, "bin", "igd_synth")


def best(cmd, env=None, n=5):
    t, out = 1e9, None
    for _ in range(n):
        t0 = time.time()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        t = min(t, time.time() - t0)
        assert p.returncode == 0, p.stderr.decode()[-300:]
        out = p.stdout
    return t, out


gpu = "--gpu" in sys.argv
print("%9s %10s %10s %10s" % ("queries", "reference", "product", "engine-only" if gpu else ""))
for n in (1000, 10000, 50000, 100000, 150000, 300000, 1000000):
    q = os.path.join(D, "q%d.bed" % n)
    if not os.path.exists(q):
        subprocess.check_call([SYN, "queries", q, "--n", str(n)], stdout=subprocess.DEVNULL)
    for extra in ([], ["-v", "500"]):
        tr, orf = best([REF, "search", db, "-q", q] + extra) if os.path.exists(REF) else (float("nan"), None)
        env = dict(os.environ)
        if not gpu:
            env["IGD_HOST_MAX_QUERIES"] = "100000000"
        tp, op = best([IGD, "search", db, "-q", q] + extra, env=env)
        te = float("nan")
        if gpu:
            te, oe = best([IGD, "search", db, "-q", q] + extra, env=dict(os.environ, IGD_HOST_MAX_QUERIES="0"))
            assert oe == op
        assert orf is None or orf == op, "stdout differs from the reference's at n=%d %s" % (n, extra)
        print("%9d %9.1fms %9.1fms %9.1fms  %s" % (n, 1e3 * tr, 1e3 * tp, 1e3 * te, " ".join(extra)), flush=True)
