#!/usr/bin/env python3
"""Pin the behaviour of the reference's PYTHON wrapper (src_py/igd_py.pyx) on the `smallrand`
fixture: build the reference's Cython extension in /tmp (never in the repo), import it, and
record what igd_py().open/get_nFiles/search_n/search_1 return.  Output: tests/golden/pywrap.json
(data only).  Build-container only (needs /root/reference and Cython)."""
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src_py"
TMP = "/tmp/ig_srcpy"


def main():
    shutil.rmtree(TMP, ignore_errors=True)
    shutil.copytree(REF, TMP)
    for f in ("igd_py.c",):                     # stale Cython 0.26 output: regenerate
        p = os.path.join(TMP, f)
        if os.path.exists(p):
            os.remove(p)
    import numpy
    env = dict(os.environ, CFLAGS="-I" + numpy.get_include() + " -w")
    subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=TMP, env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    work = "/tmp/ig_pw"
    shutil.rmtree(work, ignore_errors=True)
    shutil.copytree(os.path.join(HERE, "smallrand"), work)
    code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np, igd_py
g = igd_py.igd_py()
g.open(%r)
n = g.get_nFiles()
h = np.zeros(n, dtype="int64")
tot = g.search_n(%r, h)
one = {}
for (c, s, e) in [("chr1", 1000000, 1100000), ("chr2", 0, 50000), ("chrX", 3990000, 4100000), ("chr9", 5, 10)]:
    v = np.zeros(n, dtype="int64")
    g.search_1(c, s, e, v)
    one["%%s:%%d-%%d" %% (c, s, e)] = v.tolist()
print(json.dumps({"nFiles": n, "search_n_return": int(tot), "search_n_hits": h.tolist(), "search_1": one}))
''' % (TMP, os.path.join(work, "db.igd"), os.path.join(work, "q.bed"))
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, check=True).stdout.decode()
    data = json.loads(out.strip().splitlines()[-1])
    json.dump(data, open(os.path.join(HERE, "pywrap.json"), "w"), indent=1)
    print("pywrap.json:", data["nFiles"], data["search_n_return"])


if __name__ == "__main__":
    main()
