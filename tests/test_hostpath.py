"""Small query files are answered on the host (igd_amd/csrc/igd_hostpath.c, product code): the reference's cheap start
for small jobs -- header only, then the tiles the queries touch (src/igd_base.c:269-323, src/igd_search.c:469-476) --
instead of 0.18 s of HIP start-up and upload for a few thousand queries.  These tests run on the GPU-less container:

  - every golden `search` command line (the REAL reference's stdout, 7 fixture families: -q, -v N, -f, -r, gzip, CRLF)
    printed byte for byte by bin/igd without a GPU;
  - a differential fuzz of the host path against the oracle (both rules, -v, -f order, every thread count);
  - the Python and R flavours' file entry points on the host;
  - the limit is about the NUMBER OF QUERIES only: with IGD_HOST_MAX_QUERIES=0, or a file above the limit, a host
    without a usable GPU fails loudly (no CPU fallback for batches).
"""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, Oracle, short_tmpdir
from test_golden_oracle import CASES, materialize

EXE = os.path.join(ROOT, "bin", "igd")


def _no_gpu_env(**kw):
    """the host path must not depend on a device: hide any GPU from the child"""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    env.pop("IGD_HOST_MAX_QUERIES", None)
    env.update(kw)
    return env


@pytest.mark.parametrize("case", CASES)
def test_cli_prints_the_references_stdout_for_small_files_without_a_gpu(case):
    d, dst, man = materialize(case)
    try:
        n = 0
        for run in man["runs"]:
            if "-s" in run["args"] or "-m" in run["args"]:
                continue                                  # Seqpare and the hit map are GPU paths at any size
            args = [os.path.join(dst, a) if a in ("db.igd", "q.bed", "q.bed.gz", "q100.bed") else a for a in run["args"]]
            for threads in ("1", "3"):
                p = subprocess.run([EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                                   env=_no_gpu_env(IGD_HOST_THREADS=threads, IGD_TIMING="1"))
                assert p.returncode == 0, p.stderr.decode()[-300:]
                assert p.stdout.decode() == open(os.path.join(dst, run["stdout"])).read(), (case, run["args"], threads)
                assert b"GPU" not in p.stderr.replace(b"database -> GPU", b"") or b"on the host" in p.stderr
            n += 1
        assert n > 0
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _bind(N):
    L = N.cli()
    L.igdc_map_open.restype = C.c_void_p
    L.igdc_map_open.argtypes = [C.c_void_p, C.c_int]
    L.igdc_map_close.argtypes = [C.c_void_p]
    L.igdc_search_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int,
                                   C.c_void_p, C.POINTER(C.c_int64)]
    L.igdc_enumerate_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    return L


@pytest.fixture(scope="module")
def N():
    from igd_amd import _native
    return _native


@pytest.mark.parametrize("case", range(8))
def test_host_batches_equal_the_oracle(N, case):
    """the random databases of tests/test_gpu_parity.py (sparse, dense, gType 0, multi-chunk hot tile, tile widths that are
    not powers of two or wider than 32768) x queries that are short, many-tile, inverted, zero-length, out of range or on
    unknown contigs: counts under the reference's two rules (v = 0: NEST, no filter; v > 0: FLAT with value >= v), the
    `-f` list in the reference's order, for 1, 2 and 7 threads."""
    import random
    from test_gpu_parity import CASES as DBS, _random_db, _random_queries
    d = short_tmpdir("igh")
    try:
        rng = random.Random(9100 + case)
        nbp, gtype, nfiles, nctg, span_tiles, dens, hot = DBS[case]
        path, ctgs, span = _random_db(rng, d, "h%d" % case, nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
        ichr, qs, qe = _random_queries(rng, list(range(nctg)), nbp, span, 5000)
        L = _bind(N)
        core = L.igdc_open(path.encode())
        assert core
        tsv = L.igdc_index_path(path.encode())
        assert L.igdc_load_index(core, C.cast(tsv, C.c_char_p)) == 0
        N.free(tsv)
        fd = os.open(path, os.O_RDONLY)
        m = L.igdc_map_open(core, fd)
        os.close(fd)
        assert m
        o = Oracle(path)
        nf = o.nfiles
        NOV = -2 ** 31
        for threads in ("1", "2", "7"):
            os.environ["IGD_HOST_THREADS"] = threads
            for v in (0, 1, 300, 500, 1000, 1001):
                hits = np.zeros(nf, np.int64)
                tot = C.c_int64(0)
                use_v = v > 0 and gtype == 1
                rc = L.igdc_search_host(core, m, ichr.ctypes.data, qs.ctypes.data, qe.ctypes.data, len(qs), v if use_v else NOV,
                                        1 if use_v else 0, hits.ctypes.data, C.byref(tot))
                assert rc == 0
                want, wtot = o.search(ichr, qs, qe, v)
                np.testing.assert_array_equal(hits, want, err_msg="case %d v %d threads %s" % (case, v, threads))
                assert tot.value == wtot
            qoff = np.zeros(len(qs) + 1, np.int64)
            out = C.c_void_p()
            assert L.igdc_enumerate_host(core, m, ichr.ctypes.data, qs.ctypes.data, qe.ctypes.data, len(qs), qoff.ctypes.data,
                                         C.byref(out), C.byref(tot)) == 0
            rec = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(max(1, tot.value), 4))[:tot.value].copy()
            N.free(out.value)
            wq, wr = o.enumerate(ichr, qs, qe)
            np.testing.assert_array_equal(qoff, wq)
            np.testing.assert_array_equal(rec[:, 1:], wr)
            np.testing.assert_array_equal(rec[:, 0], np.repeat(np.arange(len(qs)), np.diff(wq)))
        o.close()
        L.igdc_map_close(m)
        L.igdc_close(core)
    finally:
        os.environ.pop("IGD_HOST_THREADS", None)
        shutil.rmtree(d, ignore_errors=True)


def test_python_and_r_flavours_answer_small_files_on_the_host():
    """open = header + index only (like the reference's open_iGD); search_n / search_1 / getOverlaps on a small fixture need
    no device and return what the reference's wrapper returned (tests/golden/pywrap.json)."""
    pins = json.load(open(os.path.join(GOLDEN, "pywrap.json")))
    code = r'''
import ctypes as C, json, sys
sys.path.insert(0, %r)
import numpy as np
from igd_amd import igd_py as P, _native as N
db, q = %r, %r
h = P.igd_py(); h.open(db)
n = h.get_nFiles(); hits = np.zeros(n, np.int64)
out = {"nFiles": n, "search_n_return": int(h.search_n(q, hits)), "search_n_hits": hits.tolist(), "search_1": {}}
for key in %r:
    c, rng = key.split(":"); s, e = rng.split("-"); v = np.zeros(n, np.int64)
    h.search_1(c, int(s), int(e), v); out["search_1"][key] = v.tolist()
R = N.rabi()
r = np.zeros(n, np.int64)
a, b = C.c_char_p(db.encode()), C.c_char_p(q.encode())
R.getOverlaps(C.byref(a), C.byref(b), r.ctypes.data_as(N.i64p))
out["r"] = r.tolist()
print(json.dumps(out))
''' % (ROOT, os.path.join(GOLDEN, "smallrand", "db.igd"), os.path.join(GOLDEN, "smallrand", "q.bed"), list(pins["search_1"].keys()))
    import sys
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=_no_gpu_env())
    assert p.returncode == 0, p.stderr.decode()[-800:]
    got = json.loads(p.stdout.decode().strip().splitlines()[-1])
    r = got.pop("r")
    assert got == pins
    assert r == pins["search_n_hits"]


def test_the_limit_is_about_the_number_of_queries_not_about_the_gpu():
    """Above the limit (here: limit 100 for a 300-query file; limit 0) a host without a usable device fails loudly:
    exit code 69, the reason on stderr, no table on stdout -- batches have no CPU fallback."""
    db, q = os.path.join(GOLDEN, "smallrand", "db.igd"), os.path.join(GOLDEN, "smallrand", "q.bed")
    nlines = sum(1 for _ in open(q))
    assert nlines > 100
    for lim in ("100", "0"):
        for extra in ([], ["-v", "5"], ["-f"]):
            p = subprocess.run([EXE, "search", db, "-q", q] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                               env=_no_gpu_env(IGD_HOST_MAX_QUERIES=lim))
            assert p.returncode == 69 and b"no CPU search path" in p.stderr and b"Total" not in p.stdout, (lim, extra, p.stderr[-300:])
    p = subprocess.run([EXE, "search", db, "-q", q], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env=_no_gpu_env(IGD_HOST_MAX_QUERIES=str(nlines)))
    assert p.returncode == 0 and b"Total" in p.stdout
