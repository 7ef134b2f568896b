"""Host side of the product (C, no GPU needed): .igd / _index.tsv loaders, contig lookup, BED
reader, bSearch, the minimal writer, the generator -- each against the CPU oracle or the golden
fixtures -- and the C-ABI surface: every function declared in include/*.h is exported by the
library that implements it.  No search is executed here."""
import ctypes as C
import json
import os
import random
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, Oracle, orc, parse_hits_table, run_oracle_cli, short_tmpdir, write_bed
from test_golden_oracle import materialize


@pytest.fixture(scope="module")
def N():
    from igd_amd import _native
    if not os.path.exists(os.path.join(_native.LIBDIR, "libigd.so")):
        _native.build()
    return _native


@pytest.mark.parametrize("case", ["edge", "quirk", "branch", "gtype0", "smallrand"])
def test_header_loader_matches_oracle(case, N):
    L = N.cli()
    path = os.path.join(GOLDEN, case, "db.igd")
    core = L.igdc_open(path.encode())
    assert core
    tsv = L.igdc_index_path(path.encode())
    assert C.cast(tsv, C.c_char_p).value.decode() == os.path.join(GOLDEN, case, "db_index.tsv")
    assert L.igdc_load_index(core, C.cast(tsv, C.c_char_p)) == 0
    N.free(tsv)
    o = Oracle(path, preload=False)
    c = core.contents
    assert (c.nbp, c.gType, c.nCtg, c.nFiles) == (o.nbp, o.gtype, o.nctg, o.nfiles)
    lib = orc()
    nrec = 0
    for i in range(c.nCtg):
        assert c.cName[i].decode() == lib.orc_ctg_name(o.h, i).decode()
        assert c.nTile[i] == lib.orc_ntile(o.h, i)
        for j in range(c.nTile[i]):
            assert c.nCnt[i][j] == lib.orc_ncnt(o.h, i, j)
            nrec += c.nCnt[i][j]
        assert L.igdc_get_id(core, c.cName[i]) == i
    assert c.nRecords == nrec
    rec = 12 if c.gType == 0 else 16
    assert c.dataOff + rec * nrec == os.path.getsize(path)
    for i in range(c.nFiles):
        assert c.fileName[i].decode() == lib.orc_file_name(o.h, i).decode()
        assert c.fileNr[i] == lib.orc_file_nr(o.h, i)
    for bad in (b"chr", b"Chr1", b"chr1 ", b"", b"chr99"):
        assert L.igdc_get_id(core, bad) == o.get_id(bad.decode())
    L.igdc_close(core)
    o.close()


def test_query_reader_matches_oracle_and_reference_rules(N):
    """parse fixture: headers, short lines, end<=0, non-chr names, 39/40-char names, CRLF,
    junk suffixes, no final newline, gz -- same accepted (contig,start,end) list as the oracle."""
    L = N.cli()
    case = os.path.join(GOLDEN, "parse")
    core = L.igdc_open(os.path.join(case, "db.igd").encode())
    o = Oracle(os.path.join(case, "db.igd"), preload=False)
    for q in ("q.bed", "q.bed.gz"):
        want = o.read_queries(os.path.join(case, q))
        qq = N.CoreQueries()
        assert L.igdc_read_queries(core, os.path.join(case, q).encode(), 1, C.byref(qq)) == 0
        got = [np.ctypeslib.as_array(p, shape=(qq.n,)).copy() for p in (qq.ichr, qq.qs, qq.qe)]
        L.igdc_queries_free(C.byref(qq))
        assert qq.n == 0 or len(got[0]) == len(want[0])
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
        assert len(want[0]) == 7        # the 7 lines the reference accepts for known contigs
    # the Python/R rule (any contig name, end may be <= 0): strictly more lines
    qq = N.CoreQueries()
    assert L.igdc_read_queries(core, os.path.join(case, "q.bed").encode(), 0, C.byref(qq)) == 0
    assert qq.n == 10
    L.igdc_queries_free(C.byref(qq))
    L.igdc_close(core)
    o.close()


def test_parallel_ingest_equals_sequential_and_oracle(N):
    """>1 MiB plain BED goes through the multi-threaded, non-mutating reader: it must accept
    exactly the lines, in order, that the sequential reader / the oracle accept (incl. junk)."""
    L = N.cli()
    rng = random.Random(17)
    d = short_tmpdir("igq")
    try:
        case = os.path.join(GOLDEN, "smallrand")
        core = L.igdc_open(os.path.join(case, "db.igd").encode())
        o = Oracle(os.path.join(case, "db.igd"), preload=False)
        junk = ["track x", "#c", "chr1\t5", "1\t2\t3", "chr1\t7\t0", "chr2\t 12\t 99 \textra", "chrX\t+3\t9z", "",
                "chr3\t99999999999\t5", "chr1\t-4\t88\r", "chr9\t1\t2", "chr1 1 2", "chr2\t1e3\t2000\t\t",
                "chr1\t9223372036854775807\t9223372036854775808", "chr" + "a" * 37 + "\t1\t2", "c\t1\t2", "ch\t1\t2"]
        for srt in (True, False):
            rows = []
            for _ in range(120000):
                if rng.random() < 0.03:
                    rows.append(rng.choice(junk))
                else:
                    c = rng.choice(["chr1", "chr2", "chr3", "chrX"])
                    s0 = rng.randrange(0, 4000000)
                    rows.append("%s\t%d\t%d" % (c, s0, s0 + rng.randint(1, 50000)))
            if srt:
                key = lambda r: (r.split("\t")[0], int(r.split("\t")[1])) if r.count("\t") == 2 and r.split("\t")[1].isdigit() else ("", 0)
                rows.sort(key=key)
            path = os.path.join(d, "q_%d.bed" % srt)
            with open(path, "w") as f:
                f.write("\n".join(rows))                 # no trailing newline
            assert os.path.getsize(path) > (1 << 20)
            want = o.read_queries(path)
            res = {}
            for mode in ("par", "seq", "par3"):
                os.environ.pop("IGD_PARSE_SEQUENTIAL", None); os.environ.pop("IGD_PARSE_THREADS", None)
                if mode == "seq": os.environ["IGD_PARSE_SEQUENTIAL"] = "1"
                if mode == "par3": os.environ["IGD_PARSE_THREADS"] = "3"
                for rule in (1, 0):
                    qq = N.CoreQueries()
                    assert L.igdc_read_queries(core, path.encode(), rule, C.byref(qq)) == 0
                    res[(mode, rule)] = ([np.ctypeslib.as_array(p, shape=(qq.n,)).copy() for p in (qq.ichr, qq.qs, qq.qe)], qq.unsorted)
                    L.igdc_queries_free(C.byref(qq))
            os.environ.pop("IGD_PARSE_SEQUENTIAL", None); os.environ.pop("IGD_PARSE_THREADS", None)
            for rule in (1, 0):
                for mode in ("par", "par3"):
                    for g, w in zip(res[(mode, rule)][0], res[("seq", rule)][0]):
                        np.testing.assert_array_equal(g, w)
                    assert bool(res[(mode, rule)][1]) == bool(res[("seq", rule)][1])
            for g, w in zip(res[("par", 1)][0], want):
                np.testing.assert_array_equal(g, w)
            assert len(want[0]) > 100000
        L.igdc_close(core)
        o.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_parse_bed_line_fuzz(N):
    L = N.cli()
    lib = orc()
    rng = random.Random(3)
    pieces = ["chr1", "chrX", "1", "chr", "Chr2", "chr" + "a" * 36, "chr" + "a" * 37, "", " chr1", "100", "-5", "0",
              "12abc", " 77", "+9", "2147483648", "99999999999", "1e5", "x", "\r"]
    for _ in range(3000):
        n = rng.randint(0, 6)
        line = "\t".join(rng.choice(pieces) for _ in range(n)).encode()
        a, b = C.create_string_buffer(line, len(line) + 2), C.create_string_buffer(line, len(line) + 2)
        s1, e1, s2, e2 = C.c_int32(7), C.c_int32(7), C.c_int32(7), C.c_int32(7)
        r1 = L.parse_bed(a, C.byref(s1), C.byref(e1))
        r2 = lib.orc_parse_bed(b, C.byref(s2), C.byref(e2))
        assert bool(r1) == bool(r2), line
        assert (s1.value, e1.value) == (s2.value, e2.value), line
        if r1:
            assert C.cast(r1, C.c_char_p).value == C.cast(r2, C.c_char_p).value


def test_bsearch_is_last_start_below_qe(N):
    L = N.cli()
    rng = random.Random(9)
    for _ in range(300):
        n = rng.randint(1, 40)
        starts = sorted(rng.randint(0, 30) for _ in range(n))
        g = np.zeros((n, 4), np.int32)
        g[:, 1] = starts
        for qe in range(-1, 33):
            want = max([i for i in range(n) if starts[i] < qe], default=-1)
            assert L.bSearch(g.ctypes.data, 0, n - 1, qe) == want


def test_generator_queries_sorted_and_reproducible(N):
    from igd_amd import synth
    a = synth.make_queries(5000, seed=7, genome=synth.HG38)
    b = synth.make_queries(5000, seed=7, genome=synth.HG38)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    key = a[0].astype(np.int64) * (1 << 32) + a[1]
    assert (np.diff(key) >= 0).all()
    c = synth.make_queries(5000, seed=7, genome=synth.HG38, sorted_=False)
    assert sorted(zip(*[v.tolist() for v in c])) == sorted(zip(*[v.tolist() for v in a]))
    assert (a[2] - a[1] >= 100).all() and (a[2] - a[1] <= 1999).all()


# --------------------------------------------------------------------------------------------
# C ABI surface
HEADER_LIBS = {"igd_hip.h": "libigd_hip.so", "igd_search.h": "libigd.so", "igd_base.h": "libigd.so",
               "igd_py_abi.h": "libigd_py.so", "igdr_abi.h": "libigdr.so", "igd_create.h": "libigd.so"}


def _declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"#ifdef IGDR_HAVE_R.*?#endif", "", txt, flags=re.S)   # .Call group needs R headers
    txt = re.sub(r"^\s*typedef\s[^;{]*\(\s*\*[^;]*;", "", txt, flags=re.M)   # function-pointer typedefs are not functions
    names = re.findall(r"^\s*(?:[A-Za-z_][\w\s\*]*?[\s\*])([A-Za-z_]\w*)\s*\([^;{]*\)\s*;", txt, flags=re.M)
    return sorted(set(n for n in names if n not in ("defined",)))


@pytest.mark.parametrize("header", sorted(HEADER_LIBS))
def test_every_declared_function_is_exported(header, N):
    lib = os.path.join(N.LIBDIR, HEADER_LIBS[header])
    assert os.path.exists(lib), "run make"
    out = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE, check=True).stdout.decode()
    exported = set(l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] in "TBDW")
    names = _declared_functions(header)
    assert len(names) >= 5, (header, names)
    missing = [n for n in names if n not in exported]
    assert not missing, "%s declares but %s does not export: %s" % (header, HEADER_LIBS[header], missing)
    C.CDLL(lib)      # and it loads (HIP runtime resolves) without a GPU


@pytest.mark.parametrize("env", [None, "37", "0", "999999999"])
def test_lazy_binding_and_engine_agree_on_the_batch_limit(env, N):
    """igd_hip_max_batch() is answered by the host flavours' lazy binding WITHOUT mapping the engine (igd_hip_lazy.c) and
    by the engine itself (igd_hip.hip); both evaluate igd_hip_max_batch_rule() of include/igd_hip.h -- same value with and without
    the test-only IGD_HIP_MAX_BATCH (each library in a fresh process: the engine reads the variable once)."""
    code = ("import ctypes, sys; L = ctypes.CDLL(sys.argv[1]); L.igd_hip_max_batch.restype = ctypes.c_int64; print(L.igd_hip_max_batch())")
    e = dict(os.environ)
    e.pop("IGD_HIP_MAX_BATCH", None)
    if env is not None:
        e["IGD_HIP_MAX_BATCH"] = env
    vals = [int(subprocess.run([sys.executable, "-c", code, os.path.join(N.LIBDIR, lib)], stdout=subprocess.PIPE, env=e, check=True).stdout)
            for lib in ("libigd.so", "libigd_hip.so", "libigd_py.so")]
    assert len(set(vals)) == 1, vals
    assert vals[0] == (37 if env == "37" else 1 << 24)


def test_cli_globals_are_exported(N):
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(N.LIBDIR, "libigd.so")],
                         stdout=subprocess.PIPE, check=True).stdout.decode()
    syms = set(l.split()[-1] for l in out.splitlines())
    for g in ("hc", "IGD", "gData", "gData0", "preIdx", "preChr", "tile_size", "fP"):
        assert g in syms


def test_no_gpu_means_loud_failure_not_a_cpu_fallback(N):
    """product path must fail loudly without a usable HIP device (this test is for GPU-less hosts)."""
    if N.hip().igd_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    from igd_amd import Database
    from igd_amd.database import IgdError
    with pytest.raises(IgdError, match="no CPU search path"):
        Database(os.path.join(GOLDEN, "edge", "db.igd"))
    exe = os.path.join(ROOT, "bin", "igd")
    # (query files of at most IGD_HOST_MAX_QUERIES lines are the host's by design -- tests/test_hostpath.py; a batch for the
    # engine, which every file is with the limit at 0, has no CPU path)
    p = subprocess.run([exe, "search", os.path.join(GOLDEN, "edge", "db.igd"), "-q", os.path.join(GOLDEN, "edge", "q.bed")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, IGD_HOST_MAX_QUERIES="0"))
    assert p.returncode == 69 and b"no CPU search path" in p.stderr and b"Total" not in p.stdout
    # ... and the package never imports the oracle
    src = "".join(open(os.path.join(ROOT, "igd_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "igd_amd")) if f.endswith(".py"))
    assert "oracle" not in src.lower().replace("no cpu", "")


def test_libraries_never_end_the_host_process(N):
    """libigd.so / libigd_py.so / libigdr.so inside an interpreter (ctypes here, Cython / R in real life): an
    unusable engine -- no GPU on this host, or IGD_DEVICE out of range on a GPU box -- must NOT exit() the
    process: the call returns like the reference's silent failures, hits stay untouched, igd_engine_status() != 0
    (the reference returns silently at src/igd_search.c:457,462,701-702)."""
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from igd_amd import _native as N
from igd_amd import igd_py as P
db, q = %r, %r
if N.hip().igd_hip_device_count() > 0:
    os.environ["IGD_DEVICE"] = "99"
os.environ["IGD_HOST_MAX_QUERIES"] = "0"     # every batch is the engine's (small files would be the host's: tests/test_hostpath.py)
# handle flavour through the ctypes twin of the .pyx class: open reads header + index only (like the reference's); the
# batch that needs the engine raises, interpreter alive
h = P.igd_py()
h.open(db)
n = h.get_nFiles(); assert n == 10
hits = np.zeros(n, np.int64)
try:
    h.search_n(q, hits); print("NO-RAISE")
except P.IgdEngineError as e:
    print("py raised:", "no CPU search path" in str(e))
assert not hits.any()
# R flavour (.C entry points)
R = N.rabi()
assert R.igd_engine_status() == 0
hits[:] = 0
a, b = C.c_char_p(db.encode()), C.c_char_p(q.encode())
R.getOverlaps(C.byref(a), C.byref(b), hits.ctypes.data_as(N.i64p)); assert not hits.any() and R.igd_engine_status() != 0
# CLI flavour as a library
L = N.cli()
g = L.get_igdinfo(db.encode()); assert g
L.getOverlaps.argtypes = [C.c_char_p, N.i64p]; L.getOverlaps.restype = C.c_int64
assert L.getOverlaps(q.encode(), hits.ctypes.data_as(N.i64p)) == 0 and not hits.any() and L.igd_engine_status() != 0
print("alive")
''' % (ROOT, os.path.join(ROOT, "tests"), os.path.join(GOLDEN, "edge", "db.igd"), os.path.join(GOLDEN, "edge", "q.bed"))
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0 and "py raised: True" in out and out.strip().endswith("alive"), (out, p.stderr.decode()[-1500:])
    assert b"no CPU search path" in p.stderr


def test_create_without_gpu_fails_loudly_and_leaves_no_database(N):
    """`igd create` has no CPU path either: on a GPU-less host it must say so, exit non-zero and not
    leave a half-written .igd behind."""
    if N.hip().igd_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    d = short_tmpdir("igc")
    try:
        os.makedirs(d + "/in")
        write_bed(d + "/in/a.bed", [("chr1", 5, 50, "n", 3)])
        p = subprocess.run([os.path.join(ROOT, "bin", "igd"), "create", d + "/in", d + "/out", "db"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert p.returncode != 0
        assert b"no CPU path" in p.stderr
        assert not os.path.exists(d + "/out/db.igd")
    finally:
        shutil.rmtree(d, ignore_errors=True)


# --------------------------------------------------------------------------------------------
# single intervals are answered on the host from the interval's own tiles (no database upload)
def _golden_r_runs():
    out = []
    for fam in ("edge", "quirk", "gtype0"):
        man = json.load(open(os.path.join(GOLDEN, fam, "manifest.json")))
        for k, run in enumerate(man["runs"]):
            if "-r" in run["args"]:
                out.append((fam, k))
    return out


@pytest.mark.parametrize("fam,k", _golden_r_runs())
def test_single_region_runs_print_what_the_reference_printed_without_a_gpu(fam, k):
    """`igd search db -r chr s e [-v N] [-f]`: byte-identical to the reference's stdout (tests/golden), read from the
    interval's own tiles by the host (igdc_walk_one) -- this runs on the GPU-less build container."""
    man = json.load(open(os.path.join(GOLDEN, fam, "manifest.json")))
    run = man["runs"][k]
    args = [os.path.join(GOLDEN, fam, a) if a in ("db.igd", "q.bed") else a for a in run["args"]]
    p = subprocess.run([os.path.join(ROOT, "bin", "igd")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    assert p.stdout.decode() == open(os.path.join(GOLDEN, fam, run["stdout"])).read(), run["args"]
    assert b"GPU" not in p.stderr


def test_single_interval_walk_equals_oracle_on_random_intervals(N):
    """igdc_walk_one (host, product code) against the oracle on random single intervals incl. inverted, zero-length,
    multi-tile and out-of-range ones, both rules, with and without the value filter."""
    L = N.cli()
    L.igdc_walk_one.restype = C.c_int64
    L.igdc_walk_one.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_int,
                                N.i64p, C.c_void_p, C.c_void_p]
    rng = random.Random(5)
    for fam in ("smallrand", "quirk", "gtype0", "edge"):
        path = os.path.join(GOLDEN, fam, "db.igd")
        core = L.igdc_open(path.encode())
        tsv = L.igdc_index_path(path.encode())
        assert L.igdc_load_index(core, C.cast(tsv, C.c_char_p)) == 0
        N.free(tsv)
        o = Oracle(path)
        fd = os.open(path, os.O_RDONLY)
        try:
            nbp, nctg = core.contents.nbp, core.contents.nCtg
            for _ in range(400):
                c = rng.randrange(-1, nctg + 1)
                span = nbp * (core.contents.nTile[c] if 0 <= c < nctg else 4)
                qs = rng.randrange(-nbp, span + 2 * nbp)
                qe = qs + rng.choice([0, 1, 50, nbp // 3, nbp, 3 * nbp + 7, 9 * nbp, -rng.randint(1, 300)])
                for v in (0, 300):
                    want, wtot = o.search(np.array([c], np.int32), np.array([qs], np.int32), np.array([qe], np.int32), v)
                    hits = np.zeros(max(o.nfiles, 1), np.int64)
                    use_v = 1 if (v > 0 and o.gtype == 1) else 0           # the CLI dispatch (src/igd_search.c:1023-1030)
                    rule = 1 if use_v else 0
                    if qs <= -nbp:
                        continue                                           # n1 < 0: out-of-bounds read in the reference
                    got = L.igdc_walk_one(core, fd, c, qs, qe, v, use_v, rule, hits.ctypes.data_as(N.i64p), None, None)
                    assert got == wtot, (fam, c, qs, qe, v)
                    np.testing.assert_array_equal(hits[:o.nfiles], want)
        finally:
            os.close(fd)
            o.close()
            L.igdc_close(core)


def test_r_flavour_search_1_needs_no_gpu(N):
    """IGDr's .C entry point search_1(igdFile, chr, start, end, hits): one interval on a database opened for the call --
    answered from its own tiles on the host, equal to the oracle's counts (and so to the reference's, see the goldens)."""
    R = N.rabi()
    path = os.path.join(GOLDEN, "smallrand", "db.igd")
    o = Oracle(path)
    try:
        for chrom, s_, e_ in (("chr1", 1000, 90000), ("chr2", 100000, 260000), ("chr9", 5, 50), ("chr1", 70000, 70000)):
            h = np.zeros(o.nfiles + 1, np.int64)
            a, b = (C.c_char_p * 1)(path.encode()), (C.c_char_p * 1)(chrom.encode())
            cs, ce = C.c_int32(s_), C.c_int32(e_)
            R.search_1(a, b, C.byref(cs), C.byref(ce), h.ctypes.data_as(N.i64p))
            cid = o.contig_id(chrom) if hasattr(o, "contig_id") else None
            ichr = np.array([cid if cid is not None else -1], np.int32)
            if cid is None:
                import ctypes
                L = N.cli()
                core = L.igdc_open(path.encode())
                ichr[0] = L.igdc_get_id(core, chrom.encode())
                L.igdc_close(core)
            want, _ = o.search(ichr, np.array([s_], np.int32), np.array([e_], np.int32), 0)
            np.testing.assert_array_equal(h[:o.nfiles], want)
        assert R.igd_engine_status() == 0
    finally:
        o.close()


def test_r_call_entry_points_compile_against_a_mock_of_the_r_api():
    """The ten `.Call` functions of the R flavour (igdr_abi.c -DIGDR_HAVE_R: IGDr/src/igd_search.c:307-355,
    IGDr/src/igd_base.c:382-461) cannot be built against R here -- the image has none -- so they are compiled, with
    implicit declarations and pointer mismatches as errors, against a MOCK of the few R C-API names they use
    (tests/mock_r, plainly labelled: not R) and linked with a harness that calls every one of them
    (tests/c/r_call_main.c; with the engine on the GPU box: tests/test_gpu_golden.py).  Here, without a GPU: iGD_new reads
    header + index only (like the reference's), the small search_nr batch is the host's (igd_hostpath.c), and every check of
    the harness passes; with the host path off (limit 0) the batch needs the engine and search_nr raises an R error that
    names the reason -- not a crash, not an exit."""
    from helpers import build_r_call_harness
    d = short_tmpdir("igr")
    try:
        exe = build_r_call_harness(d)
        db = os.path.join(GOLDEN, "smallrand", "db.igd")
        env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
        env.pop("IGD_HOST_MAX_QUERIES", None)
        p = subprocess.run([exe, db, "gpu"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
        assert p.returncode == 0 and b"R-CALL-OK" in p.stdout and b"search_nr" in p.stdout, (p.returncode, p.stdout.decode(), p.stderr.decode()[-800:])
        p = subprocess.run([exe, db, "gpu"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=dict(env, IGD_HOST_MAX_QUERIES="0"))
        assert p.returncode == 3 and b"mock Rf_error outside a guarded call" in p.stderr and b"GPU engine" in p.stderr, \
            (p.returncode, p.stdout.decode(), p.stderr.decode()[-800:])
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_seq_overlaps_one_interval_on_the_host_equals_the_oracle():
    """seq_overlaps (src/igd_search.h:19, src/igd_search.c:253-352): the per-query helper of Seqpare appends one interval's
    overlaps -- (first tile of the query, index in tile, dataset, single-precision similarity) in the reference's order --
    to the caller's overlaps_t, growing it by the reference's EXPAND rule.  One interval: host path, no GPU.  In a child
    process (the CLI flavour keeps process-wide state)."""
    code = r"""
import sys, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from helpers import Oracle, orc
from igd_amd import _native as N
db = %r
L = N.cli()
assert L.get_igdinfo(db.encode())
class Ov(C.Structure):
    _fields_ = [("nn", C.c_int32), ("mm", C.c_int32), ("olist", C.c_void_p)]
L.seq_overlaps.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(Ov)]
L.seq_overlaps.restype = None
o = Oracle(db)
lib = orc()
lib.orc_seq_overlaps.restype = C.c_int64
lib.orc_seq_overlaps.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int64]
import random
rng = random.Random(5)
ov = Ov(0, 0, None)
want_all = []
names = o.ctg_names() + ["chrNope"]
nbp = o.nbp
for k in range(400):
    c = rng.choice(names).encode()
    s = rng.randrange(0, 40 * nbp)
    e = s + rng.choice([0, 1, 50, nbp // 3, nbp, 3 * nbp + 7, -20])
    buf = (C.c_int32 * (4 * 100000))()
    n = lib.orc_seq_overlaps(o.h, c, s, e, buf, 100000)
    want_all += list(buf[: 4 * n])
    L.seq_overlaps(c, s, e, C.byref(ov))
got = np.ctypeslib.as_array(C.cast(ov.olist, C.POINTER(C.c_int32)), shape=(4 * ov.nn,)) if ov.nn else np.zeros(0, np.int32)
assert ov.nn * 4 == len(want_all) and ov.nn > 500, (ov.nn, len(want_all))
assert np.array_equal(got, np.array(want_all, np.int32)), "entries differ"
m = 0
for _ in range(10**6):                     # capacity by the reference's EXPAND rule (src/igd_base.h:262-265)
    if m >= ov.nn and m >= 16: break
    m = m + (2 + m // 8) if m else 16
assert ov.mm >= ov.nn
print("seq-ok", ov.nn)
""" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(GOLDEN, "smallrand", "db.igd"))
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0 and b"seq-ok" in p.stdout, (p.stdout.decode(), p.stderr.decode()[-1500:])


def test_contig_runs_of_a_sorted_bed_are_put_into_the_databases_order(N):
    """igdc_queries_group_contigs: a BED sorted by (chromosome, start) whose chromosomes come in another order than the
    database numbers them is ONE ordered run per contig -- the runs are permuted into contig order (same multiset of
    queries, now ordered by (contig, start): the merge join's input); anything else -- a contig in two runs, a start out
    of order -- is left exactly as it was."""
    L = N.cli()

    class Q(C.Structure):
        _fields_ = [("n", C.c_int64), ("cap", C.c_int64), ("ichr", N.i32p), ("qs", N.i32p), ("qe", N.i32p), ("unsorted", C.c_int32),
                    ("max_len", C.c_int32)]   # igd_core.h
    L.igdc_queries_push.argtypes = [C.POINTER(Q), C.c_int32, C.c_int32, C.c_int32]
    L.igdc_queries_group_contigs.argtypes = [C.POINTER(Q), C.c_int32]
    L.igdc_queries_group_contigs.restype = C.c_int
    L.igdc_queries_free.argtypes = [C.POINTER(Q)]
    rng = random.Random(11)

    def build(rows):
        q = Q()
        for c, s, e in rows:
            assert L.igdc_queries_push(C.byref(q), c, s, e) == 0
        return q

    def rows_of(q):
        return [(q.ichr[i], q.qs[i], q.qe[i]) for i in range(q.n)]

    runs = {c: sorted((rng.randrange(0, 10**6), rng.randrange(1, 5000)) for _ in range(rng.randrange(1, 40))) for c in range(6)}
    order = [3, 0, 5, 1, 4, 2]                                       # e.g. chr1 chr10 chr11 chr2 ... against natural numbering
    rows = [(c, s, s + l) for c in order for s, l in runs[c]]
    q = build(rows)
    assert q.unsorted == 1
    assert L.igdc_queries_group_contigs(C.byref(q), 6) == 1 and q.unsorted == 0
    got = rows_of(q)
    assert got == [(c, s, s + l) for c in range(6) for s, l in runs[c]] and sorted(got) == sorted(rows)
    L.igdc_queries_free(C.byref(q))
    # a contig in two runs / a start out of order inside a run / already ordered: untouched
    for bad in (rows + [(3, 7, 9)], [(1, 50, 60), (1, 40, 70), (0, 1, 2)], [(0, 1, 2), (1, 1, 2)]):
        q = build(bad)
        before, flag = rows_of(q), q.unsorted
        assert L.igdc_queries_group_contigs(C.byref(q), 6) == 0 and rows_of(q) == before and q.unsorted == flag
        L.igdc_queries_free(C.byref(q))


def test_a_parsed_query_set_earns_its_engine_flags(N):
    """igdc_queries_flags: IGD_HIP_FLAG_SORTED (1) when every pushed query was >= the one before by (contig, start), and
    IGD_HIP_FLAG_SHORT (16) on top when no query is as long as a tile -- what lets a dense file take the engine's DIRECT step.
    Both are statements the device verifies; here: that the reader makes them exactly when they hold, also across the
    seams of the threaded parser.  A set the reader saw out of order earns IGD_HIP_FLAG_BUCKET (2): no order check on the device."""
    L = N.cli()
    Q = N.CoreQueries
    L.igdc_queries_push.argtypes = [C.POINTER(Q), C.c_int32, C.c_int32, C.c_int32]
    L.igdc_queries_flags.argtypes = [C.POINTER(Q), C.c_int32]
    L.igdc_queries_flags.restype = C.c_int
    L.igdc_queries_free.argtypes = [C.POINTER(Q)]
    nbp = 16384
    for rows, want in (([(0, 10, 500), (0, 10, 16393), (1, 5, 6)], 1 | 16),          # longest = 16383 < nbp
                       ([(0, 10, 500), (0, 10, 16394), (1, 5, 6)], 1),               # one query a tile long
                       ([(0, 10, 500), (0, 9, 20)], 2),                              # out of order: IGD_HIP_FLAG_BUCKET (the device need not check again)
                       ([(0, 100, 90), (0, 100, 100)], 1 | 16),                      # inverted / zero-length: "short"
                       ([], 1 | 16)):
        q = Q()
        for c, s, e in rows:
            assert L.igdc_queries_push(C.byref(q), c, s, e) == 0
        assert L.igdc_queries_flags(C.byref(q), nbp) == want, rows
        L.igdc_queries_free(C.byref(q))
    # through the reader (threaded for files above 1 MiB): the longest query sits in the middle of a large file
    d = short_tmpdir("igf")
    try:
        from igd_amd import synth
        p = os.path.join(d, "s.igd")
        synth.make_db(p, files=3, per_file=50, seed=2, genome=synth.SMALL)
        ichr, qs, qe = synth.make_queries(80000, seed=3, genome=synth.SMALL, min_len=10, max_len=900, sorted_=True)
        db = L.igdc_open(p.encode())
        assert db
        for longest, want in ((None, 1 | 16), (16384, 1), (16383, 1 | 16)):
            a, b, c2 = ichr.copy(), qs.copy(), qe.copy()
            if longest is not None:
                c2[40000] = b[40000] + longest
            bed = os.path.join(d, "q.bed")
            synth.write_bed(bed, synth.SMALL, a, b, c2)
            assert os.path.getsize(bed) > (1 << 20)
            q = Q()
            assert L.igdc_read_queries(db, bed.encode(), 1, C.byref(q)) == 0 and q.n == 80000
            assert L.igdc_queries_flags(C.byref(q), nbp) == want, longest
            L.igdc_queries_free(C.byref(q))
        L.igdc_close(db)
    finally:
        shutil.rmtree(d, ignore_errors=True)
