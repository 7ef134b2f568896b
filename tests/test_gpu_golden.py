"""GPU: the product's front-ends against the golden fixtures (the REAL reference's stdout and the
reference's own Python wrapper's return values) and the oracle.

  - bin/igd search ... prints, byte for byte, what the reference printed (tests/golden/*/outNN.txt)
  - a C program written against include/igd_search.h the way the reference's igd.c is (it defines
    the process-wide globals itself) compiles, links with -ligd and gets the oracle's numbers
  - igd_py (the Cython wrapper's class, bound with ctypes to libigd_py.so) returns what the
    reference's wrapper returned (tests/golden/pywrap.json)
  - the R flavour's plain-C / .C entry points (libigdr.so) agree with the oracle
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

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "bin", "igd")


@pytest.mark.parametrize("case", CASES)
def test_cli_prints_what_the_reference_printed(case):
    d, dst, man = materialize(case)
    try:
        for run in man["runs"]:
            args = [os.path.join(dst, a) if a in ("db.igd", "q.bed", "q.bed.gz", "q100.bed") else a for a in run["args"]]
            p = subprocess.run([EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()[-300:]
            want = open(os.path.join(dst, run["stdout"])).read()
            assert p.stdout.decode() == want, "%s %s" % (case, run["args"])
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("case,chrom,s,e,v", [("smallrand", "chr2", 100000, 260000, 500), ("quirk", "chr1", 100, 30000, 1),
                                             ("gtype0", "chr1", 5000, 30000, 0), ("edge", "chr1", 20000, 40000, 500)])
def test_c_caller_written_like_the_reference_main(case, chrom, s, e, v):
    d = short_tmpdir("igm")
    try:
        exe = os.path.join(d, "m")
        lib = os.path.join(ROOT, "igd_amd", "lib")
        subprocess.check_call(["gcc", "-O1", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                               os.path.join(ROOT, "tests", "c", "cli_flavour_main.c"), "-L" + lib, "-ligd", "-ligd_hip",
                               "-Wl,-rpath," + lib])
        db = os.path.join(GOLDEN, case, "db.igd")
        q = os.path.join(GOLDEN, case, "q.bed")
        out = subprocess.run([exe, db, q, chrom, str(s), str(e), str(v)], stdout=subprocess.PIPE, check=True).stdout.decode()
        o = Oracle(db)
        lines = {l.split(" ret=")[0]: l for l in out.splitlines() if " ret=" in l}

        def parse(tag):
            l = lines[tag]
            ret = int(l.split("ret=")[1].split()[0])
            h = l.split("hits=")[1] if "hits=" in l else ""
            return ret, np.array([int(x) for x in h.split(",") if x], np.int64)

        want, _ = o.file_search(q, 0)
        ret, h = parse("getOverlaps")
        assert ret == 0                                     # the reference's counter is never incremented there
        np.testing.assert_array_equal(h, want)
        np.testing.assert_array_equal(parse("getOverlaps(again,accumulates)")[1], 2 * want)
        ci = o.get_id(chrom)
        one = lambda vv: o.search(np.array([ci], np.int32), np.array([s], np.int32), np.array([e], np.int32), vv)
        if o.gtype == 1:
            wv, wtot = o.file_search(q, v)
            ret, h = parse("getOverlaps_v")
            assert ret == wtot == wv.sum()
            np.testing.assert_array_equal(h, wv)
            w1, t1 = one(v)
            ret, h = parse("get_overlaps_v")
            assert ret == t1
            np.testing.assert_array_equal(h, w1)
            ret, h = parse("get_overlaps")
            assert ret == 0
            np.testing.assert_array_equal(h, one(0)[0])
            # -f prints "Query ..." + one line per overlap, returns their number (rule NEST)
            qoff, rec = o.enumerate(np.array([ci], np.int32), np.array([s], np.int32), np.array([e], np.int32))
            assert "get_overlaps_f1 ret=%d" % len(rec) in out
        else:
            np.testing.assert_array_equal(parse("getOverlaps0")[1], want)
            np.testing.assert_array_equal(parse("get_overlaps0")[1], one(0)[0])
        assert "id(%s)=%d id(nope)=-1" % (chrom, ci) in out
        assert "getOverlaps(missing file)=0" in out
        assert "parse_bed -> chr1 12 34" in out
        o.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_python_wrapper_class_returns_what_the_reference_wrapper_returned():
    from igd_amd import igd_py as iGD          # the reference's test does `import igd_py as iGD`
    pins = json.load(open(os.path.join(GOLDEN, "pywrap.json")))
    g = iGD.igd_py()
    g.open(os.path.join(GOLDEN, "smallrand", "db.igd"))
    n = g.get_nFiles()
    assert n == pins["nFiles"]
    h = np.zeros(n, dtype="int64")
    tot = g.search_n(os.path.join(GOLDEN, "smallrand", "q.bed"), h)
    assert tot == pins["search_n_return"]
    assert h.tolist() == pins["search_n_hits"]
    for key, want in pins["search_1"].items():
        c, rng = key.split(":")
        s, e = rng.split("-")
        v = np.zeros(n, dtype="int64")
        g.search_1(c, int(s), int(e), v)
        assert v.tolist() == want, key
    # a handle that was never opened can be dropped (the reference frees garbage there)
    iGD.igd_py().__del__()


def test_r_flavour_c_entry_points():
    from igd_amd import _native as N
    L = N.rabi()
    db = os.path.join(GOLDEN, "smallrand", "db.igd")
    q = os.path.join(GOLDEN, "smallrand", "q.bed")
    o = Oracle(db)
    n = o.nfiles
    # .C getOverlaps(char **igdFile, char **qFile, int64 *hits): any contig name, >= 3 fields
    h = np.zeros(n, np.int64)
    a, b = (C.c_char_p * 1)(db.encode()), (C.c_char_p * 1)(q.encode())
    L.getOverlaps(a, b, h.ctypes.data_as(N.i64p))
    np.testing.assert_array_equal(h, o.file_search(q, 0)[0])
    # .C search_1
    h1 = np.zeros(n, np.int64)
    cs, ce = C.c_int32(1000000), C.c_int32(1100000)
    L.search_1(a, (C.c_char_p * 1)(b"chr1"), C.byref(cs), C.byref(ce), h1.ctypes.data_as(N.i64p))
    want = o.search(np.array([o.get_id("chr1")], np.int32), np.array([1000000], np.int32), np.array([1100000], np.int32))[0]
    np.testing.assert_array_equal(h1, want)
    # handle + 32-bit counters (what search_1r / search_nr use)
    hnd = L.open_iGD(db.encode())
    h32 = np.zeros(n, np.int32)
    L.get_overlaps32(hnd, b"chr1", 1000000, 1100000, h32.ctypes.data_as(N.i32p))
    np.testing.assert_array_equal(h32, want)
    ichr, qs, qe = o.read_queries(q)
    names = o.ctg_names()
    arr = (C.c_char_p * len(qs))(*[names[c].encode() for c in ichr])
    h32[:] = 0
    L.igdr_search_n32(hnd, len(qs), arr, qs.ctypes.data_as(N.i32p), qe.ctypes.data_as(N.i32p), h32.ctypes.data_as(N.i32p))
    np.testing.assert_array_equal(h32, o.search(ichr, qs, qe)[0])
    L.close_iGD(hnd)
    o.close()


def test_r_call_entry_points_run_against_a_mock_of_the_r_api():
    """Every `.Call` entry point of the R flavour, compiled against the MOCK of R's C API (tests/mock_r: not R -- the image
    has no R) and called by tests/c/r_call_main.c the way IGDr.R calls them: handle object + finalizer, search_1r and
    search_nr (one GPU batch) against the plain-C get_overlaps32, the get_* accessors, get_binData's columns, and an R
    error -- not a crash -- for a call on a freed handle."""
    import shutil
    import subprocess
    from helpers import build_r_call_harness, short_tmpdir
    d = short_tmpdir("igr")
    try:
        exe = build_r_call_harness(d)
        p = subprocess.run([exe, os.path.join(GOLDEN, "smallrand", "db.igd"), "gpu"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0 and b"R-CALL-OK" in p.stdout and b"search_nr" in p.stdout, (p.stdout.decode(), p.stderr.decode()[-800:])
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("case", ["edge", "quirk", "branch", "smallrand"])
def test_cli_hitmap_file_identical_to_reference(case):
    """`igd search db.igd -m [-v N] -o file` writes the bytes the reference wrote (f3)."""
    d, dst, man = materialize(case)
    try:
        for hm in man["meta"]["hitmaps"]:
            out = os.path.join(d, "g_" + hm["file"])
            args = [os.path.join(dst, a) if a == "db.igd" else out if a == hm["file"] else a for a in hm["args"]]
            p = subprocess.run([EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()[-300:]
            assert p.stdout.decode() == open(os.path.join(dst, hm["stdout"])).read()
            assert open(out).read() == open(os.path.join(dst, hm["file"])).read(), (case, hm["args"])
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_reference_main_linked_against_libigd_prints_reference_output():
    """the reference's OWN main() (src/igd.c compiled against the reference's headers by
    oracle/Makefile, where the source lies) linked against this repository's libigd.so -- `search` in
    every mode and `create` must behave like the reference binary did (golden stdout / files)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "igd_main_on_libigd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/igd_main_on_libigd not built (needs /root/reference at build time)")
    n = 0
    for case in CASES:
        d, dst, man = materialize(case)
        try:
            for run in man["runs"]:
                if "-o" in run["args"]:
                    continue
                got = subprocess.run([exe] + run["args"], cwd=dst, stdout=subprocess.PIPE, timeout=600).stdout.decode()
                assert got == open(os.path.join(dst, run["stdout"])).read(), (case, run["args"])
                n += 1
        finally:
            shutil.rmtree(d, ignore_errors=True)
    assert n >= 30
    d = short_tmpdir()
    try:
        g = os.path.join(GOLDEN, "create")
        p = subprocess.run([exe, "create", g + "/in/", d + "/o", "db", "-b", "12"], stdout=subprocess.PIPE, timeout=600)
        assert p.stdout.decode().replace(d + "/o/", "OUT/").replace(g + "/in/", "IN/") == open(g + "/stdout.txt").read()
        from test_oracle_create import same_igd
        same_igd(d + "/o/db.igd", g + "/ref.igd")
        assert open(d + "/o/db_index.tsv", "rb").read() == open(g + "/ref_index.tsv", "rb").read()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_f_output_does_not_depend_on_the_number_of_formatting_threads():
    """`-f` text is produced by several threads (contiguous query ranges, written in order): any thread count must
    give the reference's bytes."""
    for case in ("smallrand", "edge", "gtype0"):
        d, dst, man = materialize(case)
        try:
            run = [r for r in man["runs"] if "-f" in r["args"] and "-q" in r["args"]][0]
            want = open(os.path.join(dst, run["stdout"])).read()
            for nt in ("1", "3", "16", "64"):
                got = subprocess.run([os.path.join(ROOT, "bin", "igd")] + run["args"], cwd=dst, stdout=subprocess.PIPE, timeout=600,
                                     env=dict(os.environ, IGD_PRINT_THREADS=nt)).stdout.decode()
                assert got == want, (case, nt)
        finally:
            shutil.rmtree(d, ignore_errors=True)


def test_f_output_is_the_references_through_both_record_widths():
    """`-f` moves 8 bytes per overlap over PCIe when the database's records fit them (igd_hip_enumerate_stream8, round 6) and 16
    otherwise (IGD_ENUM_HIT16=1 forces that stream): the text is the reference's recorded stdout either way, whole and in chunk
    buffers of 50 overlaps (seams inside the formatter's expansion)."""
    for case in ("smallrand", "edge", "gtype0", "quirk"):
        d, dst, man = materialize(case)
        try:
            runs = [r for r in man["runs"] if "-f" in r["args"] and "-q" in r["args"]]
            assert runs
            for run in runs:
                want = open(os.path.join(dst, run["stdout"])).read()
                for env in ({}, {"IGD_ENUM_HIT16": "1"}, {"IGD_ENUM_CHUNK_HITS": "50"}, {"IGD_ENUM_HIT16": "1", "IGD_ENUM_CHUNK_HITS": "50"}):
                    got = subprocess.run([os.path.join(ROOT, "bin", "igd")] + run["args"], cwd=dst, stdout=subprocess.PIPE, timeout=600,
                                         env=dict(os.environ, IGD_HOST_MAX_QUERIES="0", **env)).stdout.decode()
                    assert got == want, (case, env)
        finally:
            shutil.rmtree(d, ignore_errors=True)


def test_sorted_bed_with_another_chromosome_order_takes_the_merge_join():
    """`sort -k1,1 -k2,2n` orders chromosomes lexicographically, the database numbers contigs by first appearance: such a
    BED is one ordered run per contig, the runs out of contig order.  The host puts the runs into the database's order
    (igdc_queries_group_contigs; IGD_TIMING names the step) so that the batch takes the merge join; stdout equals the
    reference's on the same lines in their original order (config1's golden database and query file, re-sorted)."""
    d, dst, man = materialize("config1")
    try:
        lines = [l for l in open(os.path.join(dst, "q.bed")).read().splitlines() if l.strip()]
        key = lambda l: (l.split("\t")[0], int(l.split("\t")[1]))          # lexicographic chromosome, numeric start
        lex = sorted(lines, key=key)
        rev = sorted(lines, key=lambda l: ([-ord(c) for c in l.split("\t")[0]], int(l.split("\t")[1])))   # chromosomes in reverse order
        for name, rows in (("lex", lex), ("rev", rev)):
            qb = os.path.join(dst, name + ".bed")
            open(qb, "w").write("\n".join(rows) + "\n")
            for extra in ([], ["-v", "500"]):
                want = [r for r in man["runs"] if r["args"][:4] == ["search", "db.igd", "-q", "q.bed"] and r["args"][4:] == extra]
                p = subprocess.run([EXE, "search", os.path.join(dst, "db.igd"), "-q", qb] + extra, stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, env=dict(os.environ, IGD_TIMING="1"))
                assert p.returncode == 0, p.stderr.decode()[-300:]
                if want:                                                  # counts do not depend on the order of the lines
                    assert p.stdout.decode() == open(os.path.join(dst, want[0]["stdout"])).read(), (name, extra)
                o = subprocess.run([os.path.join(ROOT, "oracle", "_build", "igd_oracle"), "search", os.path.join(dst, "db.igd"), "-q", qb] + extra,
                                   stdout=subprocess.PIPE, check=True).stdout
                assert p.stdout == o, (name, extra)
        # the reversed order is certainly not the database's: the step must have run
        assert b"contig runs put into the database's order" in p.stderr, p.stderr.decode()[-600:]
    finally:
        shutil.rmtree(d, ignore_errors=True)
