"""Capacity edges of the merge join (igd_scan_sorted) that ordinary batches never reach:
  * more heavy tiles than the bucket path's list holds (4096): the lean build hands every tile with more than 512
    first-tile queries to heavy_sorted_body -- its list must hold every tile a batch can produce;
  * more dataset files than fit the workgroup's LDS counters (> 15360): the kernels then add to hits[] with global
    atomics and the full build's rank-method areas alone exceed the 64 KiB a launch gets without opting in.
Counts bit-exact against the oracle."""
import os
import shutil

import numpy as np
import pytest

from helpers import Oracle, short_tmpdir

pytestmark = pytest.mark.gpu
PATH = "/tmp/igdb/rm1900x26316.igd"


@pytest.fixture(scope="module")
def workdir():
    d = short_tmpdir("igl")
    yield d
    shutil.rmtree(d, ignore_errors=True)


def _roadmap():
    from igd_amd import synth
    if not (os.path.exists(PATH) and os.path.exists(PATH + ".done")):
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        synth.make_db(PATH, files=1900, per_file=26316, seed=1000, nbp_log=14, genome=synth.HG38)
        open(PATH + ".done", "w").write("ok")
    return PATH


@pytest.mark.parametrize("build", ["lean", "full"])
def test_more_dense_tiles_than_the_bucket_list_holds(build, monkeypatch):
    """5000 tiles x 600 queries, position-sorted: under the lean build every one of them is listed for the skew valve."""
    from igd_amd import Database
    monkeypatch.setenv("IGD_HIP_RANK", "0" if build == "lean" else "1")
    path = _roadmap()
    rng = np.random.default_rng(77)
    tiles, per = 5000, 600
    first = rng.choice(15000, size=tiles, replace=False)            # tiles of chr1 (248 Mbp = 15 000 tiles of 16 384 bp)
    qs = (np.repeat(first, per).astype(np.int64) * 16384 + rng.integers(0, 16384, tiles * per)).astype(np.int32)
    qe = (qs + rng.integers(1, 2500, tiles * per)).astype(np.int32)
    qe[::997] = qs[::997] - 3                                       # some inverted queries
    ichr = np.zeros(tiles * per, np.int32)
    order = np.argsort(qs, kind="stable")
    ichr, qs, qe = ichr[order], qs[order], qe[order]
    db, orc = Database(path), Oracle(path)
    try:
        want, wtot = orc.search(ichr, qs, qe, 0)
        for flags in (1, 0):
            got, gtot = db.search(ichr, qs, qe, 0, flags=flags)
            assert gtot == wtot, (build, flags)
            np.testing.assert_array_equal(got, want, err_msg="%s flags=%d" % (build, flags))
        # ... and the next, ordinary batch on the same handle is not disturbed by what the list held
        a, b, c = ichr[::50], qs[::50], qe[::50]
        np.testing.assert_array_equal(db.search(a, b, c, 0, flags=1)[0], orc.search(a, b, c, 0)[0])
    finally:
        db.close(); orc.close()


@pytest.mark.parametrize("build", ["auto", "lean", "full", "lean-global-atomics", "full-global-atomics"])
def test_more_files_than_lds_counters(build, workdir, monkeypatch):
    """16 000 files: two windows of 8000 files, one pass of scan + reduction each (LDS counters); with IGD_HIP_NO_WINDOWS
    the per-record global atomics that windows replaced (still what > 163 840 files get)."""
    from igd_amd import Database, synth
    if build.endswith("-global-atomics"):
        monkeypatch.setenv("IGD_HIP_NO_WINDOWS", "1")
        build = build.split("-")[0]
    if build != "auto":
        monkeypatch.setenv("IGD_HIP_RANK", "0" if build == "lean" else "1")
    path = os.path.join(workdir, "manyfiles.igd")
    if not os.path.exists(path):
        synth.make_db(path, files=16000, per_file=40, seed=3, nbp_log=12, genome=synth.SMALL)
    db, orc = Database(path), Oracle(path)
    try:
        assert db.nfiles == 16000
        for n in (3000, 90000):                                     # sparse (pairwise) and dense (rank method in the full build)
            ichr, qs, qe = synth.make_queries(n, seed=5, genome=synth.SMALL, min_len=1, max_len=9000, sorted_=True,
                                              unknown_every=211)
            for v in (0, 500):
                want, wtot = orc.search(ichr, qs, qe, v)
                for flags in (1, 0, 2):
                    got, gtot = db.search(ichr, qs, qe, v, flags=flags)
                    assert gtot == wtot, (build, n, v, flags)
                    np.testing.assert_array_equal(got, want, err_msg="%s n=%d v=%d flags=%d" % (build, n, v, flags))
    finally:
        db.close(); orc.close()


def test_big_image_addressing_on_an_ordinary_database(monkeypatch, workdir):
    """An image of 2^30 tile records or more (17 GB of .igd) is beyond a 32-bit byte offset: igd_scan_sorted<.., BIG = true, ..>
    gives every unit its own 64-bit base, and the batch's last launch runs the BIG twin of the skew valve.  IGD_HIP_BIG=1
    (read at open) runs an ordinary database through those instantiations: sparse, dense (rank method), -v, long
    queries, a hot tile -- against the oracle's counts."""
    import json
    from igd_amd import Database, synth
    monkeypatch.setenv("IGD_HIP_BIG", "1")
    path = _roadmap()
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bench_checksums.json")))["workloads"]
    db = Database(path)
    try:
        w = np.arange(1, db.nfiles + 1, dtype=np.uint64)
        chk = lambda h: int((h.astype(np.uint64) * w).sum() & np.uint64((1 << 63) - 1))
        base = synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=True)
        longq = synth.make_queries(100000, seed=7, genome=synth.HG38, min_len=100000, max_len=200000, sorted_=True)
        for name, q in (("config2_sorted_q1000000", base), ("long_sorted_q100000", longq)):
            for v in (0, 500):
                g = gold["%s_v%d" % (name, v)]
                for flags in (1, 0):
                    h, tot = db.search(*q, v, flags=flags)
                    assert (tot, chk(h)) == (g["total"], g["checksum"]), (name, v, flags)
        # dense (rank method) + a hot tile, whole against the oracle
        rng = np.random.default_rng(3)
        hot = (50_003_968 + rng.integers(0, 16384, 120000)).astype(np.int64)
        near = (50_003_968 + rng.integers(-20 * 16384, 20 * 16384, 180000)).astype(np.int64)
        qs = np.sort(np.concatenate([hot, near])).astype(np.int32)
        qe = (qs + rng.integers(1, 30000, len(qs))).astype(np.int32)
        ichr = np.zeros(len(qs), np.int32)
        orc = Oracle(path)
        try:
            for v in (0, 500):
                want, wtot = orc.search(ichr, qs, qe, v)
                got, gtot = db.search(ichr, qs, qe, v, flags=1)
                assert gtot == wtot
                np.testing.assert_array_equal(got, want)
        finally:
            orc.close()
    finally:
        db.close()


@pytest.mark.parametrize("files", [1, 2, 8])
def test_databases_of_very_few_files(files, workdir, monkeypatch):
    """One file: every lane of a wave names the same LDS counter (the wave adds once: build FEW = 1); up to eight: lanes
    without hits stay out (FEW = 2).  Sparse and dense batches, both builds of the merge join, -v, against the oracle."""
    from igd_amd import Database, synth
    path = os.path.join(workdir, "few%d.igd" % files)
    synth.make_db(path, files=files, per_file=60000 // files, seed=4 + files, nbp_log=12, genome=synth.SMALL)
    orc = Oracle(path)
    try:
        for build in ("0", "1"):
            monkeypatch.setenv("IGD_HIP_RANK", build)
            db = Database(path)
            try:
                for n in (2000, 150000):
                    ichr, qs, qe = synth.make_queries(n, seed=9, genome=synth.SMALL, min_len=1, max_len=12000, sorted_=True, unknown_every=301)
                    for v in (0, 400):
                        want, wtot = orc.search(ichr, qs, qe, v)
                        for flags in (1, 0, 2):
                            got, gtot = db.search(ichr, qs, qe, v, flags=flags)
                            assert gtot == wtot, (files, build, n, v, flags)
                            np.testing.assert_array_equal(got, want, err_msg="files=%d build=%s n=%d v=%d flags=%d" % (files, build, n, v, flags))
            finally:
                db.close()
    finally:
        orc.close()


def test_search_on_a_database_of_3000_contigs(workdir):
    """More contigs than k_query_bounds keeps in its LDS tables (1024): the build that gathers the per-contig tables
    from global memory (FAST = false), every mode, against the oracle.  The database: 10 files of intervals on
    chrUn_0000 .. chrUn_2999, made by the product's `create`."""
    import subprocess
    from helpers import ROOT
    from igd_amd import Database
    from test_oracle_create import write_odd_inputs
    d = os.path.join(workdir, "manyctg")
    os.makedirs(d, exist_ok=True)
    write_odd_inputs(d, "many_contigs")
    p = subprocess.run([os.path.join(ROOT, "bin", "igd"), "create", d + "/in/", d + "/o", "db", "-b", "13"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    path = d + "/o/db.igd"
    db, orc = Database(path), Oracle(path)
    try:
        assert db.nctg > 1024
        rng = np.random.default_rng(8)
        for n in (3000, 200000):
            ichr = rng.integers(-1, db.nctg + 2, n).astype(np.int32)
            qs = rng.integers(-100, 260000, n).astype(np.int32)
            qe = (qs + rng.integers(-5, 90000, n)).astype(np.int32)
            o = np.lexsort((qs, ichr))
            for q, modes in (((ichr[o], qs[o], qe[o]), (1, 0, 2)), ((ichr, qs, qe), (0, 2))):
                for v in (0, 350):
                    want, wtot = orc.search(*q, v)
                    for flags in modes:
                        got, gtot = db.search(*q, v, flags=flags)
                        assert gtot == wtot, (n, v, flags)
                        np.testing.assert_array_equal(got, want, err_msg="n=%d v=%d flags=%d" % (n, v, flags))
    finally:
        db.close(); orc.close()


@pytest.mark.parametrize("files,per_file", [(40000, 30), (70000, 12)])
def test_windows_of_files(files, per_file, workdir, monkeypatch):
    """40 000 files: three windows over the compact image; 70 000: five windows over the exact arrays (the compact image
    holds 16-bit file numbers).  Short, long (coverage arrays, exact walks: the batch's tail adds to the caller's hits[],
    not to a window of it) and piled-up queries, every mode and build, -v, against the oracle."""
    from igd_amd import Database, synth
    path = os.path.join(workdir, "win%d.igd" % files)
    synth.make_db(path, files=files, per_file=per_file, seed=11, nbp_log=12, genome=synth.SMALL)
    orc = Oracle(path)
    try:
        rng = np.random.default_rng(files)
        ichr, qs, qe = synth.make_queries(60000, seed=6, genome=synth.SMALL, min_len=1, max_len=9000, sorted_=True, unknown_every=173)
        qe[::50] = qs[::50] + rng.integers(5 * 4096, 200 * 4096, len(qs[::50]))          # long ones
        hot = 30000
        qs[hot:hot + 9000] = qs[hot] + rng.integers(0, 4096, 9000)                        # a hot tile
        qe[hot:hot + 9000] = qs[hot:hot + 9000] + rng.integers(1, 9000, 9000)
        o = np.lexsort((qs, ichr))
        srt = (ichr[o], qs[o], qe[o])
        for build in ("0", "1"):
            monkeypatch.setenv("IGD_HIP_RANK", build)
            db = Database(path)
            try:
                assert db.nfiles == files
                for v in (0, 600):
                    want, wtot = orc.search(ichr, qs, qe, v)
                    for q, flags in ((srt, 1), (srt, 0), ((ichr, qs, qe), 0), ((ichr, qs, qe), 2)):
                        got, gtot = db.search(*q, v, flags=flags)
                        assert gtot == wtot, (files, build, v, flags)
                        np.testing.assert_array_equal(got, want, err_msg="files=%d build=%s v=%d flags=%d" % (files, build, v, flags))
            finally:
                db.close()
    finally:
        orc.close()
