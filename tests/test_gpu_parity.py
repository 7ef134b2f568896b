"""GPU parity: the HIP engine (through the C ABI of include/igd_hip.h and the host flavours)
against the CPU oracle on the same inputs.  Integer work: every comparison is bit-exact.

Databases here are written by the INDEPENDENT numpy writer of tests/helpers.py (not by the
product's writer) unless a test says otherwise."""
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

from helpers import ROOT, Oracle, run_oracle_cli, short_tmpdir, write_bed, write_igd_numpy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir():
    d = short_tmpdir("igp")
    yield d
    shutil.rmtree(d, ignore_errors=True)


def _random_db(rng, d, name, nbp, gtype, nfiles, nctg, span_tiles, dens, hot=0):
    """files -> list of (chrom,start,end,value); returns path"""
    ctgs = ["chr%d" % (i + 1) for i in range(nctg)]
    span = nbp * span_tiles
    files = []
    for f in range(nfiles):
        rows = []
        for _ in range(dens):
            c = rng.choice(ctgs)
            if rng.random() < 0.5:
                L = rng.choice([1, 5, nbp // 3, nbp, 3 * nbp + 7, 6 * nbp + 1])
            else:
                L = rng.randint(1, 2 * nbp)
            s = rng.randrange(0, span)
            if rng.random() < 0.2:
                s = (s // nbp) * nbp
            rows.append((c, s, s + L, rng.randint(0, 1000)))
        for _ in range(hot):  # a very dense tile: > 512 records -> several chunks per tile
            s = 5 * nbp + rng.randrange(0, nbp)
            rows.append((ctgs[0], s, s + rng.randint(1, nbp // 2), rng.randint(0, 1000)))
        files.append(rows)
    path = os.path.join(d, name + ".igd")
    write_igd_numpy(path, files, nbp=nbp, gtype=gtype)
    return path, ctgs, span


def _random_queries(rng, ctg_ids, nbp, span, n):
    ichr = np.array([rng.choice(ctg_ids + [-1, 99]) for _ in range(n)], np.int32)
    qs = np.array([rng.randrange(0, span + 3 * nbp) for _ in range(n)], np.int32)
    ln = np.array([rng.choice([0, 1, nbp, 5 * nbp, 9 * nbp + 3, rng.randint(1, 3 * nbp), -rng.randint(1, 50)])
                   for _ in range(n)], np.int32)
    return ichr, qs, qs + ln


CASES = [
    # nbp, gtype, nfiles, nctg, span_tiles, dens, hot
    (1 << 11, 1, 5, 2, 8, 3, 0),        # sparse: many empty tiles -> rule NEST vs FLAT differ
    (1 << 11, 1, 12, 1, 3, 200, 0),     # dense small
    (1 << 12, 0, 7, 3, 40, 20, 0),      # gType 0
    (1 << 14, 1, 9, 2, 8, 40, 150),     # hot tile with > 512 records (multi-chunk)
    (1 << 12, 1, 3, 1, 40, 1, 0),       # nearly empty
    (1 << 11, 1, 33, 3, 20, 60, 40),
    (3000, 1, 8, 2, 15, 40, 0),         # tile width that is not a power of two (division path)
    (40000, 1, 6, 2, 6, 60, 0),         # wider than the compact image allows (exact arrays) and not a power of two
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_counts_match_oracle(case, workdir):
    from igd_amd import Database
    rng = random.Random(4242 + case)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[case]
    path, ctgs, span = _random_db(rng, workdir, "c%d" % case, nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc = Oracle(path)
    db = Database(path)
    try:
        assert (db.nfiles, db.nctg, db.nbp, db.gtype) == (orc.nfiles, orc.nctg, orc.nbp, orc.gtype)
        ichr, qs, qe = _random_queries(rng, list(range(nctg)), nbp, span, 3000)
        for v in (0, 1, 300, 500, 1000, 1001):
            want, wtot = orc.search(ichr, qs, qe, v)
            got, gtot = db.search(ichr, qs, qe, v)
            assert gtot == wtot, (case, v)
            np.testing.assert_array_equal(got, want, err_msg="case %d v %d" % (case, v))
        # the two rules differ exactly where the oracle says they do (empty first tile)
        if gtype == 1:
            nest, _ = db.search(ichr, qs, qe, 0)
            flat, _ = db.search(ichr, qs, qe, 1)     # v=1 keeps value>=1: compare against oracle only
            w_flat, _ = orc.search(ichr, qs, qe, 1)
            np.testing.assert_array_equal(flat, w_flat)
            assert nest.sum() <= orc.search(ichr, qs, qe, 0)[0].sum()
    finally:
        db.close()
        orc.close()


@pytest.mark.parametrize("case", [0, 1, 3, 5])
def test_enumerate_matches_oracle(case, workdir):
    from igd_amd import Database
    rng = random.Random(777 + case)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[case]
    path, ctgs, span = _random_db(rng, workdir, "e%d" % case, nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc = Oracle(path)
    db = Database(path)
    try:
        ichr, qs, qe = _random_queries(rng, list(range(nctg)), nbp, span, 1500)
        wqoff, wrec = orc.enumerate(ichr, qs, qe)
        gqoff, grec = db.enumerate(ichr, qs, qe)
        np.testing.assert_array_equal(gqoff, wqoff)
        np.testing.assert_array_equal(grec[:, 1:], wrec)          # idx,start,end in reference order
        # q column = owning query
        owner = np.repeat(np.arange(len(qs)), np.diff(wqoff))
        np.testing.assert_array_equal(grec[:, 0], owner)
    finally:
        db.close()
        orc.close()


@pytest.mark.parametrize("chunk_hits", [64, 1000, 50000])
def test_streamed_enumeration_in_small_chunks_matches_oracle(chunk_hits, workdir, monkeypatch):
    """igd_hip_enumerate_stream with chunk buffers far smaller than the result (IGD_ENUM_CHUNK_HITS): the chunks
    tile the query range in order, a query is never split (a query larger than the buffer makes the engine grow
    it), zero-overlap queries are covered too, and the concatenation is the oracle's enumeration; the one-array
    API on the same handle agrees."""
    from igd_amd import Database
    monkeypatch.setenv("IGD_ENUM_CHUNK_HITS", str(chunk_hits))
    rng = random.Random(4242 + chunk_hits)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[3]
    path, ctgs, span = _random_db(rng, workdir, "es%d" % chunk_hits, nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc = Oracle(path)
    db = Database(path)
    try:
        ichr, qs, qe = _random_queries(rng, list(range(nctg)) + [-1], nbp, span, 3000)
        wqoff, wrec = orc.enumerate(ichr, qs, qe)
        parts, ranges = [], []

        def on_chunk(q0, q1, qoff, rec):
            ranges.append((q0, q1))
            parts.append(rec.copy())
            assert len(rec) == wqoff[q1] - wqoff[q0]

        gqoff, total = db.enumerate_stream(ichr, qs, qe, on_chunk)
        np.testing.assert_array_equal(gqoff, wqoff)
        assert total == wqoff[-1]
        assert ranges[0][0] == 0 and ranges[-1][1] == len(qs) and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert len(ranges) > 3
        grec = np.concatenate(parts) if parts else np.zeros((0, 4), np.int32)
        np.testing.assert_array_equal(grec[:, 1:], wrec)
        np.testing.assert_array_equal(grec[:, 0], np.repeat(np.arange(len(qs)), np.diff(wqoff)))
        q2, r2 = db.enumerate(ichr, qs, qe)
        np.testing.assert_array_equal(q2, wqoff)
        np.testing.assert_array_equal(r2, grec)
    finally:
        db.close()
        orc.close()


@pytest.mark.parametrize("case,chunk_hits", [(0, 64), (1, 1000), (2, 50000), (3, 1000), (5, 0)])
def test_packed_stream_of_8_bytes_per_overlap_matches_oracle(case, chunk_hits, workdir, monkeypatch):
    """igd_hip_enumerate_stream8 (round 6: `-f` moves 8 instead of 16 bytes per overlap over PCIe): the packed records --
    start | (end - start) << bits | idx, bits = ceil(log2(nFiles)) per database -- expand to exactly the oracle's enumeration
    (src/igd_search.c:575-579,608-612 order), chunk seams and zero-overlap queries included, on gType 0 and 1 databases."""
    from igd_amd import Database
    if chunk_hits:
        monkeypatch.setenv("IGD_ENUM_CHUNK_HITS", str(chunk_hits))
    rng = random.Random(99 + case)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[case]
    path, ctgs, span = _random_db(rng, workdir, "p8_%d" % case, nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc, db = Oracle(path), Database(path)
    try:
        bits = db.hit8_idx_bits()
        assert bits >= 0 and (1 << bits) >= nfiles and (bits == 0 or (1 << (bits - 1)) < nfiles)
        ichr, qs, qe = _random_queries(rng, list(range(nctg)) + [-1], nbp, span, 3000)
        wqoff, wrec = orc.enumerate(ichr, qs, qe)
        parts, ranges = [], []

        def on_chunk(q0, q1, qoff, rec, b):
            assert b == bits and len(rec) == wqoff[q1] - wqoff[q0]
            ranges.append((q0, q1))
            parts.append(rec.copy())

        gqoff, total = db.enumerate_stream8(ichr, qs, qe, on_chunk)
        np.testing.assert_array_equal(gqoff, wqoff)
        assert total == wqoff[-1]
        assert ranges[0][0] == 0 and ranges[-1][1] == len(qs) and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        if chunk_hits and chunk_hits <= 1000:
            assert len(ranges) > 3
        rec = np.concatenate(parts) if parts else np.zeros((0, 2), np.uint32)
        st, en, ix = Database.expand_hit8(rec, bits)
        np.testing.assert_array_equal(np.stack([ix, st, en], axis=1), wrec)           # idx, start, end in reference order
        # the 16-byte stream on the same handle (shared chunk buffers) still agrees
        q2, r2 = db.enumerate(ichr, qs, qe)
        np.testing.assert_array_equal(r2[:, 1:], wrec)
    finally:
        db.close(); orc.close()


def test_packed_stream_refuses_a_database_whose_records_do_not_fit(workdir):
    """A record longer than the length field holds (2^(32 - bits) bp) or with end < start: igd_hip_hit8_idx_bits() = -1, the
    packed call is an argument error, and the command line tool streams 16-byte records -- same text as the oracle's."""
    import subprocess
    from helpers import ORACLE_BIN
    from igd_amd import Database
    from igd_amd.database import IgdError
    nbp = 1 << 14
    # 70 000 files (17 bits of idx -> 15 bits of length) and one interval of 40 000 bp
    files = [[] for _ in range(70000)]
    files[5] = [("chr1", 100, 40100, 1)]
    files[7] = [("chr1", 16000, 17000, 3)]
    files[69999] = [("chr1", 200, 900, 2)]
    path = os.path.join(workdir, "p8big.igd")
    write_igd_numpy(path, files, nbp=nbp, gtype=1)
    db = Database(path)
    try:
        assert db.hit8_idx_bits() == -1
        with pytest.raises(IgdError):
            db.enumerate_stream8(np.zeros(2, np.int32), np.array([0, 150], np.int32), np.array([50000, 300], np.int32))
        qoff, rec = db.enumerate(np.zeros(2, np.int32), np.array([0, 150], np.int32), np.array([50000, 300], np.int32))
        assert qoff[-1] == len(rec) >= 3
    finally:
        db.close()
    qb = os.path.join(workdir, "p8big_q.bed")
    open(qb, "w").write("chr1\t0\t50000\nchr1\t150\t300\n")
    env = dict(os.environ, IGD_HOST_MAX_QUERIES="0")
    a = subprocess.run([os.path.join(ROOT, "bin", "igd"), "search", path, "-q", qb, "-f"], stdout=subprocess.PIPE, env=env, check=True).stdout
    b = subprocess.run([ORACLE_BIN, "search", path, "-q", qb, "-f"], stdout=subprocess.PIPE, check=True).stdout
    assert a == b and b.count(b"\n") >= 5


def test_accumulates_into_caller_hits(workdir):
    """hits is caller-zeroed and ADDED to (src/igd_search.c:491): two calls sum up."""
    from igd_amd import Database
    rng = random.Random(5)
    path, ctgs, span = _random_db(rng, workdir, "acc", 1 << 12, 1, 6, 2, 10, 50)
    db = Database(path)
    orc = Oracle(path)
    try:
        ichr, qs, qe = _random_queries(rng, [0, 1], 1 << 12, span, 500)
        h = np.zeros(db.nfiles, np.int64)
        db.search(ichr, qs, qe, hits=h)
        db.search(ichr, qs, qe, hits=h)
        np.testing.assert_array_equal(h, 2 * orc.search(ichr, qs, qe)[0])
    finally:
        db.close()
        orc.close()


def test_cli_text_identical_to_oracle_cli(workdir):
    """bin/igd search -q / -v / -f / -r prints exactly what the reference prints (the oracle CLI
    is pinned to the reference's stdout by tests/test_oracle_vs_ref.py and tests/golden)."""
    rng = random.Random(99)
    nbp = 1 << 12
    path, ctgs, span = _random_db(rng, workdir, "cli", nbp, 1, 11, 3, 12, 30)
    rows = []
    for _ in range(400):
        c = rng.choice(ctgs + ["chr7", "2"])
        s = rng.randrange(0, span + 2 * nbp)
        L = rng.choice([0, 1, nbp, 5 * nbp, rng.randint(1, 3 * nbp), -rng.randint(1, 50)])
        rows.append((c, s, s + L))
    qf = os.path.join(workdir, "cli_q.bed")
    write_bed(qf, rows)
    exe = os.path.join(ROOT, "bin", "igd")
    for extra in ([], ["-v", "1"], ["-v", "500"], ["-f"]):
        args = ["search", path, "-q", qf] + extra
        got = subprocess.run([exe] + args, stdout=subprocess.PIPE, check=True).stdout.decode()
        assert got == run_oracle_cli(args), extra
    for extra in ([], ["-v", "400"], ["-f"]):
        args = ["search", path, "-r", ctgs[1], "5000", "30000"] + extra
        got = subprocess.run([exe] + args, stdout=subprocess.PIPE, check=True).stdout.decode()
        assert got == run_oracle_cli(args), extra


def test_cli_f_prints_the_query_lines_of_a_batch_without_any_overlap(workdir):
    """`-f` prints "Query c, s, e: " for every query that reaches a tile of its contig, also when the whole batch
    finds nothing (src/igd_search.c:548 comes before the tile is looked at): the engine's sink is called once for such
    a batch, too."""
    rng = random.Random(7057)
    nbp = 1 << 16
    path, ctgs, span = _random_db(rng, workdir, "cli0", nbp, 1, 3, 2, 4, 1)
    orc = Oracle(path)
    try:
        free = []
        for c in range(len(ctgs)):
            for s in range(0, nbp, 97):             # tile 0 exists on every contig of the database
                one = (np.array([c], np.int32), np.array([s], np.int32), np.array([s + 1], np.int32))
                if orc.search(*one, 0)[1] == 0:
                    free.append((ctgs[c], s, s + 1))
                if len(free) >= 30:
                    break
    finally:
        orc.close()
    assert free
    qf = os.path.join(workdir, "cli0_q.bed")
    write_bed(qf, free)
    exe = os.path.join(ROOT, "bin", "igd")
    args = ["search", path, "-q", qf, "-f"]
    want = run_oracle_cli(args)
    assert "Total overlaps: 0" in want and want.count("Query ") > 0     # the case this test is about
    got = subprocess.run([exe] + args, stdout=subprocess.PIPE, check=True).stdout.decode()
    assert got == want


@pytest.mark.parametrize("case", [0, 1, 3, 5])
def test_hitmap_matches_oracle(case, workdir):
    """`-m`: dataset x dataset matrix (getMap / getMap_v) on fuzzed databases, incl. a hot tile
    with > 256 records (several LDS stages) and records spanning many tiles (the tS skip)."""
    from igd_amd import Database
    rng = random.Random(1234 + case)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[case]
    path, ctgs, span = _random_db(rng, workdir, "m%d" % case, nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc = Oracle(path)
    db = Database(path)
    try:
        for v in (0, 1, 500, 999):
            want, wtot = orc.hitmap(v)
            got, gtot = db.hitmap(v)
            assert gtot == wtot == int(want.sum()), (case, v)
            np.testing.assert_array_equal(got, want, err_msg="case %d v %d" % (case, v))
        assert (db.hitmap(0)[0] == db.hitmap(0)[0].T).all()      # the relation is symmetric
    finally:
        db.close()
        orc.close()


def test_many_datasets_use_global_counters(workdir):
    """nFiles * 8 bytes > 128 KiB: no LDS copy of hits[]; the scan kernel adds to the global
    counters directly (LDS_HITS = false), and idx no longer fits the 16-bit compact image."""
    import os
    from igd_amd import Database
    rng = random.Random(8)
    nbp = 1 << 12
    nfiles = 70000
    files = []
    for f in range(nfiles):
        s = rng.randrange(0, 40 * nbp)
        files.append([("chr1", s, s + rng.randint(1, 3 * nbp), rng.randint(0, 1000))])
    path = os.path.join(workdir, "many.igd")
    write_igd_numpy(path, files, nbp=nbp, gtype=1)
    orc = Oracle(path)
    db = Database(path)
    try:
        assert db.nfiles == nfiles
        ichr, qs, qe = _random_queries(rng, [0], nbp, 40 * nbp, 2000)
        for v in (0, 500):
            want, wtot = orc.search(ichr, qs, qe, v)
            for flags in (0, 2):
                got, gtot = db.search(ichr, qs, qe, v, flags=flags)
                assert gtot == wtot
                np.testing.assert_array_equal(got, want)
        order = np.lexsort((qs, ichr))
        got, gtot = db.search(ichr[order], qs[order], qe[order], 0)
        np.testing.assert_array_equal(got, orc.search(ichr, qs, qe, 0)[0])
    finally:
        db.close()
        orc.close()
