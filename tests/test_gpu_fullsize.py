"""GPU, BASELINE.json's full sizes (roadmap-scale synthetic .igd: 1900 files x 26316 intervals,
53 M tile records; 10^6 queries): size-independent properties, plus the REAL reference's totals
when the prebuilt oracle/_ref/igd travelled with the repo.

  * every grouping / image choice of the engine gives the identical per-file vector
  * any permutation and any sharding of the queries gives the identical vector (hits[] is a sum)
  * sum(hits) == returned total; the -v path equals the reference at v=500
  * histogram of the enumerated (-f) records' dataset index == the counted hits vector
  * the engine's exact work statistics (algorithmic-byte model) equal the oracle's
"""
import os
import subprocess

import numpy as np
import pytest

from helpers import REF_BIN, Oracle, have_ref, parse_hits_table

pytestmark = pytest.mark.gpu
DIR = "/tmp/igdb"
FILES, PER_FILE, Q = 1900, 26316, 1000000


@pytest.fixture(scope="module")
def big():
    from igd_amd import Database, synth
    path = os.path.join(DIR, "rm%dx%d.igd" % (FILES, PER_FILE))
    if not (os.path.exists(path) and os.path.exists(path + ".done")):
        os.makedirs(DIR, exist_ok=True)
        synth.make_db(path, files=FILES, per_file=PER_FILE, seed=1000, nbp_log=14, genome=synth.HG38)
        open(path + ".done", "w").write("ok")
    db = Database(path)
    q = synth.make_queries(Q, seed=7, genome=synth.HG38, sorted_=True)
    yield db, path, q
    db.close()


def test_all_engine_paths_agree_and_match_reference(big):
    from igd_amd import synth
    db, path, (ichr, qs, qe) = big
    base, tot = db.search(ichr, qs, qe)
    assert base.sum() == tot and tot > 3e7
    for flags in (2, 4, 6):                       # bucket, exact arrays, both
        h, t = db.search(ichr, qs, qe, flags=flags)
        assert t == tot
        np.testing.assert_array_equal(h, base)
    perm = np.random.default_rng(1).permutation(Q)
    h, t = db.search(ichr[perm], qs[perm], qe[perm])
    np.testing.assert_array_equal(h, base)
    acc = np.zeros_like(base)
    for k in range(8):                            # 8 contiguous shards, as 8 GPUs would take them
        lo, hi = k * Q // 8, (k + 1) * Q // 8
        acc += db.search(ichr[lo:hi], qs[lo:hi], qe[lo:hi])[0]
    np.testing.assert_array_equal(acc, base)
    hv, tv = db.search(ichr, qs, qe, v=500)
    assert hv.sum() == tv and 0.3 * tot < tv < 0.7 * tot
    np.testing.assert_array_equal(db.search(ichr, qs, qe, v=500, flags=6)[0], hv)
    if have_ref():
        bed = os.path.join(DIR, "t_q.bed")
        synth.write_bed(bed, synth.HG38, ichr, qs, qe)
        for extra, want in (([], base), (["-v", "500"], hv)):
            out = subprocess.run([REF_BIN, "search", path, "-q", bed] + extra, stdout=subprocess.PIPE, check=True).stdout.decode()
            rh, rt = parse_hits_table(out, db.nfiles)
            np.testing.assert_array_equal(rh, want)
            assert rt == want.sum()


def test_enumeration_histogram_equals_counts(big):
    db, path, (ichr, qs, qe) = big
    n = 200000
    sl = slice(300000, 300000 + n)
    hits, tot = db.search(ichr[sl], qs[sl], qe[sl])
    qoff, rec = db.enumerate(ichr[sl], qs[sl], qe[sl])
    assert qoff[-1] == tot == len(rec)
    np.testing.assert_array_equal(np.bincount(rec[:, 1], minlength=db.nfiles), hits)
    assert (np.diff(qoff) >= 0).all()
    np.testing.assert_array_equal(rec[:, 0], np.repeat(np.arange(n), np.diff(qoff)))
    # every emitted record really overlaps its query (half-open on both sides)
    q = rec[:, 0]
    assert (rec[:, 2] < qe[sl][q]).all() and (rec[:, 3] > qs[sl][q]).all()
    # spot-check the order against the oracle on the first 2000 queries
    o = Oracle(path)
    wq, wr = o.enumerate(ichr[sl][:2000], qs[sl][:2000], qe[sl][:2000])
    np.testing.assert_array_equal(qoff[:2001], wq)
    np.testing.assert_array_equal(rec[: wq[-1], 1:], wr)
    o.close()


def test_config5_enumeration_of_all_queries_streamed_equals_counts_and_whole_array(big):
    """BASELINE config 5 at its stated size: -f on ALL 10^6 queries.  The streamed chunks (igd_hip_enumerate_stream)
    cover every query once, in order; their histogram over dataset index is the counted hits[] vector; and
    the concatenation equals what the one-array API (igd_hip_enumerate) returns."""
    db, path, (ichr, qs, qe) = big
    hits, tot = db.search(ichr, qs, qe)
    hist = np.zeros(db.nfiles, np.int64)
    seen = {"q": 0, "n": 0, "chunks": 0, "crc": 0}

    def on_chunk(q0, q1, qoff, rec):
        assert q0 == seen["q"] and q1 > q0 and len(rec) == qoff[q1] - qoff[q0] and qoff[q0] == seen["n"]
        hist[:] += np.bincount(rec[:, 1], minlength=db.nfiles)
        np.testing.assert_array_equal(rec[:, 0], np.repeat(np.arange(q0, q1), np.diff(qoff[q0:q1 + 1])))
        assert (rec[:, 2] < qe[rec[:, 0]]).all() and (rec[:, 3] > qs[rec[:, 0]]).all()
        seen["crc"] = (seen["crc"] * 31 + int(rec.astype(np.int64).sum())) & 0xFFFFFFFFFFFF
        seen["q"], seen["n"], seen["chunks"] = q1, seen["n"] + len(rec), seen["chunks"] + 1

    qoff, total = db.enumerate_stream(ichr, qs, qe, on_chunk)
    assert seen["q"] == Q and seen["n"] == total == tot == qoff[-1] and seen["chunks"] >= 5
    np.testing.assert_array_equal(hist, hits)
    qoff2, rec = db.enumerate(ichr, qs, qe)
    np.testing.assert_array_equal(qoff2, qoff)
    crc, qa, nch = 0, 0, 0
    while qa < Q:                                   # same chunking rule: longest query range within 2 Mi overlaps
        qb = max(qa + 1, int(np.searchsorted(qoff, qoff[qa] + (2 << 20), side="right")) - 1)
        crc = (crc * 31 + int(rec[qoff[qa]:qoff[qb]].astype(np.int64).sum())) & 0xFFFFFFFFFFFF
        qa, nch = qb, nch + 1
    assert nch == seen["chunks"] and crc == seen["crc"]


def test_config5_packed_stream_expands_to_the_whole_array(big):
    """Config 5 through the 8-byte stream (igd_hip_enumerate_stream8: what the command line tool now moves over PCIe): 1900 files
    -> 11 bits of idx + 21 bits of length; every chunk expanded equals the same range of the one-array API's 16-byte records."""
    db, path, (ichr, qs, qe) = big
    qoff2, whole = db.enumerate(ichr, qs, qe)
    bits = db.hit8_idx_bits()
    assert bits == 11
    seen = {"q": 0, "n": 0, "chunks": 0}

    def on_chunk(q0, q1, qoff, rec, b):
        assert b == bits and q0 == seen["q"] and q1 > q0 and len(rec) == qoff[q1] - qoff[q0] and qoff[q0] == seen["n"]
        st, en, ix = db.expand_hit8(rec, b)
        w = whole[qoff[q0]:qoff[q1]]
        assert np.array_equal(ix, w[:, 1]) and np.array_equal(st, w[:, 2]) and np.array_equal(en, w[:, 3])
        seen["q"], seen["n"], seen["chunks"] = q1, seen["n"] + len(rec), seen["chunks"] + 1

    qoff, total = db.enumerate_stream8(ichr, qs, qe, on_chunk)
    np.testing.assert_array_equal(qoff, qoff2)
    assert seen["q"] == Q and seen["n"] == total == qoff[-1] and seen["chunks"] >= 3


def test_config5_cli_f_on_all_queries_is_byte_identical_to_the_reference(big):
    """`bin/igd search db -q 10^6.bed -f` (1.2 GB of text, streamed chunk by chunk) against the real reference
    binary's stdout: same bytes (md5 + length), when oracle/_ref/igd travelled; otherwise against the oracle CLI."""
    import hashlib
    from helpers import ORACLE_BIN
    from igd_amd import synth
    db, path, (ichr, qs, qe) = big
    bed = os.path.join(DIR, "t_q.bed")
    if not os.path.exists(bed):
        synth.write_bed(bed, synth.HG38, ichr, qs, qe)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def digest(exe):
        p = subprocess.Popen([exe, "search", path, "-q", bed, "-f"], stdout=subprocess.PIPE)
        h, n = hashlib.md5(), 0
        while True:
            b = p.stdout.read(1 << 24)
            if not b:
                break
            h.update(b)
            n += len(b)
        assert p.wait() == 0
        return h.hexdigest(), n
    mine = digest(os.path.join(root, "bin", "igd"))
    theirs = digest(REF_BIN if have_ref() else ORACLE_BIN)
    assert mine == theirs and mine[1] > 1e9


def test_work_statistics_equal_oracle(big):
    import torch
    db, path, (ichr, qs, qe) = big
    n = 20000
    idx = np.arange(0, Q, Q // n)[:n]
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(np.ascontiguousarray(x[idx])).to(dev) for x in (ichr, qs, qe)]
    o = Oracle(path)
    for v in (0, 500):
        st = db.batch_stats(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, v=v)
        o.search(ichr[idx], qs[idx], qe[idx], v)
        w = o.stats()
        assert (st["pairs"], st["S"], st["B"], st["H"]) == (w["pairs"], w["S"], w["B"], w["H"]), (v, st, w)
    o.close()


def test_hitmap_full_scale_equals_reference_cli(big, tmp_path):
    """`-m` on the roadmap-scale database (1.7e9 overlapping pairs): the matrix file and the stdout
    of bin/igd are byte-identical to the reference CLI's (~15 s of CPU), and the matrix is symmetric
    with a diagonal >= the number of records per file."""
    db, path, _ = big
    m, tot = db.hitmap(0)
    assert tot == int(m.sum(dtype=np.uint64)) and (m == m.T).all()
    assert (np.diag(m) >= 26316).all()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mine = os.path.join(DIR, "hm_gpu.txt")
    p = subprocess.run([os.path.join(root, "bin", "igd"), "search", path, "-m", "-o", mine], stdout=subprocess.PIPE, check=True)
    got = np.loadtxt(mine, dtype=np.uint32, skiprows=1)
    np.testing.assert_array_equal(got, m)
    if have_ref():
        theirs = os.path.join(DIR, "hm_ref.txt")
        r = subprocess.run([REF_BIN, "search", path, "-m", "-o", theirs], stdout=subprocess.PIPE, check=True)
        assert r.stdout == p.stdout
        assert open(theirs, "rb").read() == open(mine, "rb").read()
