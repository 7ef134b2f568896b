"""GPU stress shapes (SURVEY 8d "stress"): a clustered database (half of the intervals around 2000
hot spots -> tiles with thousands of records, i.e. many chunks per tile), a sparse one (most tiles
empty -> the NEST/FLAT difference at scale), wide tiles (nbp 2^19: no compact image), and a large
batch (1.25e7 queries: BASELINE config 4's per-GPU share).  All bit-exact against the oracle on a
sample, and self-consistent (modes agree) on the whole batch."""
import os
import shutil

import numpy as np
import pytest

from helpers import Oracle, short_tmpdir

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir():
    d = short_tmpdir("igs")
    yield d
    shutil.rmtree(d, ignore_errors=True)


def _check(db, orc, ichr, qs, qe, sample=20000, vs=(0, 500)):
    n = len(qs)
    idx = np.sort(np.random.default_rng(3).choice(n, size=min(sample, n), replace=False))
    for v in vs:
        want, wtot = orc.search(ichr[idx], qs[idx], qe[idx], v)
        for flags in (0, 2, 4):
            got, gtot = db.search(ichr[idx], qs[idx], qe[idx], v, flags=flags)
            assert gtot == wtot, (v, flags)
            np.testing.assert_array_equal(got, want, err_msg="v=%d flags=%d" % (v, flags))
    full, tot = db.search(ichr, qs, qe)
    assert full.sum() == tot
    np.testing.assert_array_equal(db.search(ichr, qs, qe, flags=2)[0], full)
    return full


def test_clustered_database_hot_tiles(workdir):
    from igd_amd import Database, synth
    path = os.path.join(workdir, "cl.igd")
    synth.make_db(path, files=300, per_file=40000, seed=77, genome=synth.HG38, clustered=True)
    db, orc = Database(path), Oracle(path)
    try:
        cnt = [orc.lib.orc_ncnt(orc.h, 0, j) for j in range(orc.lib.orc_ntile(orc.h, 0))]
        assert max(cnt) > 1000          # really has multi-chunk tiles (a unit holds 320 records)
        ichr, qs, qe = synth.make_queries(300000, seed=9, genome=synth.HG38)
        _check(db, orc, ichr, qs, qe)
        # queries concentrated on the hottest tile (many queries x many chunks)
        hot = int(np.argmax(cnt))
        rng = np.random.default_rng(1)
        hs = (hot * 16384 + rng.integers(-20000, 36000, 50000)).astype(np.int32)
        hs = np.maximum(hs, 0)
        he = hs + rng.integers(1, 40000, 50000).astype(np.int32)
        hc = np.zeros(50000, np.int32)
        order = np.argsort(hs, kind="stable")
        _check(db, orc, hc[order], hs[order], he[order], sample=5000)
        # enumeration on a hot tile keeps the reference order across chunks
        wq, wr = orc.enumerate(hc[:300], hs[:300], he[:300])
        gq, gr = db.enumerate(hc[:300], hs[:300], he[:300])
        np.testing.assert_array_equal(gq, wq)
        np.testing.assert_array_equal(gr[:, 1:], wr)
    finally:
        db.close(); orc.close()


def test_sparse_database_quirk_at_scale(workdir):
    from igd_amd import Database, synth
    path = os.path.join(workdir, "sp.igd")
    synth.make_db(path, files=40, per_file=300, seed=5, genome=synth.HG38)      # 12k intervals over 3 Gbp
    db, orc = Database(path), Oracle(path)
    try:
        ichr, qs, qe = synth.make_queries(400000, seed=11, genome=synth.HG38, min_len=1000, max_len=120000)
        _check(db, orc, ichr, qs, qe, vs=(0, 1, 500))
        nest = db.search(ichr, qs, qe, 0)[1]
        flat = db.search(ichr, qs, qe, rule=1)[1]          # FLAT without value filter
        assert nest < flat                                  # the empty-first-tile rule drops hits
    finally:
        db.close(); orc.close()


def test_wide_tiles_use_exact_arrays(workdir):
    from igd_amd import Database, synth
    path = os.path.join(workdir, "wide.igd")
    synth.make_db(path, files=20, per_file=4000, seed=6, nbp_log=19, genome=synth.HG38)
    db, orc = Database(path), Oracle(path)
    try:
        assert db.nbp == 1 << 19
        ichr, qs, qe = synth.make_queries(100000, seed=12, genome=synth.HG38, min_len=1, max_len=2000000)
        _check(db, orc, ichr, qs, qe)
    finally:
        db.close(); orc.close()


def _roadmap():
    from igd_amd import synth
    path = "/tmp/igdb/rm1900x26316.igd"
    if not os.path.exists(path + ".done"):
        os.makedirs("/tmp/igdb", exist_ok=True)
        synth.make_db(path)
        open(path + ".done", "w").write("ok")
    return path


def test_config4_share_of_queries_one_gpu():
    """1.25e7 queries in ONE device batch (BASELINE config 4: 1e8 over 8 GPUs)."""
    from igd_amd import Database, synth
    path = _roadmap()
    db, orc = Database(path), Oracle(path)
    try:
        Q = 12500000
        ichr, qs, qe = synth.make_queries(Q, seed=21, genome=synth.HG38)
        full = _check(db, orc, ichr, qs, qe, sample=20000, vs=(0,))
        acc = np.zeros_like(full)
        for k in range(8):
            lo, hi = k * Q // 8, (k + 1) * Q // 8
            acc += db.search(ichr[lo:hi], qs[lo:hi], qe[lo:hi])[0]
        np.testing.assert_array_equal(acc, full)
    finally:
        db.close(); orc.close()


def _golden(key):
    import json
    from helpers import GOLDEN
    w = json.load(open(os.path.join(GOLDEN, "bench_checksums.json")))["workloads"][key]
    return w["total"], w["checksum"]


def _checksum(h):
    return int((h.astype(np.uint64) * (np.arange(len(h), dtype=np.uint64) + np.uint64(1))).sum() & np.uint64((1 << 63) - 1))


@pytest.mark.parametrize("workload", ["config4_share_q12500000", "config4_slab0_of_8"])
def test_dense_batches_whole_against_the_oracle(workload):
    """The rank method at the scale bench.py runs it: config 4's per-GPU share (1.25e7 position-sorted queries, ~66 per
    tile) and one GPU's slab of the 8-GPU job (~530 per tile on an eighth of the tiles) -- EVERY query of the batch
    against the oracle (not a sample: a sample is sparse and takes the pairwise path), with and without the value filter,
    under the order promise and with the device deciding; and against the committed oracle checksums bench.py uses."""
    from igd_amd import Database, synth
    path = _roadmap()
    n = 12500000
    if workload == "config4_share_q12500000":
        ichr, qs, qe = synth.make_queries_slab(n, 0, n, seed=7, genome=synth.HG38)
        a = synth.make_queries(n, seed=7, genome=synth.HG38, sorted_=True)       # what `bench.py --queries 12500000` runs
        assert all(np.array_equal(x, y) for x, y in zip(a, (ichr, qs, qe)))
        del a
    else:
        ichr, qs, qe = synth.make_queries_slab(8 * n, 0, n, seed=7, genome=synth.HG38)
    db, orc = Database(path), Oracle(path)
    try:
        for v in (0, 500):
            want, wtot = orc.search(ichr, qs, qe, v)
            assert (int(wtot), _checksum(want)) == _golden("%s_v%d" % (workload, v))
            for flags in (1, 0):
                got, gtot = db.search(ichr, qs, qe, v, flags=flags)
                assert gtot == wtot, (workload, v, flags)
                np.testing.assert_array_equal(got, want, err_msg="%s v=%d flags=%d" % (workload, v, flags))
    finally:
        db.close(); orc.close()


def test_mid_size_batch_lean_build_long_queries(workdir):
    """Between one and eight queries per tile on a database of tens of thousands of tiles: the lean build of
    igd_scan_sorted fed by k_query_bounds with four queries per thread (>= 65536 queries), with queries up to six tiles
    long -- later-tile words for one, two and three tiles, full later blocks, queries left to the exact walk -- and
    unknown contigs; every count against the oracle, in the promised-order, device-decides and bucket modes."""
    from igd_amd import Database, synth
    path = os.path.join(workdir, "mid.igd")
    nbp_log = 9
    synth.make_db(path, files=24, per_file=8000, seed=5, nbp_log=nbp_log, genome=synth.SMALL)
    db, orc = Database(path), Oracle(path)
    try:
        nT = int(sum(db.ntile))
        n = max(70000, 3 * nT)
        if not (nT <= n < 8 * nT):
            pytest.skip("synthetic genome gives %d tiles: not the regime this test is about" % nT)
        ichr, qs, qe = synth.make_queries(n, seed=11, genome=synth.SMALL, min_len=1, max_len=6 << nbp_log, sorted_=True,
                                          unknown_every=97, extra_span=3 << nbp_log)
        for v in (0, 400):
            want, wtot = orc.search(ichr, qs, qe, v)
            for flags in (1, 0, 2):
                got, gtot = db.search(ichr, qs, qe, v, flags=flags)
                assert gtot == wtot, (v, flags)
                np.testing.assert_array_equal(got, want, err_msg="v=%d flags=%d" % (v, flags))
    finally:
        db.close(); orc.close()
