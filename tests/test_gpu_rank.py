"""The merge join's RANK method (igd_scan_sorted: tiles with >= 32 first-tile queries are counted by two
bisections per record/query instead of pairwise compares) against the oracle: dense position-sorted batches on
small databases with every awkward query kind -- inverted (qe < qs), zero-length, negative starts, unknown contigs,
starts beyond the contig's last tile, queries over many tiles -- with and without the value filter, on multi-chunk
tiles and on gType 0; and a batch ordered by tile but NOT by start inside a tile, which must fall back to pairwise
compares (CTL_NOTSTART) and still be exact."""
import random
import shutil

import numpy as np
import pytest

from helpers import Oracle, short_tmpdir
from test_gpu_parity import CASES, _random_db, _random_queries

pytestmark = pytest.mark.gpu
FLAG_SORTED, FLAG_BUCKET, FLAG_EXACT = 1, 2, 4


@pytest.fixture(scope="module")
def workdir():
    d = short_tmpdir("igr")
    yield d
    shutil.rmtree(d, ignore_errors=True)


def _dense_queries(rng, nctg, nbp, span, n):
    ichr, qs, qe = _random_queries(rng, list(range(nctg)), nbp, span, n)
    neg = np.array([rng.random() < 0.01 for _ in range(n)])
    qs = np.where(neg, -np.array([rng.randrange(1, nbp) for _ in range(n)], np.int32), qs).astype(np.int32)
    lens = np.array([rng.choice([0, 1, 7, nbp // 50 + 1, nbp // 4, nbp, 2 * nbp + 5, 9 * nbp + 3, -rng.randint(1, 300)])
                     for _ in range(n)], np.int32)
    qe = (qs + lens).astype(np.int32)
    order = np.lexsort((qs, ichr))
    return ichr[order], qs[order], qe[order]


@pytest.mark.parametrize("build", ["auto", "lean", "full"])
@pytest.mark.parametrize("case", [1, 2, 3, 5, 6])
def test_dense_sorted_batches_take_the_rank_path_and_match_the_oracle(case, build, workdir, monkeypatch):
    """build: which igd_scan_sorted the engine launches -- auto (by queries per tile: the full build here), lean
    (pairwise only; tiles with > 512 first-tile queries go to the skew valve, heavy_sorted_body) or full (rank method from 32 on)."""
    from igd_amd import Database
    if build != "auto":
        monkeypatch.setenv("IGD_HIP_RANK", "0" if build == "lean" else "1")
    rng = random.Random(9000 + case)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[case]
    path, ctgs, span = _random_db(rng, workdir, "rk%d" % case, nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc, db = Oracle(path), Database(path)
    try:
        n = 70000                                 # (>= 65536: k_query_bounds with four queries per thread)
        ichr, qs, qe = _dense_queries(rng, nctg, nbp, span, n)
        ntiles = sum(db.ntile)
        assert n / ntiles > 64                    # far beyond IGD_DENSE_MIN per tile
        for v in (0, 300):
            want, wtot = orc.search(ichr, qs, qe, v)
            for flags in (0, FLAG_SORTED, FLAG_BUCKET, FLAG_EXACT):
                got, gtot = db.search(ichr, qs, qe, v, flags=flags)
                assert gtot == wtot, (case, v, flags)
                np.testing.assert_array_equal(got, want, err_msg="case %d v %d flags %d" % (case, v, flags))
        # rule FLAT without a value filter (every tile visited, also behind an empty first tile)
        want = orc.search_rule(ichr, qs, qe, 1) if hasattr(orc, "search_rule") else None
        got_flat = db.search(ichr, qs, qe, rule=1)[0]
        np.testing.assert_array_equal(got_flat, db.search(ichr, qs, qe, rule=1, flags=FLAG_BUCKET)[0])
        if want is not None:
            np.testing.assert_array_equal(got_flat, want[0])
    finally:
        db.close(); orc.close()


def test_tile_ordered_but_start_unordered_batch_falls_back_to_pairwise(workdir):
    """Keys (first tiles) non-decreasing, starts shuffled inside every tile: still a legal merge-join batch; the
    rank method's bisection of q_qs[] would be wrong on it, so the device must notice (CTL_NOTSTART) and compare
    pairwise -- counts equal the oracle's."""
    from igd_amd import Database
    rng = random.Random(31337)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[1]
    path, ctgs, span = _random_db(rng, workdir, "ns", nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc, db = Oracle(path), Database(path)
    try:
        n = 30000
        ichr, qs, qe = _dense_queries(rng, nctg, nbp, span, n)
        ok = (ichr >= 0) & (ichr < nctg) & (qs >= 0)
        ichr, qs, qe = ichr[ok], qs[ok], qe[ok]
        tile = qs // nbp
        perm = np.random.default_rng(5).permutation(len(qs))
        order = perm[np.lexsort((tile[perm], ichr[perm]))]          # by (contig, tile), random inside a tile
        a, b, c = ichr[order], qs[order], qe[order]
        assert (np.diff(b)[np.diff(a * 10**6 + b // nbp) == 0] < 0).any()
        for v in (0, 300):
            want, wtot = orc.search(a, b, c, v)
            got, gtot = db.search(a, b, c, v, flags=FLAG_SORTED)      # the promise (tile order) holds
            assert gtot == wtot
            np.testing.assert_array_equal(got, want)
    finally:
        db.close(); orc.close()


@pytest.mark.parametrize("build", ["lean", "full"])
def test_every_query_reaching_into_later_tiles(build, workdir, monkeypatch):
    """All queries one to three tiles long: every one of them leaves a later-tile word, so k_query_bounds' later
    blocks are full (1024 entries, 16 groups, no terminating zero) and a unit finds its candidates deep inside a
    block -- the group bisection and the walk back over block boundaries of for_later_groups -- on aligned
    (1024-query blocks) and unaligned (256-query blocks) device arrays."""
    import torch
    from igd_amd import Database
    monkeypatch.setenv("IGD_HIP_RANK", "0" if build == "lean" else "1")
    rng = random.Random(4242)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[1]
    path, ctgs, span = _random_db(rng, workdir, "lt", nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc, db = Oracle(path), Database(path)
    try:
        for n in (70000, 3000):                  # dense (rank method in the full build; >= 65536: four queries per thread) and sparse
            ichr, qs, qe = _random_queries(rng, list(range(nctg)), nbp, span, n)
            lens = np.array([rng.choice([nbp, nbp + 17, 2 * nbp + 5, 3 * nbp - 1]) for _ in range(n)], np.int32)
            qe = (qs + lens).astype(np.int32)
            order = np.lexsort((qs, ichr))
            ichr, qs, qe = ichr[order], qs[order], qe[order]
            want, wtot = orc.search(ichr, qs, qe, 0)
            got, gtot = db.search(ichr, qs, qe, 0, flags=FLAG_SORTED)
            assert gtot == wtot
            np.testing.assert_array_equal(got, want)
            # the same batch from device arrays that start 4 bytes off a 16-byte boundary (k_query_bounds<1>)
            dev = torch.device("cuda:0")
            buf = [torch.empty(n + 1, dtype=torch.int32, device=dev) for _ in range(3)]
            for b, a in zip(buf, (ichr, qs, qe)):
                b[1:].copy_(torch.from_numpy(a))
            d_hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
            d_tot = torch.zeros(1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            db.search_dev(buf[0][1:].data_ptr(), buf[1][1:].data_ptr(), buf[2][1:].data_ptr(), n, d_hits.data_ptr(),
                          d_tot.data_ptr(), flags=FLAG_SORTED)
            db.sync()
            np.testing.assert_array_equal(d_hits.cpu().numpy(), want)
            assert int(d_tot.item()) == wtot
    finally:
        db.close(); orc.close()
