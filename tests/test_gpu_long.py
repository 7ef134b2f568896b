"""Queries longer than four tiles on the merge join: tiles n1 .. n1+3 by the scan kernel, the last tile by the exact
walk (WALK_LAST) and every tile between them by the coverage difference arrays (coverage_body in the batch's last
launch).  Bit-exact against the oracle (get_overlaps, igd_search.c:455-520), including what a batch leaves behind for
the next one on the same handle: the arrays of one batch are cleared by the last launch of the next."""
import os
import shutil

import numpy as np
import pytest

from helpers import Oracle, short_tmpdir

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir():
    d = short_tmpdir("igL")
    yield d
    shutil.rmtree(d, ignore_errors=True)


@pytest.fixture(scope="module")
def dbpath(workdir):
    from igd_amd import synth
    path = os.path.join(workdir, "long.igd")
    synth.make_db(path, files=40, per_file=6000, seed=21, nbp_log=12, genome=synth.SMALL)
    return path


def _mixed(n, seed, lo, hi, sorted_=True):
    from igd_amd import synth
    return synth.make_queries(n, seed=seed, genome=synth.SMALL, min_len=lo, max_len=hi, sorted_=sorted_, unknown_every=97)


@pytest.mark.parametrize("build", ["auto", "lean", "full"])
def test_long_queries_every_path(build, dbpath, monkeypatch):
    from igd_amd import Database
    if build != "auto":
        monkeypatch.setenv("IGD_HIP_RANK", "0" if build == "lean" else "1")
    db, orc = Database(dbpath), Oracle(dbpath)
    try:
        for n, lo, hi in ((500, 4096 * 3, 4096 * 9), (20000, 1, 4096 * 12), (3000, 4096 * 50, 4096 * 4000), (1, 1, 2 ** 30)):
            for sorted_ in (True, False):
                ichr, qs, qe = _mixed(n, 5 + n, lo, hi, sorted_)
                if n == 1:
                    qs[:] = 0; qe[:] = 2 ** 31 - 1; ichr[:] = 0          # one query over a whole contig
                for v in (0, 300):
                    want, wtot = orc.search(ichr, qs, qe, v)
                    for flags in ((1, 0, 2) if sorted_ else (0, 2)):
                        got, gtot = db.search(ichr, qs, qe, v, flags=flags)
                        assert gtot == wtot, (build, n, lo, sorted_, v, flags)
                        np.testing.assert_array_equal(got, want, err_msg="%s n=%d lo=%d sorted=%s v=%d flags=%d" % (build, n, lo, sorted_, v, flags))
    finally:
        db.close(); orc.close()


def test_long_then_short_then_long_on_one_handle(dbpath):
    """Batches alternate between the two sets of difference arrays; what one leaves is cleared by the next one's last
    launch -- also when that next batch has no long query, breaks its promise of order, or takes the bucket path."""
    from igd_amd import Database
    db, orc = Database(dbpath), Oracle(dbpath)
    try:
        longq = _mixed(4000, 1, 4096 * 6, 4096 * 300)
        long2 = _mixed(2500, 2, 4096 * 5, 4096 * 40)
        short = _mixed(9000, 3, 1, 3000)
        uns = _mixed(5000, 4, 4096 * 6, 4096 * 90, sorted_=False)
        plan = [(longq, 1), (long2, 1), (short, 1), (longq, 0), (uns, 1), (long2, 1), (uns, 0), (longq, 1), (short, 2), (long2, 0),
                (longq, 1), (longq, 1), (short, 0), (short, 1), (long2, 1)]
        for step, (q, flags) in enumerate(plan):
            want, wtot = orc.search(*q, 0)
            # (uns, 1): a broken promise of order -- the batch adds nothing, the blocking call repeats it in auto mode
            got, gtot = db.search(*q, 0, flags=flags)
            assert gtot == wtot, step
            np.testing.assert_array_equal(got, want, err_msg="step %d" % step)
    finally:
        db.close(); orc.close()


def test_long_queries_roadmap_scale():
    """10^5 queries of 100-200 kbp against the 1900-file database: every tile is covered some dozen times."""
    from igd_amd import Database, synth
    from test_gpu_limits import _roadmap
    path = _roadmap()
    db, orc = Database(path), Oracle(path)
    try:
        ichr, qs, qe = synth.make_queries(100000, seed=7, genome=synth.HG38, min_len=100000, max_len=200000, sorted_=True)
        want, wtot = orc.search(ichr[::20], qs[::20], qe[::20], 0)
        got, gtot = db.search(ichr[::20], qs[::20], qe[::20], 0, flags=1)
        assert gtot == wtot
        np.testing.assert_array_equal(got, want)
        # the full batch, every mode, against the oracle's committed counts of it (tools/make_bench_checksums.py)
        import json
        gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bench_checksums.json")))["workloads"]
        w = np.arange(1, db.nfiles + 1, dtype=np.uint64)
        for v in (0, 500):
            g = gold["long_sorted_q100000_v%d" % v]
            for flags in (1, 0, 2):
                h, tot = db.search(ichr, qs, qe, v, flags=flags)
                assert tot == g["total"] and int(h.sum()) == tot, (v, flags)
                assert int((h.astype(np.uint64) * w).sum() & np.uint64((1 << 63) - 1)) == g["checksum"], (v, flags)
    finally:
        db.close(); orc.close()
