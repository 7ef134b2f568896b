"""The engine groups queries by tile in one of two ways (include/igd_hip.h, "flags"): a merge
join when the batch is ordered by (contig, start), a counting sort otherwise.  Both must give
the oracle's counts on the SAME batch; the device-side order check must pick correctly; and a
broken IGD_HIP_FLAG_SORTED promise must be reported, adding nothing."""
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
    d = short_tmpdir("igg")
    yield d
    shutil.rmtree(d, ignore_errors=True)


def _sorted(ichr, qs, qe):
    order = np.lexsort((qs, ichr))            # by contig index, then start (stable)
    return ichr[order], qs[order], qe[order]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_sorted_batches_all_modes(case, workdir):
    from igd_amd import Database
    rng = random.Random(31337 + case)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[case]
    path, ctgs, span = _random_db(rng, workdir, "g%d" % case, nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc = Oracle(path)
    db = Database(path)
    try:
        ichr, qs, qe = _random_queries(rng, list(range(nctg)), nbp, span, 4000)
        # some very long queries (dozens of tiles) and negative starts inside tile 0
        ichr[:40] = 0
        qs[:40] = np.array([rng.randrange(0, span) for _ in range(40)], np.int32)
        qe[:40] = qs[:40] + np.array([rng.randrange(5 * nbp, 60 * nbp) for _ in range(40)], np.int32)
        qs[40:50] = -np.array([rng.randrange(1, nbp) for _ in range(10)], np.int32)
        qe[40:50] = np.array([rng.randrange(1, 3 * nbp) for _ in range(10)], np.int32)
        ichr, qs, qe = _sorted(ichr, qs, qe)
        for v in (0, 1, 500):
            want, wtot = orc.search(ichr, qs, qe, v)
            for flags in (0, FLAG_BUCKET, FLAG_EXACT, FLAG_BUCKET | FLAG_EXACT):
                got, gtot = db.search(ichr, qs, qe, v, flags=flags)
                assert gtot == wtot, (case, v, flags)
                np.testing.assert_array_equal(got, want, err_msg="case %d v %d flags %d" % (case, v, flags))
    finally:
        db.close()
        orc.close()


def test_device_api_sorted_promise(workdir):
    """igd_hip_search_dev with IGD_HIP_FLAG_SORTED: same counts; a broken promise is reported by
    igd_hip_sync and adds nothing."""
    import torch
    from igd_amd import Database
    from igd_amd.database import IgdError
    rng = random.Random(2024)
    nbp = 1 << 12
    path, ctgs, span = _random_db(rng, workdir, "prom", nbp, 1, 9, 3, 30, 80, 20)
    orc = Oracle(path)
    db = Database(path)
    try:
        ichr, qs, qe = _random_queries(rng, [0, 1, 2], nbp, span, 5000)
        si, ss, se = _sorted(ichr, qs, qe)
        dev = torch.device("cuda", 0)
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            for (a, b, c, promise_ok) in ((si, ss, se, True), (ichr, qs, qe, False)):
                t = [torch.from_numpy(x).to(dev) for x in (a, b, c)]
                hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
                tot = torch.zeros(1, dtype=torch.int64, device=dev)
                for rule_v in (0, 300):
                    hits.zero_(); tot.zero_()
                    stream.synchronize()
                    db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(a), hits.data_ptr(),
                                  tot.data_ptr(), v=rule_v, stream=stream.cuda_stream, flags=FLAG_SORTED)
                    if promise_ok:
                        db.sync(stream.cuda_stream)
                        want, wtot = orc.search(a, b, c, rule_v)
                        np.testing.assert_array_equal(hits.cpu().numpy(), want)
                        assert int(tot.item()) == wtot
                    else:
                        with pytest.raises(IgdError):
                            db.sync(stream.cuda_stream)
                        assert int(hits.sum().item()) == 0 and int(tot.item()) == 0
                # auto mode on the same arrays is always right
                hits.zero_(); tot.zero_()
                stream.synchronize()
                db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(a), hits.data_ptr(),
                              tot.data_ptr(), v=0, stream=stream.cuda_stream, flags=0)
                db.sync(stream.cuda_stream)
                np.testing.assert_array_equal(hits.cpu().numpy(), orc.search(a, b, c, 0)[0])
                # the API adds ... unless IGD_HIP_FLAG_ZERO_FIRST (8) asks the batch to clear first
                for fl in (0, 2):
                    db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(a), hits.data_ptr(),
                                  tot.data_ptr(), v=0, stream=stream.cuda_stream, flags=fl)
                db.sync(stream.cuda_stream)
                np.testing.assert_array_equal(hits.cpu().numpy(), 3 * orc.search(a, b, c, 0)[0])
                for fl in (8, 8 | 2):
                    db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(a), hits.data_ptr(),
                                  tot.data_ptr(), v=0, stream=stream.cuda_stream, flags=fl)
                    db.sync(stream.cuda_stream)
                    np.testing.assert_array_equal(hits.cpu().numpy(), orc.search(a, b, c, 0)[0])
                    assert int(tot.item()) == orc.search(a, b, c, 0)[1]
    finally:
        db.close()
        orc.close()


def test_alternating_batches_keep_workspace_clean(workdir):
    """sorted -> unsorted -> enumerate -> sorted ... on one handle: the per-batch device state
    (pair counters, long-query counters, order flags) never leaks into the next batch."""
    from igd_amd import Database
    rng = random.Random(77)
    nbp = 1 << 11
    path, ctgs, span = _random_db(rng, workdir, "alt", nbp, 1, 8, 2, 30, 60, 0)
    orc = Oracle(path)
    db = Database(path)
    try:
        for rnd in range(6):
            ichr, qs, qe = _random_queries(rng, [0, 1], nbp, span, 1500)
            qe[:30] = qs[:30] + 20 * nbp                 # long queries every time
            if rnd % 2 == 0:
                ichr, qs, qe = _sorted(ichr, qs, qe)
            v = (0, 500)[rnd % 2]
            want, wtot = orc.search(ichr, qs, qe, v)
            got, gtot = db.search(ichr, qs, qe, v)
            assert gtot == wtot
            np.testing.assert_array_equal(got, want)
            if rnd % 3 == 0:
                wq, wr = orc.enumerate(ichr[:300], qs[:300], qe[:300])
                gq, gr = db.enumerate(ichr[:300], qs[:300], qe[:300])
                np.testing.assert_array_equal(gq, wq)
                np.testing.assert_array_equal(gr[:, 1:], wr)
    finally:
        db.close()
        orc.close()


@pytest.mark.parametrize("nbp_log,wide_values", [(11, False), (12, True), (15, False), (16, False)])
def test_compact_image_edges(nbp_log, wide_values, workdir):
    """The 6-byte tile-relative image must agree with the exact arrays everywhere, including
    the one case it cannot express (first-tile query with qe <= tile start: inverted queries that
    reach back over a tile boundary), values outside int16 (falls back to exact arrays for -v)
    and tile widths at / beyond its 16-bit limit (2^15 packs, 2^16 does not)."""
    import os
    from helpers import write_igd_numpy
    from igd_amd import Database
    rng = random.Random(900 + nbp_log)
    nbp = 1 << nbp_log
    span = 12 * nbp
    files = []
    for f in range(7):
        rows = []
        for _ in range(150):
            s = rng.randrange(0, span)
            L = rng.choice([1, 2, nbp - 1, nbp, nbp + 1, 3 * nbp + 5, rng.randint(1, 2 * nbp)])
            if rng.random() < 0.3:
                s = (s // nbp) * nbp - rng.choice([0, 1])          # starts at / just before a tile start
                s = max(s, 0)
            val = rng.choice([0, 1, 500, 32767, 40000, -5, -40000]) if wide_values else rng.randint(0, 1000)
            rows.append(("chr1", s, s + L, val))
        files.append(rows)
    path = os.path.join(workdir, "cmp%d%d.igd" % (nbp_log, wide_values))
    write_igd_numpy(path, files, nbp=nbp, gtype=1)
    orc = Oracle(path)
    db = Database(path)
    try:
        qs, qe = [], []
        for t in range(0, 13):
            T = t * nbp
            for dq in (-3, -1, 0, 1, 5, nbp - 1):
                for de in (-nbp - 2, -7, -1, 0, 1, 2, nbp, 2 * nbp + 1):
                    qs.append(T + dq)
                    qe.append(T + dq + de)
        for _ in range(1500):
            a = rng.randrange(-nbp + 1, span + nbp)
            qs.append(a)
            qe.append(a + rng.choice([-rng.randint(1, 3 * nbp), 0, 1, rng.randint(1, 2 * nbp)]))
        qs = np.array(qs, np.int32); qe = np.array(qe, np.int32)
        keep = qs > -nbp                                           # qs <= -nbp is reference UB
        qs, qe = qs[keep], qe[keep]
        ichr = np.zeros(len(qs), np.int32)
        for order in ("as-is", "sorted"):
            if order == "sorted":
                ichr, qs, qe = _sorted(ichr, qs, qe)
            for v in (0, 1, 501, 33000):
                want, wtot = orc.search(ichr, qs, qe, v)
                for flags in (0, FLAG_BUCKET, FLAG_EXACT):
                    got, gtot = db.search(ichr, qs, qe, v, flags=flags)
                    assert gtot == wtot, (order, v, flags)
                    np.testing.assert_array_equal(got, want, err_msg="%s v=%d flags=%d" % (order, v, flags))
    finally:
        db.close()
        orc.close()


def test_a_broken_promise_among_several_async_batches_is_not_lost(workdir):
    """Several IGD_HIP_FLAG_SORTED batches enqueued before ONE igd_hip_sync (what bench.py does): if ANY of
    them was unordered the sync reports it -- also when a later batch was ordered again -- and that batch
    added nothing; after the report the slate is clean."""
    import torch
    from igd_amd import Database
    from igd_amd.database import IgdError
    rng = random.Random(77)
    nbp = 1 << 12
    path, ctgs, span = _random_db(rng, workdir, "stick", nbp, 1, 9, 3, 30, 80, 20)
    orc = Oracle(path)
    db = Database(path)
    try:
        ichr, qs, qe = _random_queries(rng, [0, 1, 2], nbp, span, 4000)
        si, ss, se = _sorted(ichr, qs, qe)
        want = orc.search(si, ss, se, 0)[0]
        dev = torch.device("cuda", 0)
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            good = [torch.from_numpy(x).to(dev) for x in (si, ss, se)]
            bad = [torch.from_numpy(x).to(dev) for x in (ichr, qs, qe)]
            hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
            for order in (("good", "bad", "good"), ("bad", "good", "good"), ("good", "good", "bad")):
                hits.zero_()
                stream.synchronize()
                for which in order:
                    t = good if which == "good" else bad
                    db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(si), hits.data_ptr(), None,
                                  v=0, stream=stream.cuda_stream, flags=FLAG_SORTED)
                with pytest.raises(IgdError):
                    db.sync(stream.cuda_stream)
                np.testing.assert_array_equal(hits.cpu().numpy(), 2 * want)     # the two ordered batches only
                # reported once: the next sync, and a following all-ordered job, are clean
                db.sync(stream.cuda_stream)
                hits.zero_()
                for _ in range(2):
                    db.search_dev(good[0].data_ptr(), good[1].data_ptr(), good[2].data_ptr(), len(si), hits.data_ptr(), None,
                                  v=0, stream=stream.cuda_stream, flags=FLAG_SORTED)
                db.sync(stream.cuda_stream)
                np.testing.assert_array_equal(hits.cpu().numpy(), 2 * want)
    finally:
        db.close()
        orc.close()


def test_unaligned_device_arrays_take_the_scalar_reader(workdir):
    """igd_hip_search_dev reads the query arrays as dwordx4 when they are 16-byte aligned and one query per thread
    otherwise (views into larger tensors): same counts."""
    import torch
    from igd_amd import Database
    rng = random.Random(99)
    nbp = 1 << 12
    path, ctgs, span = _random_db(rng, workdir, "unal", nbp, 1, 9, 3, 30, 80, 20)
    orc = Oracle(path)
    db = Database(path)
    try:
        ichr, qs, qe = _random_queries(rng, [0, 1, 2], nbp, span, 5003)
        si, ss, se = _sorted(ichr, qs, qe)
        want = orc.search(si, ss, se, 0)[0]
        dev = torch.device("cuda", 0)
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            for shift in (0, 1, 2, 3):
                t = [torch.cat([torch.zeros(shift, dtype=torch.int32), torch.from_numpy(x)]).to(dev)[shift:] for x in (si, ss, se)]
                assert all((x.data_ptr() % 16) == (4 * shift) % 16 for x in t)
                hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
                for fl in (0, FLAG_SORTED, FLAG_BUCKET):
                    hits.zero_()
                    db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(si), hits.data_ptr(), None,
                                  v=0, stream=stream.cuda_stream, flags=fl)
                    db.sync(stream.cuda_stream)
                    np.testing.assert_array_equal(hits.cpu().numpy(), want, err_msg="shift %d flags %d" % (shift, fl))
    finally:
        db.close()
        orc.close()


def test_blocking_api_repairs_a_wrong_order_promise(workdir):
    """igd_hip_search_ex(..., IGD_HIP_FLAG_SORTED) on UNSORTED host arrays: the device reports the
    broken promise, the call repeats the slice in auto mode -- counts are right, never doubled."""
    from igd_amd import Database
    rng = random.Random(404)
    nbp = 1 << 12
    path, ctgs, span = _random_db(rng, workdir, "fix", nbp, 1, 7, 2, 25, 70, 0)
    orc = Oracle(path)
    db = Database(path)
    try:
        ichr, qs, qe = _random_queries(rng, [0, 1], nbp, span, 2500)
        for v in (0, 400):
            want, wtot = orc.search(ichr, qs, qe, v)
            got, gtot = db.search(ichr, qs, qe, v, flags=FLAG_SORTED)
            assert gtot == wtot
            np.testing.assert_array_equal(got, want)
            si, ss, se = _sorted(ichr, qs, qe)
            got, gtot = db.search(si, ss, se, v, flags=FLAG_SORTED)
            assert gtot == wtot
            np.testing.assert_array_equal(got, want)
    finally:
        db.close()
        orc.close()


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from igd_amd import Database
d = np.load(sys.argv[3])
db = Database(sys.argv[2])
out = []
for tag in ("few", "many", "lumpy"):
    for v in (0, 300):
        for rule_flags in (2, 2 | 4):
            got, tot = db.search(d["c_" + tag], d["s_" + tag], d["e_" + tag], v, flags=rule_flags)
            out.append(got); out.append(np.array([tot], np.int64))
db.close()
np.save(sys.argv[4], np.concatenate(out))
"""


@pytest.mark.parametrize("switch", ["", "IGD_HIP_SPLIT_NOSTAGE", "IGD_HIP_SPLIT_NOBITS", "IGD_HIP_SPLIT_NOREGION"])
def test_unordered_batches_through_every_form_of_the_grouping_kernels(switch, workdir):
    """Round 5's grouping kernels put their output together in LDS -- the tile-has-records bits and contig tables, a workgroup's
    region of tuples (k_split_local), a coarse bucket's pairs (k_split_fine_a) -- and each has the older form behind it for what
    does not fit: a workgroup with more pairs than its area holds (many queries of 2-4 tiles), a bucket with more pairs than its
    area, a database of more tiles than the bits have room for.  Three unordered batches over an hg38-sized tile table -- one of
    short queries (everything staged), one where 40 % of the queries span 2-4 tiles (regions overflow), one with half of its
    queries on chr1 (buckets overflow) -- under
    IGD_HIP_FLAG_BUCKET, compact image and exact arrays, with and without a value filter, must give the oracle's counts in the
    shipped form and with each stage switched off (the switches are read once per process: child processes)."""
    import os, subprocess, sys
    from igd_amd import synth
    from helpers import ROOT
    p = os.path.join(workdir, "hgsmall.igd")
    if not os.path.exists(p):
        synth.make_db(p, files=12, per_file=6000, seed=5, genome=synth.HG38)
    rng = np.random.default_rng(99)
    sets = {}
    for tag, frac in (("few", 0.02), ("many", 0.40)):
        c, s, e = synth.make_queries(150000, seed=11 if tag == "few" else 12, genome=synth.HG38, sorted_=False)
        long_ = rng.random(len(s)) < frac
        e = np.where(long_, s + rng.integers(16384, 3 * 16384 + 8000, len(s)), e).astype(np.int32)
        sets[tag] = (c, s, e)
    # ... and one whose coarse buckets on chr1 hold 7 x the average: more pairs than the bucket's staging area, in short segments
    c, s, e = synth.make_queries(600000, seed=13, genome=synth.HG38, sorted_=False)
    hot = rng.random(len(s)) < 0.5
    hs = rng.integers(0, 248000000 - 4000, len(s))
    c = np.where(hot, 0, c).astype(np.int32)
    e = np.where(hot, hs + (e - s), e).astype(np.int32)
    s = np.where(hot, hs, s).astype(np.int32)
    sets["lumpy"] = (c, s, e)
    orc = Oracle(p)
    want = []
    try:
        for tag in ("few", "many", "lumpy"):
            for v in (0, 300):
                h, t = orc.search(*sets[tag], v)
                for _ in range(2):
                    want.append(h); want.append(np.array([t], np.int64))
    finally:
        orc.close()
    want = np.concatenate(want)
    qf = os.path.join(workdir, "gq.npz")
    np.savez(qf, **{"%s_%s" % (k, tag): a for tag, (c, s, e) in sets.items() for k, a in (("c", c), ("s", s), ("e", e))})
    outf = os.path.join(workdir, "gout_%s.npy" % (switch or "shipped"))
    env = dict(os.environ)
    if switch:
        env[switch] = "1"
    subprocess.check_call([sys.executable, "-c", _CHILD, ROOT, p, qf, outf], env=env, timeout=600)
    np.testing.assert_array_equal(np.load(outf), want, err_msg=switch or "shipped")
