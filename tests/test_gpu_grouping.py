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

FLAG_SORTED, FLAG_BUCKET = 1, 2


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
            for flags in (0, FLAG_BUCKET):
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
