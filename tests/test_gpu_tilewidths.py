"""GPU: databases bucketed with other tile widths than the image is made for.  The reference's `create` accepts -b 11..19
(src/igd_create.c:454-457); the engine's compact image holds tiles of 2^14 / 2^15 bp.  The counting searches of any other
power-of-two width run on a RE-TILED copy (the same records in tiles of 2^14 bp, igd_hip_open) and what depends on the FILE's
tiles is applied per query: a first tile outside the contig's tiles counts nothing (:462), and under rule NEST neither
does an empty first tile (:468) -- while an empty tile of the COPY means nothing.  Every count against the oracle:

  - sparse databases (about half of the file's tiles empty: the two rules differ, and differ from what the copy's tiles
    would say) and dense ones, -b 10 .. 19, both record types;
  - queries with starts in (-nbp, 0) (tile 0 of the file by C division), at and beyond the contig's last tile, inverted,
    zero-length, many tiles long; sorted under the promise, device-decides, bucket path, exact arrays, contig runs; -v;
  - `-f` and the hit map stay on the file's own tiles: compared as well.
"""
import os
import shutil

import numpy as np
import pytest

from helpers import Oracle, short_tmpdir

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir():
    d = short_tmpdir("igw")
    yield d
    shutil.rmtree(d, ignore_errors=True)


def _queries(synth, n, seed, nbp, rng):
    ichr, qs, qe = synth.make_queries(n, seed=seed, genome=synth.SMALL, min_len=1, max_len=3 * nbp if nbp < 70000 else 200000,
                                      sorted_=False, unknown_every=211, extra_span=2 * nbp)
    k = len(qs)
    sel = rng.choice(k, k // 40, replace=False)
    qs[sel] = -rng.integers(1, 2 * nbp, len(sel))                  # starts before the contig: (-nbp, 0) is tile 0, below is nothing
    qe[sel] = qs[sel] + rng.integers(1, 4 * nbp, len(sel))
    sel = rng.choice(k, k // 50, replace=False)
    qe[sel] = qs[sel] - rng.integers(0, 300, len(sel))              # inverted and zero-length
    sel = rng.choice(k, k // 60, replace=False)
    qe[sel] = qs[sel] + rng.integers(5 * 16384, 40 * 16384, len(sel))   # many tiles of the copy
    return ichr.astype(np.int32), qs.astype(np.int32), qe.astype(np.int32)


@pytest.mark.parametrize("b,files,per_file,gtype0", [(10, 6, 400, False), (11, 9, 3000, False), (12, 5, 200, False), (13, 12, 5000, True),
                                                       (16, 6, 300, False), (16, 20, 8000, False), (17, 7, 500, True), (18, 10, 6000, False),
                                                       (19, 6, 300, False)])
def test_counts_on_the_retiled_copy_equal_the_oracle(b, files, per_file, gtype0, workdir):
    import torch
    from igd_amd import Database, synth
    path = os.path.join(workdir, "w%d_%d.igd" % (b, files))
    synth.make_db(path, files=files, per_file=per_file, seed=40 + b, nbp_log=b, genome=synth.SMALL, gtype=0 if gtype0 else 1)
    db, orc = Database(path), Oracle(path)
    try:
        assert db.nbp == 1 << b
        empty = sum(1 for c in range(orc.nctg) for j in range(orc.lib.orc_ntile(orc.h, c)) if orc.lib.orc_ncnt(orc.h, c, j) == 0)
        rng = np.random.default_rng(b)
        for n in (3000, 90000):
            ichr, qs, qe = _queries(synth, n, 3 + b, 1 << b, rng)
            o = np.lexsort((qs, ichr))
            srt = (ichr[o], qs[o], qe[o])
            for v in (0, 400):
                want, wtot = orc.search(ichr, qs, qe, v)
                for q, flags in ((srt, 1), (srt, 0), ((ichr, qs, qe), 0), ((ichr, qs, qe), 2), (srt, 1 | 4), ((ichr, qs, qe), 2 | 4)):
                    got, gtot = db.search(*q, v, flags=flags)
                    assert gtot == wtot, (b, n, v, flags, empty)
                    np.testing.assert_array_equal(got, want, err_msg="b=%d n=%d v=%d flags=%d" % (b, n, v, flags))
        # the two rules differ on a sparse file (v = 1 takes rule FLAT with a filter that passes nearly everything)
        if per_file <= 500 and not gtype0:
            nest, _ = db.search(ichr, qs, qe, 0)
            flat, _ = db.search(ichr, qs, qe, 1)
            assert empty > 0 and nest.sum() < flat.sum()
        # contig runs (known contigs only)
        keep = (srt[0] >= 0) & (srt[0] < db.nctg)
        ic, s, e = (a[keep] for a in srt)
        runs = Database.contig_runs(ic, db.nctg)
        dev = torch.device("cuda", 0)
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            t = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (runs, s, e)]
            hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
            stream.synchronize()
            db.search_runs_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(s), hits.data_ptr(), None, v=0, stream=stream.cuda_stream)
            db.sync(stream.cuda_stream)
        np.testing.assert_array_equal(hits.cpu().numpy(), orc.search(ic, s, e, 0)[0])
        # what depends on the file's own tiles and record order: -f
        gq, gr = db.enumerate(srt[0][:2000], srt[1][:2000], srt[2][:2000])
        wq, wr = orc.enumerate(srt[0][:2000], srt[1][:2000], srt[2][:2000])
        np.testing.assert_array_equal(gq, wq)
        np.testing.assert_array_equal(gr[:, 1:], wr)
    finally:
        db.close(); orc.close()
