"""GPU: position-sorted batches given as contig RUNS (igd_hip_search_runs_dev: run_start[nCtg + 1] instead of one contig number
per query -- k_query_bounds' RUNS build reads 8 of the 12 bytes per query).  Every count against the oracle and against the
ichr[] form of the same batch:

  - a large batch on the synthetic genome (four queries per thread, compact image: the RUNS build itself), with contigs
    that have no query at all (empty runs at the front, in the middle, at the end), long queries (the exact walk finds the
    contig in k_query_bounds' list entry), inverted ones, a dense region (rank method);
  - the batches the RUNS build does not take -- fewer than 65 536 queries, the exact arrays, a tile width that is no power
    of two -- whose contig numbers are written out first (k_expand_runs);
  - a run table that is not one (not monotone / does not end at nq) and queries out of order inside a run: a broken
    promise, reported by igd_hip_sync, nothing added.
"""
import os
import random
import shutil

import numpy as np
import pytest

from helpers import Oracle, short_tmpdir, write_igd_numpy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir():
    d = short_tmpdir("igu")
    yield d
    shutil.rmtree(d, ignore_errors=True)


def _run(db, torch, runs, qs, qe, v=0, flags=0):
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        t = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(dev) for x in (runs, qs, qe)]
        hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
        tot = torch.zeros(1, dtype=torch.int64, device=dev)
        stream.synchronize()
        db.search_runs_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(qs), hits.data_ptr(), tot.data_ptr(), v=v,
                           stream=stream.cuda_stream, flags=flags)
        db.sync(stream.cuda_stream)
        return hits.cpu().numpy(), int(tot.item())


def _queries(synth, n, seed, genome, drop=(), **kw):
    ichr, qs, qe = synth.make_queries(n, seed=seed, genome=genome, sorted_=True, **kw)
    keep = ~np.isin(ichr, list(drop)) & (ichr >= 0)
    return ichr[keep], qs[keep], qe[keep]


def test_large_batches_as_runs_equal_the_oracle_and_the_ichr_form(workdir):
    import torch
    from igd_amd import Database, synth
    path = os.path.join(workdir, "r.igd")
    synth.make_db(path, files=40, per_file=20000, seed=9, nbp_log=12, genome=synth.HG38)
    db, orc = Database(path), Oracle(path)
    try:
        nctg = db.nctg
        cases = [
            dict(n=300000, seed=3, drop=()),                                      # every contig has queries
            dict(n=300000, seed=4, drop=(0, 1, 7, nctg - 1)),                     # empty runs: front, middle, end
            dict(n=200000, seed=5, drop=(2,), min_len=1, max_len=40000),          # up to ten tiles long: exact walks + coverage
        ]
        for c in cases:
            kw = {k: c[k] for k in c if k not in ("n", "seed", "drop")}
            ichr, qs, qe = _queries(synth, c["n"], c["seed"], synth.HG38, c["drop"], **kw)
            if c["seed"] == 5:                                                     # some inverted ones (qe < qs), order by start kept
                inv = np.arange(len(qs)) % 97 == 0
                qe = np.where(inv, qs - 50, qe).astype(np.int32)
            assert len(qs) >= 65536
            runs = Database.contig_runs(ichr, nctg)
            assert runs[0] == 0 and runs[-1] == len(qs)
            for v in (0, 500):
                want, wtot = orc.search(ichr, qs, qe, v)
                got, gtot = _run(db, torch, runs, qs, qe, v=v)
                assert gtot == wtot, (c, v)
                np.testing.assert_array_equal(got, want, err_msg=str((c, v)))
                same, _ = db.search(ichr, qs, qe, v, flags=1)
                np.testing.assert_array_equal(got, same)
        # a dense region: 200 000 queries inside 300 tiles of one contig (the rank method), the other runs empty
        rng = np.random.default_rng(1)
        qs = np.sort(rng.integers(4096 * 100, 4096 * 400, 200000)).astype(np.int32)
        qe = (qs + rng.integers(1, 3000, len(qs))).astype(np.int32)
        ichr = np.full(len(qs), 3, np.int32)
        runs = Database.contig_runs(ichr, nctg)
        want, wtot = orc.search(ichr, qs, qe, 0)
        got, gtot = _run(db, torch, runs, qs, qe)
        assert gtot == wtot
        np.testing.assert_array_equal(got, want)
    finally:
        db.close(); orc.close()


def test_batches_the_runs_build_does_not_take(workdir):
    import torch
    from igd_amd import Database, synth
    rng = random.Random(5)
    # (a) a small batch and (b) the exact arrays, on the compact-image database; (c) a tile width that is no power of two
    path = os.path.join(workdir, "s.igd")
    synth.make_db(path, files=12, per_file=5000, seed=2, genome=synth.SMALL)
    files = [[("chr%d" % (1 + rng.randrange(3)), s, s + rng.randint(1, 9000), rng.randint(0, 1000))
              for s in (rng.randrange(0, 3000 * 40) for _ in range(400))] for _ in range(7)]
    path3 = os.path.join(workdir, "w3000.igd")
    write_igd_numpy(path3, files, nbp=3000, gtype=1, contig_order=["chr1", "chr2", "chr3"])
    for p, genome_kw, n, flags in ((path, dict(genome=synth.SMALL), 20000, 0), (path, dict(genome=synth.SMALL), 200000, 4)):
        db, orc = Database(p), Oracle(p)
        try:
            ichr, qs, qe = _queries(synth, n, 8, synth.SMALL, (1,), min_len=1, max_len=50000)
            runs = Database.contig_runs(ichr, db.nctg)
            for v in (0, 300):
                want, wtot = orc.search(ichr, qs, qe, v)
                got, gtot = _run(db, torch, runs, qs, qe, v=v, flags=flags)
                assert gtot == wtot
                np.testing.assert_array_equal(got, want)
        finally:
            db.close(); orc.close()
    db, orc = Database(path3), Oracle(path3)
    try:
        r = np.random.default_rng(3)
        ichr = np.sort(r.integers(0, 3, 100000)).astype(np.int32)
        qs = r.integers(0, 3000 * 42, len(ichr)).astype(np.int32)
        order = np.lexsort((qs, ichr))
        ichr, qs = ichr[order], qs[order]
        qe = (qs + r.integers(1, 7000, len(qs))).astype(np.int32)
        runs = Database.contig_runs(ichr, db.nctg)
        want, wtot = orc.search(ichr, qs, qe, 0)
        got, gtot = _run(db, torch, runs, qs, qe)
        assert gtot == wtot
        np.testing.assert_array_equal(got, want)
    finally:
        db.close(); orc.close()


def test_a_run_table_that_is_not_one_is_a_broken_promise(workdir):
    import torch
    from igd_amd import Database, synth
    from igd_amd.database import IgdError
    path = os.path.join(workdir, "b.igd")
    synth.make_db(path, files=10, per_file=8000, seed=6, nbp_log=12, genome=synth.HG38)
    db, orc = Database(path), Oracle(path)
    try:
        ichr, qs, qe = _queries(synth, 150000, 12, synth.HG38)
        runs = Database.contig_runs(ichr, db.nctg)
        good, gtot = _run(db, torch, runs, qs, qe)
        want, wtot = orc.search(ichr, qs, qe, 0)
        assert gtot == wtot
        np.testing.assert_array_equal(good, want)
        bad1 = runs.copy(); bad1[5], bad1[6] = bad1[6] + 10, bad1[5]          # not monotone
        bad2 = runs.copy(); bad2[-1] -= 1                                      # does not cover [0, nq)
        bad3 = runs.copy(); bad3[0] = 1
        for bad in (bad1, bad2, bad3):
            with pytest.raises(IgdError):
                _run(db, torch, bad, qs, qe)
        p = np.arange(len(qs)); p[70000], p[70001 + 300] = p[70001 + 300], p[70000]   # starts out of order inside a run
        if ichr[70000] == ichr[70301] and qs[70000] != qs[70301]:
            with pytest.raises(IgdError):
                _run(db, torch, runs, qs[p], qe[p])
        # ... and the handle is fine afterwards
        again, _ = _run(db, torch, runs, qs, qe)
        np.testing.assert_array_equal(again, want)
    finally:
        db.close(); orc.close()


def test_the_batch_right_after_a_bad_run_table_counts_exactly(workdir):
    """a malformed run table made every workgroup of k_query_bounds leave before the next batch's list counters
    were reset, so the batch after it appended to the stale exact-walk list of two batches ago.  Good batch WITH long
    queries (a non-empty list), ONE bad table, then at once a smaller good batch with long queries -- against the oracle."""
    import torch
    from igd_amd import Database, synth
    from igd_amd.database import IgdError
    path = os.path.join(workdir, "b2.igd")
    synth.make_db(path, files=10, per_file=8000, seed=6, nbp_log=12, genome=synth.HG38)
    db, orc = Database(path), Oracle(path)
    try:
        nctg = db.nctg
        big = _queries(synth, 200000, 21, synth.HG38, min_len=1, max_len=60000)       # up to 15 tiles long: walks + coverage
        small = _queries(synth, 70000, 22, synth.HG38, min_len=1, max_len=60000)
        for rounds in range(2):                                                        # both batch parities meet the bad table
            for first, second in ((big, small), (small, big)):
                runs1 = Database.contig_runs(first[0], nctg)
                got, gtot = _run(db, torch, runs1, first[1], first[2])
                want, wtot = orc.search(*first, 0)
                assert gtot == wtot
                np.testing.assert_array_equal(got, want)
                bad = runs1.copy(); bad[3], bad[4] = bad[4] + 7, bad[3]
                with pytest.raises(IgdError):
                    _run(db, torch, bad, first[1], first[2])
                runs2 = Database.contig_runs(second[0], nctg)
                got, gtot = _run(db, torch, runs2, second[1], second[2])
                want, wtot = orc.search(*second, 0)
                assert gtot == wtot, "the batch after a bad run table"
                np.testing.assert_array_equal(got, want)
            # an odd number of batches in between flips the parity the bad table meets
            _run(db, torch, Database.contig_runs(small[0], nctg), small[1], small[2])
    finally:
        db.close(); orc.close()
