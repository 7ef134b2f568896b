"""GPU: the DIRECT step of dense position-sorted batches (engine/scan_direct.hpp: k_tile_bounds -> igd_scan_direct ->
k_reduce_slabs -- no per-query pre-pass, the scan kernel reads q_qs / q_qe itself, pushes the later tiles and verifies the
order where it reads the queries) against the oracle:

  - every awkward query kind on small databases (inverted, zero-length, negative starts, unknown contigs, starts beyond the
    contig's last tile, queries of many tiles -> WALK_REST / WALK_LAST / coverage, queries that reach back over their tile's
    start -> WALK_FIRST), rules NEST and FLAT, the value filter, gType 0, multi-chunk tiles, a sparse database whose empty
    first tiles end queries under rule NEST only (quirk #1);
  - tiles whose next tile holds more records than ride along (IGD_D_APP = 64): queries beyond what they cover are walked;
  - a tile with more queries than one wave takes (IGD_HEAVY_FIRST): slices in the batch's last launch;
  - the run-table form of the batch; alternating batches on one handle (the exact-walk lists' parity);
  - broken promises: contigs out of order, starts out of order inside a tile and across tiles -> IGD_HIP_ERR_UNSORTED,
    nothing added, the handle fine afterwards;
  - the selection: IGD_HIP_FLAG_SORTED | IGD_HIP_FLAG_SHORT on a dense batch takes the DIRECT step, anything else does not;
    a batch that breaks the SHORT promise is still exact.
"""
import os
import random
import shutil

import numpy as np
import pytest

from helpers import Oracle, short_tmpdir, write_igd_numpy
from test_gpu_parity import CASES, _random_db, _random_queries
from test_gpu_rank import _dense_queries

pytestmark = pytest.mark.gpu
FLAG_SORTED, FLAG_SHORT = 1, 16
# the DIRECT step's scan kernel (one name today; a tuple so that a second form of the step can be tried without touching the tests)
DIRECT_KERNELS = ("igd_scan_direct",)


@pytest.fixture(scope="module")
def workdir():
    d = short_tmpdir("igd")
    yield d
    shutil.rmtree(d, ignore_errors=True)


def _dev_search(db, torch, ichr, qs, qe, v=0, rule=None, flags=FLAG_SORTED, runs=None, hits=None):
    """one resident batch; returns (hits, total, kernel name)"""
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(dev) for x in (ichr, qs, qe)]
    d_hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev) if hits is None else hits
    d_tot = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    kw = dict(v=v, flags=flags) if rule is None else dict(rule=rule, flags=flags)
    if runs is not None:
        d_runs = torch.from_numpy(np.ascontiguousarray(runs, dtype=np.int32)).to(dev)
        db.search_runs_dev(d_runs.data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(qs), d_hits.data_ptr(), d_tot.data_ptr(), **kw)
    else:
        db.search_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(qs), d_hits.data_ptr(), d_tot.data_ptr(), **kw)
    db.sync()
    return d_hits.cpu().numpy(), int(d_tot.item()), db.last_scan_kernel()


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4, 5])
def test_forced_direct_step_matches_the_oracle_on_every_query_kind(case, workdir, monkeypatch):
    import torch
    from igd_amd import Database
    monkeypatch.setenv("IGD_HIP_DIRECT", "1")
    monkeypatch.setenv("IGD_HIP_NO_RETILE", "1")         # (tiles of 2^11 / 2^12 bp searched as they are: a re-tiled copy takes the ordinary step)
    rng = random.Random(7100 + case)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[case]
    path, ctgs, span = _random_db(rng, workdir, "d%d" % case, nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc, db = Oracle(path), Database(path)
    try:
        for n in (70000, 900, 40):                               # dense, sparse, nearly nothing
            ichr, qs, qe = _dense_queries(rng, nctg, nbp, span, n)
            for v in (0, 300):
                want, wtot = orc.search(ichr, qs, qe, v)
                got, gtot, kern = _dev_search(db, torch, ichr, qs, qe, v=v)
                assert kern in DIRECT_KERNELS
                assert gtot == wtot, (case, n, v)
                np.testing.assert_array_equal(got, want, err_msg="case %d n %d v %d" % (case, n, v))
            # rule FLAT without a value filter: later tiles count behind an empty first tile
            ref = db.search(ichr, qs, qe, rule=1, flags=2)[0]           # (the bucket path, checked against the oracle elsewhere)
            got, _, kern = _dev_search(db, torch, ichr, qs, qe, rule=1)
            assert kern in DIRECT_KERNELS
            np.testing.assert_array_equal(got, ref)
            # the run-table form of the same batch (known contigs only)
            ok = (ichr >= 0) & (ichr < nctg)
            a, b, c = ichr[ok], qs[ok], qe[ok]
            runs = Database.contig_runs(a, nctg)
            want, wtot = orc.search(a, b, c, 0)
            got, gtot, kern = _dev_search(db, torch, a, b, c, runs=runs)
            assert kern in DIRECT_KERNELS and gtot == wtot
            np.testing.assert_array_equal(got, want)
    finally:
        db.close(); orc.close()


def test_short_queries_next_tile_fuller_than_what_rides_along(workdir, monkeypatch):
    """Tiles of a few hundred records: the 64 records of tile t+1 that ride with tile t's first unit cover only its first
    stretch, so short queries reaching further into t+1 are listed for the exact walk (WALK_REST) while the others are
    served by the appended records -- both kinds in every tile, under both rules."""
    import torch
    from igd_amd import Database
    monkeypatch.setenv("IGD_HIP_DIRECT", "1")
    monkeypatch.setenv("IGD_HIP_NO_RETILE", "1")         # (tiles of 2^11 / 2^12 bp searched as they are: a re-tiled copy takes the ordinary step)
    rng = random.Random(99)
    nbp = 1 << 12
    files = [[("chr1", s, s + rng.randint(1, 900), rng.randint(0, 1000)) for s in (rng.randrange(0, nbp * 30) for _ in range(900))]
             for _ in range(9)]                                  # ~270 records per tile, one or two chunks
    path = os.path.join(workdir, "full.igd")
    write_igd_numpy(path, files, nbp=nbp, gtype=1)
    orc, db = Oracle(path), Database(path)
    try:
        n = 60000
        qs = np.sort(np.array([rng.randrange(0, nbp * 31) for _ in range(n)], np.int32))
        qe = (qs + np.array([rng.choice([1, 30, 200, 700, 1500, nbp - 1, nbp + 9]) for _ in range(n)], np.int32)).astype(np.int32)
        ichr = np.zeros(n, np.int32)
        for v in (0, 400):
            want, wtot = orc.search(ichr, qs, qe, v)
            got, gtot, kern = _dev_search(db, torch, ichr, qs, qe, v=v)
            assert kern in DIRECT_KERNELS and gtot == wtot
            np.testing.assert_array_equal(got, want)
    finally:
        db.close(); orc.close()


def test_a_tile_with_more_queries_than_one_wave_takes(workdir, monkeypatch):
    """30 000 queries inside ONE tile (> IGD_HEAVY_FIRST = 8192): listed by igd_scan_direct, counted in slices by the batch's
    last launch -- including the slices' own order checks, their long / inverted queries (walked there and then) and the
    records of the next tile."""
    import torch
    from igd_amd import Database
    monkeypatch.setenv("IGD_HIP_DIRECT", "1")
    monkeypatch.setenv("IGD_HIP_NO_RETILE", "1")         # (tiles of 2^11 / 2^12 bp searched as they are: a re-tiled copy takes the ordinary step)
    rng = random.Random(5)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[3]
    path, ctgs, span = _random_db(rng, workdir, "hv", nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc, db = Oracle(path), Database(path)
    try:
        n = 30000
        qs = np.sort(np.array([5 * nbp + rng.randrange(0, nbp) for _ in range(n)], np.int32))       # the hot tile (multi-chunk)
        lens = np.array([rng.choice([0, 1, 40, nbp // 3, nbp, 2 * nbp + 5, 7 * nbp, -rng.randint(1, 300), -nbp]) for _ in range(n)], np.int32)
        qe = (qs + lens).astype(np.int32)
        more = _dense_queries(rng, nctg, nbp, span, 20000)
        ichr = np.concatenate([np.zeros(n, np.int32), more[0]])
        qs2, qe2 = np.concatenate([qs, more[1]]), np.concatenate([qe, more[2]])
        order = np.lexsort((qs2, ichr))
        ichr, qs2, qe2 = ichr[order], qs2[order], qe2[order]
        for v in (0, 300):
            want, wtot = orc.search(ichr, qs2, qe2, v)
            got, gtot, kern = _dev_search(db, torch, ichr, qs2, qe2, v=v)
            assert kern in DIRECT_KERNELS and gtot == wtot, v
            np.testing.assert_array_equal(got, want)
    finally:
        db.close(); orc.close()


def test_alternating_batches_and_accumulation_on_one_handle(workdir, monkeypatch):
    """The exact-walk list and the coverage arrays are per batch parity: long-query batches and short ones in turn, DIRECT and
    ordinary steps in turn, hits[] accumulating across them."""
    import torch
    from igd_amd import Database
    monkeypatch.setenv("IGD_HIP_DIRECT", "1")
    monkeypatch.setenv("IGD_HIP_NO_RETILE", "1")         # (tiles of 2^11 / 2^12 bp searched as they are: a re-tiled copy takes the ordinary step)
    rng = random.Random(77)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[5]
    path, ctgs, span = _random_db(rng, workdir, "alt", nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc, db = Oracle(path), Database(path)
    try:
        dev = torch.device("cuda", 0)
        acc = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
        want_acc = np.zeros(db.nfiles, np.int64)
        for k in range(7):
            n = (50000, 3000, 70000)[k % 3]
            ichr, qs, qe = _dense_queries(rng, nctg, nbp, span, n)
            flags = FLAG_SORTED if k % 4 != 3 else 0             # (flags 0: the device decides -- never the DIRECT step)
            got, _, kern = _dev_search(db, torch, ichr, qs, qe, flags=flags, hits=acc)
            assert (kern in DIRECT_KERNELS) == (flags == FLAG_SORTED)
            want_acc += orc.search(ichr, qs, qe, 0)[0]
            np.testing.assert_array_equal(got, want_acc, err_msg="batch %d" % k)
    finally:
        db.close(); orc.close()


def test_broken_promises_add_nothing_and_leave_the_handle_usable(workdir, monkeypatch):
    import torch
    from igd_amd import Database
    from igd_amd.database import IgdError
    monkeypatch.setenv("IGD_HIP_DIRECT", "1")
    monkeypatch.setenv("IGD_HIP_NO_RETILE", "1")         # (tiles of 2^11 / 2^12 bp searched as they are: a re-tiled copy takes the ordinary step)
    rng = random.Random(3)
    nbp, gtype, nfiles, nctg, span_tiles, dens, hot = CASES[5]
    path, ctgs, span = _random_db(rng, workdir, "bp", nbp, gtype, nfiles, nctg, span_tiles, dens, hot)
    orc, db = Oracle(path), Database(path)
    try:
        ichr, qs, qe = _dense_queries(rng, nctg, nbp, span, 60000)
        ok = (ichr >= 0) & (ichr < nctg) & (qs >= 0)
        ichr, qs, qe = ichr[ok], qs[ok], qe[ok]
        want, wtot = orc.search(ichr, qs, qe, 0)
        dev = torch.device("cuda", 0)
        n = len(qs)

        def swapped(i, k):
            p = np.arange(n); p[i], p[k] = p[k], p[i]
            return ichr[p], qs[p], qe[p]
        same_tile = [i for i in range(1000, n - 1) if ichr[i] == ichr[i + 1] and qs[i] // nbp == qs[i + 1] // nbp and qs[i] != qs[i + 1]][0]
        other_tile = [i for i in range(1000, n - 400) if ichr[i] == ichr[i + 300] and qs[i] // nbp != qs[i + 300] // nbp][0]
        other_ctg = [i for i in range(n - 1) if ichr[i] != ichr[i + 1]][0]
        for what, (a, b, c) in (("inside a tile", swapped(same_tile, same_tile + 1)), ("across tiles", swapped(other_tile, other_tile + 300)),
                                ("contigs", swapped(other_ctg, other_ctg + 1))):
            hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
            with pytest.raises(IgdError):
                _dev_search(db, torch, a, b, c, hits=hits)
            assert int(hits.sum().item()) == 0, what                 # a broken batch adds nothing
            got, gtot, _ = _dev_search(db, torch, ichr, qs, qe)
            assert gtot == wtot, what
            np.testing.assert_array_equal(got, want, err_msg=what)
            # ... and the host entry point redoes such a batch with the device choosing the grouping
            h2, t2 = db.search(a, b, c, 0, flags=FLAG_SORTED)
            np.testing.assert_array_equal(h2, want)
    finally:
        db.close(); orc.close()


def test_broken_start_order_inside_a_heavy_tile_adds_nothing(workdir, monkeypatch):
    """a tile with more than IGD_HEAVY_FIRST (8192) queries is left to the batch's last launch, whose slices
    add to the caller's hits[] with global atomics -- after k_reduce_slabs has looked at the order mark.  Two starts swapped
    INSIDE such a tile must be found before anything is counted (the bounds pass reads every start): the raw batch reports
    IGD_HIP_ERR_UNSORTED and adds nothing; Database.search repairs it exactly (no double count).  Same contract on the
    ordinary step (flags SORTED without SHORT on a sparse-on-average batch)."""
    import torch
    from igd_amd import Database, synth
    from igd_amd.database import IgdError
    path = os.path.join(workdir, "hv.igd")
    synth.make_db(path, files=20, per_file=4000, seed=9, nbp_log=14, genome=synth.SMALL)
    orc, db = Oracle(path), Database(path)
    try:
        dev = torch.device("cuda", 0)
        rng = np.random.default_rng(12)
        ntiles = sum(db.ntile)
        base = synth.make_queries(30 * ntiles, seed=5, genome=synth.SMALL, min_len=100, max_len=1999, sorted_=True)
        # + 20 000 queries inside tile 3 of contig 0: a heavy tile
        ps = (3 * 16384 + rng.integers(0, 16384, 20000)).astype(np.int32)
        ichr = np.concatenate([base[0], np.zeros(20000, np.int32)])
        qs = np.concatenate([base[1], ps])
        qe = np.concatenate([base[2], (ps + rng.integers(100, 1999, 20000)).astype(np.int32)])
        o = np.lexsort((qs, ichr))
        ichr, qs, qe = ichr[o], qs[o], qe[o]
        want, wtot = orc.search(ichr, qs, qe, 0)
        got, gtot, kern = _dev_search(db, torch, ichr, qs, qe, flags=FLAG_SORTED | FLAG_SHORT)
        assert kern in DIRECT_KERNELS and gtot == wtot
        np.testing.assert_array_equal(got, want)
        inside = np.flatnonzero((ichr == 0) & (qs // 16384 == 3))
        assert len(inside) > 8192 * 2
        for where in (inside[10], inside[len(inside) // 2], inside[-3]):        # first slice, a middle slice, the last one
            k = where + 1
            while qs[k] == qs[where]:
                k += 1
            assert ichr[k] == 0 and qs[k] // 16384 == 3
            p = np.arange(len(qs)); p[where], p[k] = p[k], p[where]
            a, b, c = ichr[p], qs[p], qe[p]
            for flags in (FLAG_SORTED | FLAG_SHORT, FLAG_SORTED):
                hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
                with pytest.raises(IgdError):
                    _dev_search(db, torch, a, b, c, flags=flags, hits=hits)
                assert int(hits.sum().item()) == 0, (where, flags)
                h2, t2 = db.search(a, b, c, 0, flags=flags)
                assert t2 == wtot
                np.testing.assert_array_equal(h2, want)
            # unpromised, the same batch is an ordinary merge-join batch (pairwise compares; CTL_NOTSTART)
            got, gtot, _ = _dev_search(db, torch, a, b, c, flags=0)
            assert gtot == wtot
            np.testing.assert_array_equal(got, want)
    finally:
        db.close(); orc.close()


def test_the_engine_picks_the_direct_step_for_dense_short_sorted_batches_only(workdir):
    import torch
    from igd_amd import Database, synth
    path = os.path.join(workdir, "sel.igd")
    synth.make_db(path, files=30, per_file=20000, seed=4, nbp_log=14, genome=synth.SMALL)
    orc, db = Oracle(path), Database(path)
    try:
        ntiles = sum(db.ntile)
        dense = synth.make_queries(40 * ntiles, seed=7, genome=synth.SMALL, min_len=100, max_len=1999, sorted_=True)
        sparse = synth.make_queries(5 * ntiles, seed=8, genome=synth.SMALL, min_len=100, max_len=1999, sorted_=True)
        longer = synth.make_queries(40 * ntiles, seed=9, genome=synth.SMALL, min_len=100, max_len=60000, sorted_=True)   # breaks SHORT
        for (q, flags, direct) in ((dense, FLAG_SORTED | FLAG_SHORT, True), (dense, FLAG_SORTED, False), (dense, 0, False),
                                   (sparse, FLAG_SORTED | FLAG_SHORT, False), (longer, FLAG_SORTED | FLAG_SHORT, True)):
            for v in (0, 500):
                want, wtot = orc.search(*q, v)
                got, gtot, kern = _dev_search(db, torch, *q, v=v, flags=flags)
                assert (kern in DIRECT_KERNELS) == direct, (flags, kern)
                assert gtot == wtot
                np.testing.assert_array_equal(got, want)
    finally:
        db.close(); orc.close()


from helpers import ROOT                                    # noqa: E402
from test_golden_oracle import CASES as GOLDEN_CASES, materialize      # noqa: E402


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_golden_command_lines_through_the_direct_step(case):
    """`bin/igd search ... -q` on the golden fixtures with every sorted file sent to the engine's DIRECT step (IGD_HIP_DIRECT=1), whole
    and in device batches of 37 queries: stdout byte-identical to what the REAL reference printed (tests/golden/*/outNN.txt) --
    edge cases, quirk #1, the < 16 branch, parse rules, gType 0, -v N.  (Unsorted query files are redone unpromised by the host
    entry point, as always; `-f` and `-r` do not touch the step.)"""
    import subprocess
    exe = os.path.join(ROOT, "bin", "igd")
    d, dst, man = materialize(case)
    try:
        n = 0
        for run in man["runs"]:
            if "-q" not in run["args"] or "-f" in run["args"]:
                continue
            args = [os.path.join(dst, a) if a in ("db.igd", "q.bed", "q.bed.gz", "q100.bed") else a for a in run["args"]]
            want = open(os.path.join(dst, run["stdout"])).read()
            for limit in ((0,) if case == "config1" else (0, 37)):
                env = dict(os.environ, IGD_HIP_DIRECT="1", IGD_HIP_NO_RETILE="1", IGD_HOST_MAX_QUERIES="0", IGD_TIMING="1")
                if limit:
                    env["IGD_HIP_MAX_BATCH"] = str(limit)
                p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
                assert p.returncode == 0, p.stderr.decode()[-300:]
                assert p.stdout.decode() == want, (case, run["args"], limit)
            n += 1
        assert n > 0
    finally:
        shutil.rmtree(d, ignore_errors=True)
