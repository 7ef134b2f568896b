"""GPU: the seams between device batches.  The host-buffer entry points cut a query set into engine calls of at most
igd_hip_max_batch() queries (2^24): the loop of igd_hip_search_ex (engine/host_search.hpp), the `-f` loop over
igd_hip_enumerate_stream and the `-s` loop over igd_hip_seqpare_add (igd_cli_abi.c).  No fixture has 1.7e7 queries, so

  - the TEST-ONLY variable IGD_HIP_MAX_BATCH lowers the limit and the golden command lines (the real reference's stdout:
    -q, -v N, -f, gzip, CRLF) and the Seqpare golden run again with batches of 37 and 1000 queries: every seam is crossed
    hundreds of times, output byte-identical;
  - the engine's host-buffer search is compared with the oracle under a limit of 999 queries, accumulating `hits`,
    with the order promise kept and broken inside a later batch;
  - ONE real `bin/igd search -q` on 2 x 10^7 position-sorted queries (two device batches at the production limit)
    against the oracle's committed total and checksum (tests/golden/bench_checksums.json: cli_sorted_q20000000);
  - slab 3 and slab 7 of the 10^8-query set of BASELINE config 4 (what ranks 3 and 7 of the 8-GPU job run) against the
    oracle's committed checksums -- config 4 is not pinned on slab 0 alone.
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, parse_hits_table
from test_golden_oracle import CASES, materialize

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "bin", "igd")


def _env(limit):
    return dict(os.environ, IGD_HIP_MAX_BATCH=str(limit), IGD_HOST_MAX_QUERIES="0")


@pytest.mark.parametrize("case", CASES)
def test_golden_command_lines_in_batches_of_37_and_1000_queries(case):
    d, dst, man = materialize(case)
    try:
        n = 0
        for run in man["runs"]:
            if "-q" not in run["args"]:
                continue
            args = [os.path.join(dst, a) if a in ("db.igd", "q.bed", "q.bed.gz", "q100.bed") else a for a in run["args"]]
            want = open(os.path.join(dst, run["stdout"])).read()
            for limit in ((1000,) if case == "config1" else (37, 1000)):
                p = subprocess.run([EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_env(limit), timeout=600)
                assert p.returncode == 0, p.stderr.decode()[-300:]
                assert p.stdout.decode() == want, (case, run["args"], limit)
            n += 1
        assert n > 0
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_seqpare_golden_in_small_batches():
    g = os.path.join(GOLDEN, "create")
    want = open(g + "/search_s.txt").read()
    for limit in (50, 1000):
        p = subprocess.run([EXE, "search", g + "/ref.igd", "-q", g + "/q.bed", "-s"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=_env(limit), timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-300:]
        assert p.stdout.decode() == want, limit


def test_host_buffer_search_across_batches_equals_the_oracle():
    """igd_hip_search_ex under a 999-query limit: sorted (promise kept), sorted with a disorder inside the fourth batch
    (that slice is redone with the device grouping), unsorted; counts accumulate into the caller's hits[] over the
    batches and over two calls.  In a child process: the limit is read once per process."""
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from helpers import Oracle, short_tmpdir
from igd_amd import Database, synth, _native as N
assert N.hip().igd_hip_max_batch() == 999
d = short_tmpdir("igb")
path = os.path.join(d, "b.igd")
synth.make_db(path, files=20, per_file=6000, seed=3, genome=synth.SMALL)
db, orc = Database(path), Oracle(path)
ichr, qs, qe = synth.make_queries(7777, seed=5, genome=synth.SMALL, min_len=1, max_len=70000, unknown_every=41, extra_span=50000)
for v in (0, 500):
    want, wtot = orc.search(ichr, qs, qe, v)
    for flags in (1, 0, 2):
        got, gtot = db.search(ichr, qs, qe, v, flags=flags)
        assert gtot == wtot and np.array_equal(got, want), (v, flags)
    # a broken promise inside batch 3 (queries 2997..3995)
    p = np.arange(len(qs)); p[3100], p[3500] = p[3500], p[3100]
    got, gtot = db.search(ichr[p], qs[p], qe[p], v, flags=1)
    assert gtot == wtot and np.array_equal(got, want), ("swapped", v)
    rng = np.random.default_rng(2); p = rng.permutation(len(qs))
    got, gtot = db.search(ichr[p], qs[p], qe[p], v)
    assert gtot == wtot and np.array_equal(got, want), ("shuffled", v)
gq, gr = db.enumerate(ichr[:999], qs[:999], qe[:999])
wq, wr = orc.enumerate(ichr[:999], qs[:999], qe[:999])
assert np.array_equal(gq, wq) and np.array_equal(gr[:, 1:], wr)
print("BATCHES-OK")
''' % (ROOT, os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_env(999), timeout=900)
    assert p.returncode == 0 and b"BATCHES-OK" in p.stdout, (p.stdout.decode()[-500:], p.stderr.decode()[-1500:])


def _golden(key):
    w = json.load(open(os.path.join(GOLDEN, "bench_checksums.json")))["workloads"][key]
    return w["total"], w["checksum"]


def _checksum(h):
    return int((h.astype(np.uint64) * (np.arange(len(h), dtype=np.uint64) + np.uint64(1))).sum() & np.uint64((1 << 63) - 1))


def _roadmap():
    from igd_amd import synth
    path = "/tmp/igdb/rm1900x26316.igd"
    if not os.path.exists(path + ".done"):
        os.makedirs("/tmp/igdb", exist_ok=True)
        synth.make_db(path)
        open(path + ".done", "w").write("ok")
    return path


def test_cli_query_file_of_two_device_batches():
    """`bin/igd search -q` on 2 x 10^7 position-sorted queries: 16 777 216 + 3 222 784, the production limit."""
    from igd_amd import synth, _native as N
    assert N.hip().igd_hip_max_batch() == 1 << 24
    path = _roadmap()
    n = 20000000
    bed = "/tmp/igdb/q2e7.bed"
    ichr, qs, qe = synth.make_queries(n, seed=11, genome=synth.HG38, sorted_=True)
    synth.write_bed(bed, synth.HG38, ichr, qs, qe)
    del ichr, qs, qe
    try:
        env = dict(os.environ, IGD_HOST_MAX_QUERIES="0", IGD_TIMING="1")
        env.pop("IGD_HIP_MAX_BATCH", None)
        p = subprocess.run([EXE, "search", path, "-q", bed], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1200)
        assert p.returncode == 0, p.stderr.decode()[-500:]
        hits, total = parse_hits_table(p.stdout.decode(), 1900)
        assert (int(hits.sum()), _checksum(hits)) == _golden("cli_sorted_q20000000_v0")
        assert total is not None
    finally:
        os.unlink(bed)


@pytest.mark.parametrize("slab,vs", [(3, (0,)), (7, (0, 500))])
def test_config4_slabs_3_and_7_of_8_against_the_oracles_checksums(slab, vs):
    from igd_amd import Database, synth
    path = _roadmap()
    n = 12500000
    ichr, qs, qe = synth.make_queries_slab(8 * n, slab * n, (slab + 1) * n, seed=7, genome=synth.HG38)
    db = Database(path)
    try:
        for v in vs:
            for flags in (1, 0):
                got, gtot = db.search(ichr, qs, qe, v, flags=flags)
                assert (int(gtot), _checksum(got)) == _golden("config4_slab%d_of_8_v%d" % (slab, v)), (slab, v, flags)
                assert int(got.sum()) == gtot
    finally:
        db.close()
