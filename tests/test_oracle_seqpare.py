"""Seqpare (`igd search db.igd -q f.bed -s`): CPU oracle vs the REAL reference,
complete stdout (similarity printed with %10.6f per dataset).  Inputs have many equal scores
(identical intervals in several queries and datasets), queries spanning tiles, duplicates, unknown
contigs and zero-length queries, so the greedy matching's tie-breaking is exercised."""
import os
import random
import shutil

import pytest

from helpers import ref_create, run_oracle_cli, run_ref, short_tmpdir
from test_oracle_create import write_beds

pytestmark = pytest.mark.ref


def write_queries(rng, path, n, nbp, span_tiles=40, dup=0.2):
    rows = []
    for i in range(n):
        if rows and rng.random() < dup:
            rows.append(rng.choice(rows))
            continue
        c = rng.choice(["chr1", "chr2", "chrX", "chr10", "chr7", "1"])
        m = rng.random()
        if m < 0.4:
            s = rng.randrange(0, nbp * span_tiles)
        elif m < 0.7:
            s = 5 * nbp + rng.randrange(0, 300)
        else:
            s = 7 * nbp + 256 * rng.randrange(0, 8) + rng.randrange(0, 3)
        L = rng.choice([0, 1, 7, nbp // 2, nbp, 3 * nbp + 5, rng.randrange(1, 2 * nbp)])
        rows.append("%s\t%d\t%d" % (c, s, s + L))
    open(path, "w").write("\n".join(rows) + "\n")


@pytest.mark.parametrize("seed,b,nfiles,n,nq", [(1, 12, 10, 120, 150), (2, 14, 12, 400, 400), (3, 11, 15, 60, 90),
                                                  (4, 13, 10, 800, 60), (5, 12, 11, 30, 500)])
def test_seqpare_stdout_is_identical(seed, b, nfiles, n, nq):
    rng = random.Random(seed)
    d = short_tmpdir()
    try:
        write_beds(rng, os.path.join(d, "in"), nfiles, n, 1 << b, 5)
        igd = ref_create(os.path.join(d, "in") + "/*", os.path.join(d, "o"), "db", b=b)
        q = os.path.join(d, "q.bed")
        write_queries(rng, q, nq, 1 << b)
        want = run_ref(["search", igd, "-q", q, "-s"])
        got = run_oracle_cli(["search", igd, "-q", q, "-s"])
        assert got == want
        assert any(float(l.split("\t")[2]) > 0 for l in want.splitlines()[1:])
    finally:
        shutil.rmtree(d, ignore_errors=True)
