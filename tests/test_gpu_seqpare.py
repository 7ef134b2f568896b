"""Seqpare (`igd search db.igd -q f.bed -s`) on the GPU against the CPU oracle
(pinned to the reference by tests/test_oracle_seqpare.py and tests/golden/create/search_s.txt):
the complete stdout must be identical -- one similarity per dataset printed with %10.6f, which only
comes out right if the single-precision similarities, the greedy matching's tie-breaking and the
order of the double additions all follow the reference."""
import ctypes as C
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ORACLE_BIN, ROOT, build_oracle, short_tmpdir
from test_oracle_create import write_beds
from test_oracle_seqpare import write_queries

pytestmark = pytest.mark.gpu
IGD_BIN = os.path.join(ROOT, "bin", "igd")


def both(igd, q):
    build_oracle()
    a = subprocess.run([IGD_BIN, "search", igd, "-q", q, "-s"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert a.returncode == 0, a.stderr.decode()[-500:]
    b = subprocess.run([ORACLE_BIN, "search", igd, "-q", q, "-s"], stdout=subprocess.PIPE, timeout=900, check=True)
    return a.stdout.decode(), b.stdout.decode()


def test_golden_reference_output():
    g = os.path.join(GOLDEN, "create")
    out = subprocess.run([IGD_BIN, "search", g + "/ref.igd", "-q", g + "/q.bed", "-s"], stdout=subprocess.PIPE, timeout=600).stdout.decode()
    assert out == open(g + "/search_s.txt").read()


@pytest.mark.parametrize("seed,b,nfiles,n,nq", [(1, 12, 10, 120, 150), (2, 14, 12, 400, 400), (3, 11, 15, 60, 90),
                                                  (4, 13, 3, 800, 60), (5, 12, 11, 30, 500), (6, 14, 40, 1000, 3000)])
def test_cli_seqpare_equals_oracle(seed, b, nfiles, n, nq):
    rng = random.Random(seed)
    d = short_tmpdir()
    try:
        write_beds(rng, os.path.join(d, "in"), nfiles, n, 1 << b, 5)
        subprocess.run([IGD_BIN, "create", d + "/in/", d + "/o", "db", "-b", str(b)], stdout=subprocess.PIPE, check=True, timeout=600)
        q = os.path.join(d, "q.bed")
        write_queries(rng, q, nq, 1 << b)
        got, want = both(d + "/o/db.igd", q)
        assert got == want
        assert any(float(l.split("\t")[2]) > 0 for l in want.splitlines()[1:])
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_groups_beyond_the_lds_capacity_and_identical_scores():
    """One dataset whose intervals pile up under thousands of queries: (contig, dataset) groups of far
    more than 1024 pairs (HBM scratch path), most scores equal (identical intervals) -> the order of
    equal candidates decides which pairs are matched."""
    rng = random.Random(9)
    d = short_tmpdir()
    try:
        os.makedirs(d + "/in")
        nbp = 1 << 14
        for f in range(3):
            rows = []
            for i in range(2500):
                s = 3 * nbp + rng.randrange(0, 40) * 50 + (0 if f else rng.randrange(0, 2))
                rows.append("chr1\t%d\t%d\tn\t%d" % (s, s + rng.choice([100, 100, 100, 250]), i))
            for i in range(200):
                s = rng.randrange(0, 40 * nbp)
                rows.append("chr2\t%d\t%d\tn\t1" % (s, s + rng.randrange(1, 2 * nbp)))
            open(d + "/in/f%d.bed" % f, "w").write("\n".join(rows) + "\n")
        subprocess.run([IGD_BIN, "create", d + "/in/", d + "/o", "db"], stdout=subprocess.PIPE, check=True, timeout=600)
        rows = []
        for i in range(3000):
            s = 3 * nbp + rng.randrange(0, 40) * 50
            rows.append("chr1\t%d\t%d" % (s, s + rng.choice([100, 100, 180])))
        for i in range(300):
            s = rng.randrange(0, 40 * nbp)
            rows.append("chr2\t%d\t%d" % (s, s + rng.randrange(0, 3 * nbp)))
        rng.shuffle(rows)
        open(d + "/q.bed", "w").write("\n".join(rows) + "\n")
        got, want = both(d + "/o/db.igd", d + "/q.bed")
        assert got == want
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_c_entry_point_seqOverlaps():
    """libigd.so: seqOverlaps(char *qFile, double *sm) after the reference's own set-up calls."""
    from igd_amd import _native
    L = _native.cli()
    g = os.path.join(GOLDEN, "create")
    d = short_tmpdir()
    try:
        shutil.copy(g + "/ref.igd", d + "/db.igd")
        shutil.copy(g + "/ref_index.tsv", d + "/db_index.tsv")
        script = os.path.join(d, "t.c")
        open(script, "w").write(r'''
#include <stdio.h>
#include <stdlib.h>
#include "igd_search.h"
int main(int argc, char **argv) {
    IGD = get_igdinfo(argv[1]);
    IGD->finfo = get_fileinfo(argv[2], &IGD->nFiles);
    fP = fopen(argv[1], "rb");
    double *sm = calloc(IGD->nFiles, sizeof(double));
    seqOverlaps(argv[3], sm);
    for (int i = 0; i < IGD->nFiles; i++) printf("%i\t%i\t%10.6f\t%s\n", i, IGD->finfo[i].nr, sm[i], IGD->finfo[i].fileName);
    return 0;
}
''')
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), "-o", exe, script, "-L" + _native.LIBDIR, "-ligd", "-ligd_hip",
                               "-Wl,-rpath," + _native.LIBDIR])
        out = subprocess.run([exe, d + "/db.igd", d + "/db_index.tsv", g + "/q.bed"], stdout=subprocess.PIPE, check=True, timeout=600).stdout.decode()
        assert out.splitlines() == open(g + "/search_s.txt").read().splitlines()[1:]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_python_database_seqpare_matches_golden_reference_numbers():
    """igd_amd.Database.seqpare (ctypes over igd_hip_seqpare) with the query grouping done in Python."""
    from igd_amd import Database
    g = os.path.join(GOLDEN, "create")
    d = short_tmpdir()
    try:
        shutil.copy(g + "/ref.igd", d + "/db.igd")
        shutil.copy(g + "/ref_index.tsv", d + "/db_index.tsv")
        db = Database(d + "/db.igd")
        # contigs of the database, from the oracle-independent header reader of the test helpers
        from test_oracle_create import split_igd
        _, cn, _ = split_igd(d + "/db.igd")
        cid = {n.decode(): i for i, n in enumerate(cn)}
        groups, order = {}, []
        nq_total = 0
        for line in open(g + "/q.bed"):
            f = line.rstrip("\n").split("\t")
            if len(f) < 3 or not f[0].startswith("chr") or int(f[2]) <= 0 or int(f[1]) > int(f[2]):
                continue
            nq_total += 1
            if f[0] not in groups:
                groups[f[0]] = []
                order.append(f[0])
            groups[f[0]].append((int(f[1]), int(f[2])))
        ichr, qs, qe, grp = [], [], [], []
        ng = 0
        for name in order:
            if name not in cid:
                continue
            for s, e in sorted(groups[name], key=lambda x: x[0]):        # stable, by start
                ichr.append(cid[name]); qs.append(s); qe.append(e); grp.append(ng)
            ng += 1
        nr = [int(l.split("\t")[2]) for l in open(d + "/db_index.tsv").read().splitlines()[1:]]
        sm = db.seqpare(ichr, qs, qe, grp, ng, n_queries_total=nq_total, nr=nr)
        want = [l.split("\t")[2].strip() for l in open(g + "/search_s.txt").read().splitlines()[1:]]
        assert ["%.6f" % x for x in sm] == want
        db.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)
