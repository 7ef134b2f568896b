"""`igd create`: CPU oracle (oracle/igd_oracle_create.c) vs the REAL reference
binary (oracle/_ref/igd) -- the files they write must be the same bytes.

What "same" means: `<db>_index.tsv` byte for byte; `<db>.igd` byte for byte EXCEPT the padding of the
40-byte contig-name fields, which the reference fills with whatever follows its strdup'd string on
the heap (src/igd_base.c:420); stdout text identical.  The tile bytes include the order of records
with equal start, which is a property of the reference's unstable radix sort
(src/igd_base.h:196-249) -- the inputs here are built to have many ties, buckets larger than 64 on
every radix level, tiles of <= 64 records, start >= end lines and intervals spanning many tiles.
"""
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

from helpers import ORACLE_BIN, REF_BIN, build_oracle, short_tmpdir

pytestmark = pytest.mark.ref


def split_igd(path):
    """-> (header ints, names, data bytes)."""
    b = open(path, "rb").read()
    nbp, gtype, nctg = np.frombuffer(b, "<i4", 3)
    ntile = np.frombuffer(b, "<i4", nctg, 12)
    tot = int(ntile.sum())
    ncnt = np.frombuffer(b, "<i4", tot, 12 + 4 * nctg)
    o = 12 + 4 * nctg + 4 * tot
    names = [b[o + 40 * i:o + 40 * i + 40].split(b"\0")[0] for i in range(nctg)]
    return (int(nbp), int(gtype), int(nctg), ntile.tolist(), ncnt.tolist()), names, b[o + 40 * nctg:]


def same_igd(a, b):
    ha, na, da = split_igd(a)
    hb, nb, db = split_igd(b)
    assert ha == hb
    assert na == nb
    assert len(da) == len(db)
    assert da == db


def write_beds(rng, d, nfiles, n, nbp, ncols=5, span_tiles=40, gz_some=False):
    os.makedirs(d)
    ctgs = ["chr1", "chr2", "chrX", "chr10", "chrUn_x"]
    for f in range(nfiles):
        lines = []
        for i in range(n):
            c = rng.choice(ctgs[:rng.choice([1, 3, 5])])
            m = rng.random()
            if m < 0.35:
                s = rng.randrange(0, nbp * span_tiles)
            elif m < 0.6:
                s = 5 * nbp + rng.randrange(0, 300)                 # one dense bucket: > 64 on the low levels
            elif m < 0.8:
                s = 7 * nbp + 256 * rng.randrange(0, 8) + rng.randrange(0, 3)   # heavy ties
            else:
                s = rng.choice([0, nbp - 1, nbp, 65536, 65535, 1 << 24, (1 << 24) - 1, 3 * nbp])
            L = rng.choice([0, 1, 7, nbp // 2, nbp, 3 * nbp + 5, rng.randrange(1, 2 * nbp)])
            if rng.random() < 0.01:
                L = -5
            cols = [c, str(s), str(s + L), "n%d" % i, str(rng.randrange(0, 1000)), "+"][:ncols]
            lines.append("\t".join(cols))
        path = os.path.join(d, "f%03d.bed" % f)
        data = ("\n".join(lines) + "\n").encode()
        if gz_some and f % 3 == 1:
            import gzip
            with gzip.open(path + ".gz", "wb") as fh:
                fh.write(data)
        else:
            open(path, "wb").write(data)


def run_both(d, in_arg, extra):
    outs = {}
    for who, exe in (("ref", REF_BIN), ("orc", ORACLE_BIN)):
        o = os.path.join(d, "o")                       # same output path for both: identical stdout
        shutil.rmtree(o, ignore_errors=True)
        p = subprocess.run([exe, "create", in_arg, o, "db"] + extra, stdout=subprocess.PIPE, timeout=600)
        assert p.returncode == 0
        keep = os.path.join(d, who)
        shutil.rmtree(keep, ignore_errors=True)
        shutil.copytree(o, keep)
        outs[who] = p.stdout
    return outs


@pytest.mark.parametrize("seed,b,nfiles,n,ncols", [
    (1, 12, 12, 400, 5), (2, 14, 10, 1500, 5), (3, 11, 23, 200, 6), (4, 14, 11, 60, 3), (5, 13, 30, 900, 4),
])
def test_default_mode_files_are_identical(seed, b, nfiles, n, ncols):
    build_oracle()
    rng = random.Random(seed)
    d = short_tmpdir()
    try:
        write_beds(rng, os.path.join(d, "in"), nfiles, n, 1 << b, ncols, gz_some=(seed == 2))
        outs = run_both(d, os.path.join(d, "in") + "/", ["-b", str(b)])
        assert outs["ref"] == outs["orc"]
        assert open(d + "/ref/db_index.tsv", "rb").read() == open(d + "/orc/db_index.tsv", "rb").read()
        same_igd(d + "/ref/db.igd", d + "/orc/db.igd")
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("seed,b,nfiles,n", [(11, 12, 3, 500), (12, 14, 12, 700), (13, 11, 1, 2000)])
def test_gtype0_mode_files_are_identical(seed, b, nfiles, n):
    build_oracle()
    rng = random.Random(seed)
    d = short_tmpdir()
    try:
        write_beds(rng, os.path.join(d, "in"), nfiles, n, 1 << b, 3)
        outs = run_both(d, os.path.join(d, "in"), ["-b", str(b), "-s", "0"])
        assert outs["ref"] == outs["orc"]
        assert open(d + "/ref/db_index.tsv", "rb").read() == open(d + "/orc/db_index.tsv", "rb").read()
        same_igd(d + "/ref/db.igd", d + "/orc/db.igd")
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_file_list_mode_files_are_identical():
    build_oracle()
    rng = random.Random(21)
    d = short_tmpdir()
    try:
        write_beds(rng, os.path.join(d, "in"), 14, 300, 1 << 13, 5)
        names = sorted(os.listdir(os.path.join(d, "in")))
        rng.shuffle(names)
        open(os.path.join(d, "in", names[0]), "w").write("track name=x\nchr1\t5\t9\n")   # first line invalid: file dropped
        lst = os.path.join(d, "list.txt")
        open(lst, "w").write("".join(os.path.join(d, "in", x) + "\n" for x in names) + d + "/missing.bed\n")
        outs = run_both(d, lst, ["-b", "13", "-f"])
        assert outs["ref"] == outs["orc"]
        assert open(d + "/ref/db_index.tsv", "rb").read() == open(d + "/orc/db_index.tsv", "rb").read()
        # the reference passes an uninitialised value in this mode: compare everything but that column
        ha, na, da = split_igd(d + "/ref/db.igd")
        hb, nb, db = split_igd(d + "/orc/db.igd")
        assert ha == hb and na == nb and len(da) == len(db)
        ra = np.frombuffer(da, "<i4").reshape(-1, 4)[:, :3]
        rb = np.frombuffer(db, "<i4").reshape(-1, 4)[:, :3]
        assert np.array_equal(ra, rb)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_bed4_mode_files_are_identical():
    build_oracle()
    rng = random.Random(31)
    d = short_tmpdir()
    try:
        nbp = 1 << 12
        lines = []
        for i in range(6000):
            s = rng.choice([rng.randrange(0, 50 * nbp), 9 * nbp + rng.randrange(0, 200)])
            L = rng.choice([0, 3, nbp, rng.randrange(1, 3 * nbp)])
            lines.append("chr%d\t%d\t%d\tTF%d\t%d" % (rng.randrange(1, 4), s, s + L, rng.randrange(0, 17), rng.randrange(0, 900)))
        src = os.path.join(d, "all.bed")
        open(src, "w").write("\n".join(lines) + "\n")
        outs = run_both(d, src, ["-b", "12", "-s", "2"])
        assert outs["ref"] == outs["orc"]
        assert open(d + "/ref/db_index.tsv", "rb").read() == open(d + "/orc/db_index.tsv", "rb").read()
        same_igd(d + "/ref/db.igd", d + "/orc/db.igd")
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_value_column_quirk_of_the_creeping_column_limit():
    """str_splits overwrites its column limit with the count of the line just split
    (src/igd_base.c:37-51), so the 5-column line right after a 3-column line loses its value."""
    build_oracle()
    d = short_tmpdir()
    try:
        os.makedirs(d + "/in")
        for f in range(10):
            rows = ["chr1\t%d\t%d\tx\t%d" % (100 * i + f, 100 * i + f + 50, 10 + i) for i in range(6)]
            if f == 4:
                rows.insert(2, "chr1\t777\t900")
                rows.insert(5, "chr1\t10\t20\tonly4")
            open(d + "/in/f%02d.bed" % f, "w").write("\n".join(rows) + "\n")
        outs = run_both(d, d + "/in/", [])
        assert outs["ref"] == outs["orc"]
        same_igd(d + "/ref/db.igd", d + "/orc/db.igd")
        _, _, data = split_igd(d + "/orc/db.igd")
        recs = np.frombuffer(data, "<i4").reshape(-1, 4)
        f4 = recs[recs[:, 0] == 4]
        assert (f4[:, 3] == 0).sum() >= 3            # the two short lines and the line after the 3-column one
    finally:
        shutil.rmtree(d, ignore_errors=True)


def write_odd_inputs(d, which):
    """inputs off the beaten path, shared with tests/test_gpu_create.py"""
    os.makedirs(d + "/in")
    rng = random.Random(1234)
    if which == "nothing_valid":                 # every line has start >= end: header-only database
        for f in range(10):
            open(d + "/in/f%02d.bed" % f, "w").write("".join("chr1\t%d\t%d\tx\t1\n" % (100 + i, 100 + i - f % 2) for i in range(20)))
    elif which == "many_contigs":                # more contigs than the LDS table of k_span holds
        for f in range(10):
            rows = []
            for i in range(700):
                c = "chrUn_%04d" % rng.randrange(0, 3000)
                s = rng.randrange(0, 200000)
                rows.append("%s\t%d\t%d\tx\t%d" % (c, s, s + rng.randrange(1, 40000), i))
            open(d + "/in/f%02d.bed" % f, "w").write("\n".join(rows) + "\n")
    elif which == "huge_spans":                  # intervals covering thousands of tiles, one covering a contig
        for f in range(10):
            rows = ["chr1\t%d\t%d\tx\t7" % (1000 * f, 1000 * f + 30000000 + f), "chr2\t0\t250000000\tw\t1"]
            for i in range(50):
                s = rng.randrange(0, 1 << 27)
                rows.append("chr1\t%d\t%d\ty\t%d" % (s, s + rng.choice([1, 5000, 3000000]), i))
            open(d + "/in/f%02d.bed" % f, "w").write("\n".join(rows) + "\n")
    elif which == "subdir_and_empty_file":       # a directory and an empty file among the inputs: 0 regions, "-nan" average
        for f in range(11):
            rows = ["chr1\t%d\t%d\tx\t%d" % (s, s + rng.randrange(1, 5000), i) for i, s in enumerate(rng.sample(range(100000), 50))]
            open(d + "/in/f%02d.bed" % f, "w").write("\n".join(rows) + "\n")
        os.makedirs(d + "/in/sub")
        open(d + "/in/sub/inner.bed", "w").write("chr1\t1\t2\n")
        open(d + "/in/zempty.bed", "w").write("")
    elif which == "no_trailing_newline_crlf":
        for f in range(10):
            body = "\r\n".join("chr%d\t%d\t%d\tn\t%d" % (1 + i % 2, 10 * i, 10 * i + 25, i) for i in range(40))
            open(d + "/in/f%02d.bed" % f, "wb").write(body.encode())      # CRLF, last line unterminated


@pytest.mark.parametrize("which,b", [("nothing_valid", 14), ("many_contigs", 13), ("huge_spans", 11), ("no_trailing_newline_crlf", 14),
                                     ("subdir_and_empty_file", 14)])
def test_odd_inputs_files_are_identical(which, b):
    build_oracle()
    d = short_tmpdir()
    try:
        write_odd_inputs(d, which)
        outs = run_both(d, d + "/in/", ["-b", str(b)])
        assert outs["ref"] == outs["orc"]
        assert open(d + "/ref/db_index.tsv", "rb").read() == open(d + "/orc/db_index.tsv", "rb").read()
        same_igd(d + "/ref/db.igd", d + "/orc/db.igd")
    finally:
        shutil.rmtree(d, ignore_errors=True)
