"""`igd create` on the GPU against the CPU oracle: the files must be the same
bytes -- header, tile counts, and every tile's records IN THE REFERENCE'S ORDER (its unstable radix
sort decides how records with equal start are ordered; oracle/igd_oracle_create.c restates it and is
pinned to the real reference by tests/test_oracle_create.py and tests/golden/create/).

Everything goes through the shipped entry points: `bin/igd create` (CLI flavour), `create_iGD` of
libigd_py.so / libigdr.so, and the engine's C ABI igd_hip_create.
"""
import ctypes as C
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ORACLE_BIN, ROOT, build_oracle, orc, short_tmpdir
from test_oracle_create import same_igd, split_igd, write_beds

pytestmark = pytest.mark.gpu

IGD_BIN = os.path.join(ROOT, "bin", "igd")


def run_pair(d, in_arg, extra):
    """bin/igd create vs igd_oracle create into the same output path -> (stdout_gpu, stdout_orc)."""
    build_oracle()
    outs = {}
    for who, exe in (("gpu", IGD_BIN), ("orc", ORACLE_BIN)):
        o = os.path.join(d, "o")
        shutil.rmtree(o, ignore_errors=True)
        p = subprocess.run([exe, "create", in_arg, o, "db"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert p.returncode == 0, p.stderr.decode()[-500:]
        keep = os.path.join(d, who)
        shutil.rmtree(keep, ignore_errors=True)
        shutil.copytree(o, keep)
        outs[who] = p.stdout
    return outs


def assert_same_files(d):
    assert open(d + "/gpu/db_index.tsv", "rb").read() == open(d + "/orc/db_index.tsv", "rb").read()
    assert open(d + "/gpu/db.igd", "rb").read() == open(d + "/orc/db.igd", "rb").read()


@pytest.mark.parametrize("seed,b,nfiles,n,ncols", [
    (1, 12, 12, 400, 5), (2, 14, 10, 1500, 5), (3, 11, 23, 200, 6), (4, 14, 3, 60, 3), (5, 13, 30, 900, 4),
    (6, 14, 1, 5000, 5), (7, 19, 4, 3000, 5),
])
def test_cli_create_default_mode(seed, b, nfiles, n, ncols):
    rng = random.Random(seed)
    d = short_tmpdir()
    try:
        write_beds(rng, os.path.join(d, "in"), nfiles, n, 1 << b, ncols, gz_some=(seed == 2))
        outs = run_pair(d, os.path.join(d, "in") + "/", ["-b", str(b)])
        assert outs["gpu"] == outs["orc"]
        assert_same_files(d)
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("seed,b,nfiles,n", [(11, 12, 3, 500), (12, 14, 12, 700), (13, 11, 1, 2000)])
def test_cli_create_gtype0(seed, b, nfiles, n):
    rng = random.Random(seed)
    d = short_tmpdir()
    try:
        write_beds(rng, os.path.join(d, "in"), nfiles, n, 1 << b, 3)
        outs = run_pair(d, os.path.join(d, "in"), ["-b", str(b), "-s", "0"])
        assert outs["gpu"] == outs["orc"]
        assert_same_files(d)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_cli_create_file_list_and_bed4():
    rng = random.Random(21)
    d = short_tmpdir()
    try:
        write_beds(rng, os.path.join(d, "in"), 14, 300, 1 << 13, 5)
        names = sorted(os.listdir(os.path.join(d, "in")))
        rng.shuffle(names)
        open(os.path.join(d, "in", names[0]), "w").write("track name=x\nchr1\t5\t9\n")
        lst = os.path.join(d, "list.txt")
        open(lst, "w").write("".join(os.path.join(d, "in", x) + "\n" for x in names) + d + "/missing.bed\n")
        outs = run_pair(d, lst, ["-b", "13", "-f"])
        assert outs["gpu"] == outs["orc"]
        assert_same_files(d)
        nbp = 1 << 12
        lines = []
        for i in range(6000):
            s = rng.choice([rng.randrange(0, 50 * nbp), 9 * nbp + rng.randrange(0, 200)])
            L = rng.choice([0, 3, nbp, rng.randrange(1, 3 * nbp)])
            lines.append("chr%d\t%d\t%d\tTF%d\t%d" % (rng.randrange(1, 4), s, s + L, rng.randrange(0, 17), rng.randrange(0, 900)))
        src = os.path.join(d, "all.bed")
        open(src, "w").write("\n".join(lines) + "\n")
        outs = run_pair(d, src, ["-b", "12", "-s", "2"])
        assert outs["gpu"] == outs["orc"]
        assert_same_files(d)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_creeping_column_limit_and_long_lines():
    """Mixed column counts force the sequential re-parse; a line longer than the 1024-byte gzgets
    buffer is cut into several "lines" exactly like the reference cuts it."""
    d = short_tmpdir()
    try:
        os.makedirs(d + "/in")
        for f in range(10):
            rows = ["chr1\t%d\t%d\tx\t%d" % (100 * i + f, 100 * i + f + 50, 10 + i) for i in range(6)]
            if f == 4:
                rows.insert(2, "chr1\t777\t900")
                rows.insert(5, "chr1\t10\t20\tonly4")
            if f == 7:
                rows.insert(1, "chr2\t5\t99\t" + "y" * 1100 + "\t7")
                rows.insert(3, "chr2\t" + " " * 1015 + "12\t40\tz\t3")     # cut inside the numbers
            open(d + "/in/f%02d.bed" % f, "w").write("\n".join(rows) + "\n")
        outs = run_pair(d, d + "/in/", [])
        assert outs["gpu"] == outs["orc"]
        assert_same_files(d)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_big_tiles_sort_in_hbm_scratch():
    """Tiles beyond the LDS capacity (1024 records) take the HBM-scratch path of k_tile_sort; deep
    recursion (buckets > 64 on all four radix levels) and a tile made of one repeated start."""
    rng = random.Random(77)
    d = short_tmpdir()
    try:
        os.makedirs(d + "/in")
        nbp = 1 << 14
        for f in range(6):
            rows = []
            for i in range(9000):
                m = rng.random()
                if m < 0.3:
                    s = 3 * nbp + rng.randrange(0, 900)                      # ~16k records in one tile
                elif m < 0.5:
                    s = (1 << 24) + 5 * nbp + 17                              # one start, thousands of times
                elif m < 0.7:
                    s = rng.choice([255, 256, 65535, 65536, (1 << 24) - 1, 1 << 24]) + rng.randrange(0, 2)
                else:
                    s = rng.randrange(0, 1 << 25)
                L = rng.choice([1, 10, 1000, nbp + 3])
                rows.append("chr1\t%d\t%d\tn\t%d" % (s, s + L, rng.randrange(0, 1000)))
            open(d + "/in/f%d.bed" % f, "w").write("\n".join(rows) + "\n")
        outs = run_pair(d, d + "/in/", [])
        assert outs["gpu"] == outs["orc"]
        assert_same_files(d)
        hdr, _, _ = split_igd(d + "/gpu/db.igd")
        assert max(hdr[4]) > 4096
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_created_database_is_searchable_and_matches_golden_reference_files():
    """tests/golden/create/: BED inputs + the .igd/_index.tsv the REAL reference wrote from them."""
    g = os.path.join(GOLDEN, "create")
    d = short_tmpdir()
    try:
        p = subprocess.run([IGD_BIN, "create", g + "/in/", d + "/o", "db", "-b", "12"], stdout=subprocess.PIPE, timeout=600)
        assert p.returncode == 0
        assert p.stdout.decode().replace(d + "/o/", "OUT/").replace(g + "/in/", "IN/") == open(g + "/stdout.txt").read()
        assert open(d + "/o/db_index.tsv", "rb").read() == open(g + "/ref_index.tsv", "rb").read()
        same_igd(d + "/o/db.igd", g + "/ref.igd")
        q = subprocess.run([IGD_BIN, "search", d + "/o/db.igd", "-q", g + "/q.bed", "-f"], stdout=subprocess.PIPE, timeout=600)
        assert q.stdout.decode().replace(d + "/o/db.igd", "DB") == open(g + "/search_f.txt").read()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_engine_c_abi_direct_and_python_flavour():
    from igd_amd import _native
    L = _native.hip()
    rng = np.random.default_rng(3)
    n, nbp = 20000, 4096
    ctg = rng.integers(0, 3, n).astype(np.int32)
    start = rng.integers(0, 400000, n).astype(np.int32)
    start[::3] = 77 * nbp + rng.integers(0, 64, len(start[::3]))
    end = (start + rng.integers(1, 3 * nbp, n)).astype(np.int32)
    value = rng.integers(0, 1000, n).astype(np.int32)
    file = np.sort(rng.integers(0, 40, n)).astype(np.int32)
    created = _native.create_arrays(nbp, 1, 3, ctg, start, end, value, file)
    ntile = created["nTile"]
    assert ntile.tolist() == [int(((end[ctg == c] - 1) // nbp).max()) + 1 for c in range(3)]
    # expected: oracle's tile sort applied to input-order tiles
    O = orc()
    recs = created["records"]
    tb = np.concatenate([[0], np.cumsum(ntile)])
    pos = 0
    for c in range(3):
        idx = np.nonzero(ctg == c)[0]
        for j in range(int(ntile[c])):
            sel = idx[(start[idx] // nbp <= j) & ((end[idx] - 1) // nbp >= j)]
            assert created["nCnt"][tb[c] + j] == len(sel)
            if len(sel) == 0:
                continue
            key = start[sel].astype(np.int32).copy()
            src = sel.astype(np.int32).copy()
            O.orc_tile_sort(key.ctypes.data_as(C.POINTER(C.c_int32)), src.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(len(sel)))
            exp = np.stack([file[src], start[src], end[src], value[src]], axis=1)
            assert np.array_equal(recs[pos:pos + len(sel)], exp), (c, j)
            pos += len(sel)
    assert pos == len(recs)

    # Python flavour: create_iGD writes the files AND leaves the database open in the handle
    from igd_amd import igd_py as iGD
    d = short_tmpdir()
    try:
        write_beds(random.Random(5), os.path.join(d, "in"), 4, 300, 1 << 14, 5)
        h = iGD.igd_py()
        h.create(os.path.join(d, "in"), os.path.join(d, "out"), "pydb", 16384)
        assert h.get_nFiles() == 4
        build_oracle()
        subprocess.run([ORACLE_BIN, "create", os.path.join(d, "in") + "/", os.path.join(d, "o2"), "pydb"], stdout=subprocess.PIPE, check=True)
        same_igd(os.path.join(d, "out", "pydb.igd"), os.path.join(d, "o2", "pydb.igd"))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_golden_smallrand_beds_give_the_reference_database():
    """tests/golden/smallrand holds gzip'd BED inputs AND the db.igd the reference created from them
    (-b 14): `bin/igd create` must reproduce that file (tile bytes included), and searching our file
    must print what the reference printed when searching its own."""
    from test_golden_oracle import materialize
    d, dst, man = materialize("smallrand")
    try:
        p = subprocess.run([IGD_BIN, "create", os.path.join(dst, "beds") + "/", d + "/w", "db"], stdout=subprocess.PIPE, timeout=600)
        assert p.returncode == 0
        same_igd(d + "/w/db.igd", os.path.join(dst, "db.igd"))
        assert open(d + "/w/db_index.tsv", "rb").read() == open(os.path.join(dst, "db_index.tsv"), "rb").read()
        for run in man["runs"]:
            if run["args"][0] != "search" or "-m" in run["args"]:
                continue
            args = [d + "/w/db.igd" if x == "db.igd" else os.path.join(dst, x) if x.endswith(".bed") else x for x in run["args"]]
            got = subprocess.run([IGD_BIN] + args, stdout=subprocess.PIPE, timeout=600).stdout.decode()
            assert got.replace(d + "/w/db.igd", "db.igd") == open(os.path.join(dst, run["stdout"])).read().replace(os.path.join(dst, "db.igd"), "db.igd")
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_fewer_than_ten_files_and_gtype0():
    """the reference's default create divides by n_files/10 (SIGFPE below 10 files): ours must not."""
    from helpers import Oracle, write_bed
    d = short_tmpdir("igw")
    try:
        beds = os.path.join(d, "b")
        os.makedirs(beds)
        write_bed(os.path.join(beds, "a.bed"), [("chr1", 5, 50, "n", 3), ("chr2", 70000, 70001, "n", 9), ("chr1", 9, 9, "n", 1)])
        write_bed(os.path.join(beds, "b.bed"), [("chr2", 16384, 40000, "n", 7)])
        for gt in (1, 0):
            out = os.path.join(d, "o%d" % gt)
            p = subprocess.run([IGD_BIN, "create", beds, out, "x", "-s", str(gt)], stdout=subprocess.PIPE, timeout=600)
            assert p.returncode == 0
            o = Oracle(os.path.join(out, "x.igd"))
            assert (o.nfiles, o.gtype, o.ctg_names()) == (2, gt, ["chr1", "chr2"])
            h, _ = o.search(np.array([0, 1, 1], np.int32), np.array([0, 16000, 69999], np.int32), np.array([100, 17000, 70001], np.int32))
            np.testing.assert_array_equal(h, [2, 0])       # query 2 starts in chr2's EMPTY tile 0: rule NEST drops it
            o.close()
        # a second create into the same place is refused like the reference refuses it
        p = subprocess.run([IGD_BIN, "create", beds, os.path.join(d, "o1"), "x"], stdout=subprocess.PIPE, timeout=600)
        assert b"exists!" in p.stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("which,b", [("nothing_valid", 14), ("many_contigs", 13), ("huge_spans", 11), ("no_trailing_newline_crlf", 14),
                                     ("subdir_and_empty_file", 14)])
def test_odd_inputs(which, b):
    """header-only database, 3000 contigs (global atomicMax path of k_span), intervals covering > 10^5 tiles,
    CRLF lines without a final newline -- same bytes as the oracle (pinned to the reference on the same inputs
    by tests/test_oracle_create.py::test_odd_inputs_files_are_identical)."""
    from test_oracle_create import write_odd_inputs
    d = short_tmpdir()
    try:
        write_odd_inputs(d, which)
        outs = run_pair(d, d + "/in/", ["-b", str(b)])
        assert outs["gpu"] == outs["orc"]
        assert_same_files(d)
    finally:
        shutil.rmtree(d, ignore_errors=True)
