#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ with the REAL reference binary.

Run in the build container only (needs oracle/_ref/igd, built by `make -C oracle` from
/root/reference/src).  Everything written here is DATA: BED inputs, the .igd/_index.tsv the
reference's own `igd create` produced from them, query files, and the reference's stdout for a
list of command lines (manifest.json).  No reference source text is copied.

    python tests/golden/make_golden.py          # rewrites tests/golden/<case>/

The reference cannot take paths longer than ~60 characters (char fname[64]), so every case is
built in /tmp/ig_<case>/ and copied here; manifests store paths relative to the case dir and the
tests re-create the same short directory before running anything against them.
"""
import gzip
import json
import os
import random
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import REF_BIN, have_ref, ref_create, write_bed  # noqa: E402


def run_ref(args, cwd):
    p = subprocess.run([REF_BIN] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if p.returncode != 0:
        raise RuntimeError("reference failed: %s\n%s" % (args, p.stderr.decode()[-300:]))
    return p.stdout.decode()


class Case:
    def __init__(self, name):
        self.name = name
        self.tmp = "/tmp/ig_" + name
        shutil.rmtree(self.tmp, ignore_errors=True)
        os.makedirs(os.path.join(self.tmp, "beds"))
        self.runs = []
        self.meta = {}

    def beds(self, files, gz=False):
        """files: list of row lists; written as beds/fNN.bed[.gz] (>= 10 files for `igd create`)."""
        for i, rows in enumerate(files):
            write_bed(os.path.join(self.tmp, "beds", "f%02d.bed%s" % (i, ".gz" if gz else "")), rows, gz=gz)

    def create(self, b=14, s0=False):
        ref_create(os.path.join(self.tmp, "beds") + "/*", self.tmp, "db", b=b, s0=s0)
        self.meta.update(nbp_log=b, gtype=0 if s0 else 1)

    def queries(self, name, rows, gz=False, raw=None):
        path = os.path.join(self.tmp, name)
        if raw is not None:
            (gzip.open if gz else open)(path, "wb").write(raw)
        else:
            write_bed(path, rows, gz=gz)

    def run(self, args):
        """args relative to the case dir, e.g. ["search", "db.igd", "-q", "q.bed", "-v", "500"]"""
        out = run_ref(args, cwd=self.tmp)
        key = "out%02d.txt" % len(self.runs)
        open(os.path.join(self.tmp, key), "w").write(out)
        self.runs.append({"args": args, "stdout": key})

    def hitmap(self, v):
        """`igd search db.igd -m [-v v] -o <name>`: keeps the written matrix file and the stdout."""
        name = "hm%d.txt" % v
        args = ["search", "db.igd", "-m", "-o", name] + (["-v", str(v)] if v else [])
        out = run_ref(args, cwd=self.tmp)
        key = "hm%d.stdout" % v
        open(os.path.join(self.tmp, key), "w").write(out)
        self.meta.setdefault("hitmaps", []).append({"args": args, "file": name, "stdout": key})

    def finish(self, keep_beds=True):
        json.dump({"runs": self.runs, "meta": self.meta}, open(os.path.join(self.tmp, "manifest.json"), "w"), indent=1)
        dst = os.path.join(HERE, self.name)
        shutil.rmtree(dst, ignore_errors=True)
        shutil.copytree(self.tmp, dst, ignore=None if keep_beds else shutil.ignore_patterns("beds"))
        size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(dst) for f in fs)
        print("%-14s %3d runs  %7d bytes" % (self.name, len(self.runs), size))


def q_modes(c, qname, vs=(500,), full=True):
    c.run(["search", "db.igd", "-q", qname])
    for v in vs:
        c.run(["search", "db.igd", "-q", qname, "-v", str(v)])
    if full:
        c.run(["search", "db.igd", "-q", qname, "-f"])


def main():
    if not have_ref():
        raise SystemExit("oracle/_ref/igd missing: run `make -C oracle` where /root/reference exists")

    # 1. edge: half-open ends, zero-length / inverted queries, past-the-end tiles
    c = Case("edge")
    c.beds([[("chr1", 1000, 2000, "a", 500), ("chr1", 16000, 50000, "b", 100 * i), ("chr1", 70000, 70010, "c", 1000)]
            for i in range(10)])
    c.create(14)
    qs = [(2000, 3000), (1999, 2000), (500, 1000), (500, 1001), (1500, 1500), (1600, 1500), (0, 100000),
          (20000, 40000), (33000, 34000), (60000, 80000), (81920, 90000), (900000, 900100), (-5, 1500)]
    rows = [("chr1", s, e) for s, e in qs] + [("chr2", 0, 100), ("1", 1000, 2000)]
    c.queries("q.bed", rows)
    q_modes(c, "q.bed", vs=(500, 1, 1000))
    for s, e in [(20000, 40000), (60000, 80000), (81920, 90000), (33000, 34000)]:
        c.run(["search", "db.igd", "-r", "chr1", str(s), str(e)])
        c.run(["search", "db.igd", "-r", "chr1", str(s), str(e), "-v", "500"])
        c.run(["search", "db.igd", "-r", "chr1", str(s), str(e), "-f"])
    c.hitmap(0); c.hitmap(500)
    c.finish()

    # 2. quirk: empty first tile -> default finds nothing, -v N does
    c = Case("quirk")
    rng = random.Random(11)
    files = [[("chr1", 20000, 20100, "n", 900)] for _ in range(10)]
    for f in files:                       # sparse second/third contig: many empty first tiles
        for _ in range(6):
            ctg = rng.choice(["chr2", "chr3"])
            s = rng.randrange(0, 60) * 16384 + rng.randrange(0, 16384)
            f.append((ctg, s, s + rng.choice([50, 3000, 20000, 40000]), "n", rng.randint(0, 1000)))
        f.sort(key=lambda r: (r[0], r[1]))
    c.beds(files)
    c.create(14)
    rows = [("chr1", 100, 30000), ("chr1", 100, 16384), ("chr1", 16384, 30000)]
    for _ in range(120):
        ctg = rng.choice(["chr2", "chr3"])
        s = rng.randrange(0, 62 * 16384)
        rows.append((ctg, s, s + rng.choice([100, 16384, 40000, 100000])))
    c.queries("q.bed", rows)
    q_modes(c, "q.bed", vs=(1, 500))
    c.run(["search", "db.igd", "-r", "chr1", "100", "30000"])
    c.run(["search", "db.igd", "-r", "chr1", "100", "30000", "-v", "1"])
    c.hitmap(0); c.hitmap(300)
    c.finish()

    # 3. branch: tiles with exactly 1,2,15,16,17 records (the <16 branch of get_overlaps_v),
    #    duplicate starts, records spanning 3+ tiles, queries spanning 5+ tiles, n2 clamped
    c = Case("branch")
    nbp = 2048
    files = [[] for _ in range(10)]
    want = {3: 1, 5: 2, 7: 15, 9: 16, 11: 17}
    k = 0
    for tile, cnt in want.items():
        for i in range(cnt):
            s = tile * nbp + (100 if i % 3 == 0 else 100 + 7 * i)     # duplicate starts
            files[k % 10].append(("chr1", s, s + 50 + i, "n", 60 * i))
            k += 1
    files[0].append(("chr1", 14 * nbp + 5, 18 * nbp + 9, "span", 700))  # record over 5 tiles
    files[1].append(("chr1", 20 * nbp, 20 * nbp + 1, "last", 10))       # defines mTile
    for f in files:
        if not f:
            f.append(("chr1", 20 * nbp + 3, 20 * nbp + 9, "pad", 1))
        f.sort(key=lambda r: r[1])
    c.beds(files)
    c.create(11)
    rows = []
    for tile in (3, 5, 7, 9, 11, 14, 15, 16, 17, 18, 20):
        for (ds, L) in ((0, 1), (50, 200), (99, 2), (100, 1), (0, nbp), (0, 6 * nbp), (nbp - 1, 2)):
            rows.append(("chr1", tile * nbp + ds, tile * nbp + ds + L))
    rows += [("chr1", 0, 40 * nbp), ("chr1", 19 * nbp, 500 * nbp), ("chr1", 21 * nbp, 22 * nbp)]
    c.queries("q.bed", rows)
    q_modes(c, "q.bed", vs=(1, 60, 500, 900))
    c.hitmap(0); c.hitmap(60)
    c.finish()

    # 4. parse: what the query reader accepts / skips (parse_bed + ks_getuntil)
    c = Case("parse")
    c.beds([[("chr1", 1000 * i, 1000 * i + 5000, "n", 100 * i), ("chrUn_gl000220_abcdefghijklmnopqrstuvwxy", 10, 900, "n", 5)]
            for i in range(10)])
    c.create(14)
    raw = (b"track name=x\n# comment\nchr1\t100\t6000\n"
           b"chr1\t2000\n"                      # < 3 columns
           b"chr1\t3000\t0\n"                   # end <= 0 -> skipped
           b"chr1\t3000\t-5\n"
           b"1\t100\t6000\n"                    # no chr prefix
           b"Chr1\t100\t6000\n"
           b"chr1\t 4000\t 9000\textra\tcols\t1\n"
           b"chr1\t5000abc\t7000xyz\n"          # atol stops at junk
           b"chr1\t+100\t+2500\n"
           b"chr1\t1e3\t9000\n"
           b"chrUn_gl000220_abcdefghijklmnopqrstuvwxy\t0\t1000\n"      # 39 chars: accepted
           b"chrUn_gl000220_abcdefghijklmnopqrstuvwxyz\t0\t1000\n"     # 40 chars: rejected
           b"chr1 100 6000\n"                   # spaces, not tabs
           b"chr1\t7000\t12000\r\n"             # CRLF
           b"\n"
           b"chr1\t0\t100000")                  # no trailing newline
    c.queries("q.bed", None, raw=raw)
    c.queries("q.bed.gz", None, gz=True, raw=raw)
    q_modes(c, "q.bed", vs=(300,))
    q_modes(c, "q.bed.gz", vs=(300,))
    c.finish()

    # 5. gType 0 (create -s 0): 12-byte records, -v ignored
    c = Case("gtype0")
    rng = random.Random(5)
    files = []
    for i in range(10):
        rows = []
        for _ in range(40):
            s = rng.randrange(0, 30 * 4096)
            rows.append(("chr%d" % rng.randint(1, 2), s, s + rng.choice([1, 100, 5000, 9000])))
        rows.sort(key=lambda r: (r[0], r[1]))
        files.append(rows)
    c.beds(files)
    c.create(12, s0=True)
    rows = []
    for _ in range(200):
        s = rng.randrange(0, 33 * 4096)
        rows.append((rng.choice(["chr1", "chr2", "chr5"]), s, s + rng.choice([0, 1, 50, 4096, 20000, -3])))
    c.queries("q.bed", rows)
    q_modes(c, "q.bed", vs=(500,))
    c.run(["search", "db.igd", "-r", "chr2", "5000", "30000"])
    c.finish()

    # 6. small-random: 12 files x 600 intervals, built by the reference's `igd create`; the tests
    #    also rebuild it from the same BEDs with the product's writer and expect equal counts
    c = Case("smallrand")
    rng = random.Random(61)
    ctgs = ["chr1", "chr2", "chr3", "chrX"]
    files = []
    for i in range(12):
        rows = []
        for _ in range(600):
            s = rng.randrange(0, 4_000_000)
            rows.append((rng.choice(ctgs), s, s + rng.randint(50, 30000), "p", rng.randint(0, 1000)))
        rows.sort(key=lambda r: (r[0], r[1]))
        files.append(rows)
    c.beds(files, gz=True)
    c.create(14)
    rows = []
    for _ in range(2000):
        s = rng.randrange(0, 4_050_000)
        rows.append((rng.choice(ctgs + ["chr9"]), s, s + rng.randint(1, 60000)))
    rows.sort(key=lambda r: (r[0], r[1]))
    c.queries("q.bed", rows)
    c.queries("q100.bed", rows[:100])
    q_modes(c, "q.bed", vs=(500,), full=False)
    c.run(["search", "db.igd", "-q", "q100.bed", "-f"])
    c.hitmap(0); c.hitmap(500)
    c.finish()

    # 7. BASELINE config 1: 8 files x 10k intervals (-b 14) made by the product's deterministic
    #    generator (the reference's create dies with < 10 files), searched by the REFERENCE.
    #    Only the expected stdout is kept; the test regenerates the byte-identical inputs.
    c = Case("config1")
    synth = os.path.join(ROOT, "bin", "igd_synth")
    subprocess.check_call([synth, "db", os.path.join(c.tmp, "db.igd"), "--small", "--files", "8", "--per-file", "10000"])
    subprocess.check_call([synth, "queries", os.path.join(c.tmp, "q.bed"), "--small", "--n", "10000"])
    import hashlib
    c.meta["md5"] = {f: hashlib.md5(open(os.path.join(c.tmp, f), "rb").read()).hexdigest()
                     for f in ("db.igd", "db_index.tsv", "q.bed")}
    q_modes(c, "q.bed", vs=(500,), full=False)
    for f in ("db.igd", "q.bed"):
        os.remove(os.path.join(c.tmp, f))
    c.finish(keep_beds=False)


if __name__ == "__main__":
    main()
