"""Regenerates tests/golden/create/ with the REAL reference (oracle/_ref/igd, built by oracle/Makefile from
/root/reference/src).  Run in the build container:  python tests/golden/make_create_golden.py

  in/f00.bed .. f11.bed   inputs (many equal starts, one dense region -> tiles > 64 records, start>=end lines)
  ref.igd, ref_index.tsv  what `igd create in/ OUT/ db -b 12` of the reference wrote
  stdout.txt              its stdout (paths replaced by IN/ and OUT/)
  q.bed, search_f.txt     queries and the reference's `search -f` output on ref.igd (prints records in
                          tile order, i.e. it exposes the sort's order of equal starts)
  search_s.txt            the reference's `search -q q.bed -s` (Seqpare) output on ref.igd
"""
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import REF_BIN  # noqa: E402

out = os.path.join(HERE, "create")
shutil.rmtree(out, ignore_errors=True)
os.makedirs(out + "/in")
rng = random.Random(20261003)
nbp = 1 << 12
for f in range(12):
    rows = []
    for i in range(160):
        m = rng.random()
        if m < 0.4:
            s = rng.randrange(0, 30 * nbp)
        elif m < 0.7:
            s = 4 * nbp + rng.randrange(0, 120)
        else:
            s = 9 * nbp + 256 * rng.randrange(0, 4) + rng.randrange(0, 2)
        L = rng.choice([0, 1, 9, nbp // 2, nbp, 2 * nbp + 5, rng.randrange(1, 2 * nbp)])
        rows.append("%s\t%d\t%d\tp%d\t%d" % (rng.choice(["chr1", "chr2", "chrX"]), s, s + L, i, rng.randrange(0, 1000)))
    open(out + "/in/f%02d.bed" % f, "w").write("\n".join(rows) + "\n")
tmp = tempfile.mkdtemp(prefix="igc", dir="/tmp")
p = subprocess.run([REF_BIN, "create", out + "/in/", tmp + "/o", "db", "-b", "12"], stdout=subprocess.PIPE, check=True)
open(out + "/stdout.txt", "w").write(p.stdout.decode().replace(tmp + "/o/", "OUT/").replace(out + "/in/", "IN/"))
shutil.copy(tmp + "/o/db.igd", out + "/ref.igd")
shutil.copy(tmp + "/o/db_index.tsv", out + "/ref_index.tsv")
q = []
for i in range(60):
    s = rng.choice([rng.randrange(0, 30 * nbp), 4 * nbp + rng.randrange(0, 120), 9 * nbp + rng.randrange(0, 1024)])
    q.append("%s\t%d\t%d" % (rng.choice(["chr1", "chr2", "chrX"]), s, s + rng.choice([1, 50, nbp, 3 * nbp])))
open(out + "/q.bed", "w").write("\n".join(q) + "\n")
p = subprocess.run([REF_BIN, "search", tmp + "/o/db.igd", "-q", out + "/q.bed", "-f"], stdout=subprocess.PIPE, check=True)
open(out + "/search_f.txt", "w").write(p.stdout.decode().replace(tmp + "/o/db.igd", "DB"))
p = subprocess.run([REF_BIN, "search", tmp + "/o/db.igd", "-q", out + "/q.bed", "-s"], stdout=subprocess.PIPE, check=True)
open(out + "/search_s.txt", "w").write(p.stdout.decode())
shutil.rmtree(tmp, ignore_errors=True)
print("wrote", out, {f: os.path.getsize(os.path.join(out, f)) for f in os.listdir(out) if os.path.isfile(os.path.join(out, f))})
