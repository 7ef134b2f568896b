#!/usr/bin/env python3
"""Differential fuzz on the GPU box: bin/igd (GPU) vs oracle/_build/igd_oracle (CPU restatement, pinned to
the reference) on random inputs -- `create` in all modes (files compared byte for byte), then on the
created database `search -q`, `-q -v N`, `-q -f`, `-q -s`, `-r` and `-m` (stdout / map file compared).
usage: python tools/fuzz_gpu.py [cases] [seed0]     -> prints one line per case, exits 1 on a mismatch"""
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_oracle_create import write_beds          # noqa: E402
from test_oracle_seqpare import write_queries       # noqa: E402

IGD, ORC = os.path.join(ROOT, "bin", "igd"), os.path.join(ROOT, "oracle", "_build", "igd_oracle")
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def run(exe, args, cwd=None):
    p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=cwd, timeout=1200)
    if p.returncode != 0:
        raise SystemExit("FAILED rc=%d: %s %s\n%s" % (p.returncode, exe, args, p.stderr.decode()[-800:]))
    return p.stdout


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    bad = 0
    for c in range(cases):
        rng = random.Random(seed0 + c)
        d = tempfile.mkdtemp(prefix="igz", dir="/tmp")
        try:
            b = rng.choice([11, 12, 13, 14, 16])
            nfiles = rng.choice([1, 3, 10, 25, 60])
            n = rng.choice([5, 80, 600, 4000])
            s0 = rng.random() < 0.2
            ncols = 3 if s0 else rng.choice([3, 4, 5, 6])
            write_beds(rng, d + "/in", nfiles, n, 1 << b, ncols, gz_some=rng.random() < 0.3)
            extra = ["-b", str(b)] + (["-s", "0"] if s0 else [])
            outs = {}
            for who, exe in (("gpu", IGD), ("orc", ORC)):
                shutil.rmtree(d + "/o", ignore_errors=True)
                outs[who] = run(exe, ["create", d + "/in/", d + "/o", "db"] + extra)
                shutil.copytree(d + "/o", d + "/" + who)
            ok = outs["gpu"] == outs["orc"]
            for f in ("db.igd", "db_index.tsv"):
                ok = ok and open(d + "/gpu/" + f, "rb").read() == open(d + "/orc/" + f, "rb").read()
            db = d + "/gpu/db.igd"
            write_queries(rng, d + "/q.bed", rng.choice([1, 50, 2000, 30000]), 1 << b, dup=rng.choice([0.0, 0.3]))
            cmds = [["search", db, "-q", d + "/q.bed"], ["search", db, "-q", d + "/q.bed", "-f"],
                    ["search", db, "-r", "chr1", str(rng.randrange(0, 5 << b)), str(rng.randrange(0, 9 << b))]]
            if not s0:
                cmds += [["search", db, "-q", d + "/q.bed", "-v", str(rng.randrange(1, 900))], ["search", db, "-q", d + "/q.bed", "-s"]]
            what = []
            for cmd in cmds:
                same = run(IGD, cmd) == run(ORC, cmd)
                ok = ok and same
                label = "-r" if "-r" in cmd else "-q" + "".join(x for x in cmd[4:] if x.startswith("-"))
                what.append(label + ("" if same else "!"))
            if not s0 and nfiles <= 25:
                run(IGD, ["search", db, "-m", "-o", d + "/m_gpu.txt"]); run(ORC, ["search", db, "-m", "-o", d + "/m_orc.txt"])
                same = open(d + "/m_gpu.txt").read() == open(d + "/m_orc.txt").read()
                ok = ok and same
                what.append("-m" + ("" if same else "!"))
            print("case %4d  b=%2d files=%3d n=%5d %s  %s  %s" % (seed0 + c, b, nfiles, n, "gType0" if s0 else "gType1",
                                                                  " ".join(what), "ok" if ok else "MISMATCH"), flush=True)
            bad += 0 if ok else 1
            if not ok:
                keep = "/tmp/igz_bad_%d" % (seed0 + c)
                shutil.rmtree(keep, ignore_errors=True)
                shutil.copytree(d, keep)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    print("fuzz: %d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
