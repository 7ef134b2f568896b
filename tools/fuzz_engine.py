#!/usr/bin/env python3
"""tools/fuzz_engine.py [cases] [seed0] -- GPU box: the ENGINE (igd_hip_search through igd_amd.Database) against the oracle on
random databases (tile size 2^10 .. 2^19 -- all but 2^14 / 2^15 searched over the re-tiled copy -- one that is no power of two is left to tests/, 1 .. 300 files and 21 000 (windows of files), gType 0/1,
clustered or not) and random batches that mix what the kernels treat differently: short queries, queries of many
tiles, inverted ones, unknown contigs, starts beyond the contig, duplicates, a hot tile with 10^3 .. 10^5 queries --
position-sorted (order promise and device decides), unordered (device decides, bucket path), both builds of the merge
join, with and without -v.  One line per case; exits 1 on a mismatch."""
import os, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import Oracle                      # noqa: E402  (test infrastructure)
from igd_amd import Database, synth             # noqa: E402


def batch(rng, n, genome_len, nbp):
    kind = rng.integers(0, 5, n)
    c = rng.integers(0, 4, n).astype(np.int32)
    qs = rng.integers(0, genome_len, n).astype(np.int64)
    ln = np.where(kind == 0, rng.integers(1, 3000, n),
         np.where(kind == 1, rng.integers(1, 6 * nbp, n),
         np.where(kind == 2, rng.integers(4 * nbp, 400 * nbp, n), rng.integers(1, 800, n))))
    qe = qs + ln
    inv = rng.random(n) < 0.01
    qe = np.where(inv, qs - rng.integers(0, 50, n), qe)
    c = np.where(rng.random(n) < 0.01, -1, c)                    # unknown contig
    c = np.where(rng.random(n) < 0.005, 99, c)
    qs = np.where(rng.random(n) < 0.01, qs + 3 * genome_len, qs)  # beyond the contig
    neg = rng.random(n) < 0.01                                    # before the contig: (-nbp, 0) is tile 0 by C division, below is nothing
    qs = np.where(neg, -rng.integers(1, 2 * nbp, n), qs)
    qe = np.where(neg, qs + rng.integers(1, 5 * nbp, n), qe)
    if rng.random() < 0.6:                                       # a hot tile
        m = int(rng.choice([1000, 20000, 100000])); m = min(m, n)
        t0 = int(rng.integers(0, max(1, genome_len // nbp - 2))) * nbp
        qs[:m] = t0 + rng.integers(0, nbp, m); qe[:m] = qs[:m] + rng.integers(1, 3 * nbp, m); c[:m] = int(rng.integers(0, 4))
    if rng.random() < 0.3:
        k = n // 10
        qs[-k:] = qs[:k]; qe[-k:] = qe[:k]; c[-k:] = c[:k]       # duplicates
    qe = np.clip(qe, -2**31, 2**31 - 1); qs = np.clip(qs, -2**31, 2**31 - 1)
    return c.astype(np.int32), qs.astype(np.int32), qe.astype(np.int32)


# IGD_FUZZ_B="14,15": only these tile widths (the file's own tiles: k_split_local<FAST>, the DIRECT step, no re-tiled copy);
# IGD_FUZZ_BUILDS=",D": only these builds ("" = as shipped)
TILE_LOGS = [int(x) for x in os.environ.get("IGD_FUZZ_B", "10,11,12,13,14,15,16,17,18,19").split(",")]
BUILDS = os.environ["IGD_FUZZ_BUILDS"].split(",") if "IGD_FUZZ_BUILDS" in os.environ else ["", "0", "1", "D"]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad = 0
    for ci in range(cases):
        rng = np.random.default_rng(seed0 + ci)
        d = tempfile.mkdtemp(prefix="ige", dir="/tmp")
        try:
            b = int(rng.choice(TILE_LOGS)); files = int(rng.choice([1, 7, 60, 300, 300, 21000]))
            per = int(rng.choice([200, 3000, 20000])); gtype = int(rng.choice([0, 1, 1])); cl = bool(rng.random() < 0.3)
            if files > 10000: per = int(rng.choice([15, 150]))        # (more files than LDS counters: windows of files)
            path = os.path.join(d, "f.igd")
            synth.make_db(path, files=files, per_file=per, seed=int(rng.integers(1, 1 << 30)), nbp_log=b, genome=synth.SMALL,
                          clustered=cl, gtype=gtype)
            n = int(rng.choice([300, 5000, 70000, 400000]))
            ichr, qs, qe = batch(rng, n, 50_000_000, 1 << b)
            o = np.lexsort((qs, ichr))
            srt = (ichr[o], qs[o], qe[o])
            msg = []
            orc = Oracle(path)
            for build in BUILDS:
                # "D" (round 5): every promised-sorted batch takes the DIRECT step (engine/scan_direct.hpp) over the file's own tiles
                for k in ("IGD_HIP_RANK", "IGD_HIP_DIRECT", "IGD_HIP_NO_RETILE"): os.environ.pop(k, None)
                if build == "D": os.environ["IGD_HIP_DIRECT"] = "1"; os.environ["IGD_HIP_NO_RETILE"] = "1"
                elif build: os.environ["IGD_HIP_RANK"] = build
                db = Database(path)
                for v in (0, int(rng.choice([1, 300, 900]))):
                    want, wtot = orc.search(ichr, qs, qe, v)
                    for q, flags in ((srt, 1), (srt, 17), (srt, 0), ((ichr, qs, qe), 0), ((ichr, qs, qe), 2)):
                        if build and flags == 2: continue
                        if build == "D" and (q is not srt or flags == 0): continue
                        if build != "D" and flags == 17 and build: continue
                        got, gtot = db.search(*q, v, flags=flags)
                        if gtot != wtot or not np.array_equal(got, want):
                            msg.append("build=%r v=%d flags=%d sorted=%s: total %d vs %d" % (build, v, flags, q is srt, gtot, wtot))
                db.close()
            orc.close()
            print("case %d  b=%d files=%3d per=%5d gType%d %s n=%6d  %s" % (seed0 + ci, b, files, per, gtype, "clustered" if cl else "uniform  ", n,
                                                                          "ok" if not msg else "MISMATCH " + "; ".join(msg)), flush=True)
            bad += bool(msg)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    print("fuzz_engine: %d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
