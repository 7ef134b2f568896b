#!/usr/bin/env python3
"""Generate tests/golden/bench_checksums.json: per-file overlap counts of bench.py's workloads on the roadmap-scale
synthetic .igd, computed by the CPU ORACLE (oracle/, pinned to the reference) -- total and the position-weighted
checksum bench.py prints (sum hits[i] * (i + 1) mod 2^63).  bench.py and tests/test_gpu_stress.py compare the GPU's
counts of the same workloads with these numbers.  Runs without a GPU (about two minutes, 1 GB under /tmp/igdb).
    python tools/make_bench_checksums.py [--only NAME ...]      (--only: recompute these workloads, keep the others)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import Oracle          # noqa: E402  (test infrastructure: this generator is not product code)
from igd_amd import synth           # noqa: E402

PER_GPU = 12500000


def checksum(h):
    return int((h.astype(np.uint64) * (np.arange(len(h), dtype=np.uint64) + np.uint64(1))).sum() & np.uint64((1 << 63) - 1))


def stress_piled(span, Q=1000000):
    """bench.py's skew rows: 10^6 position-sorted queries inside `span` tiles of chr1 (numpy PCG64, seed 5, drawn in this order)."""
    rng = np.random.default_rng(5)
    out = {}
    for sp in (1, 10):
        ps = np.sort((50000000 + rng.integers(0, 16384 * sp, Q)).astype(np.int32))
        out[sp] = (np.zeros(Q, np.int32), ps, (ps + rng.integers(100, 2000, Q)).astype(np.int32))
    return out[span]


# the other databases bench.py's stress rows open (name -> igd_synth_db arguments); the same calls as in bench.py
OTHER_DBS = {
    "cl300x40000": dict(files=300, per_file=40000, seed=77, genome=synth.HG38, clustered=True),
    "sparse100x1000": dict(files=100, per_file=1000, seed=31, genome=synth.HG38),
    # the roadmap-scale database with real-data clustering: half of its 5e7 intervals around 2000 hot spots
    "clrm1900x26316": dict(files=1900, per_file=26316, seed=1000, genome=synth.HG38, clustered=True),
}


def main():
    path = "/tmp/igdb/rm1900x26316.igd"
    os.makedirs("/tmp/igdb", exist_ok=True)
    if not os.path.exists(path + ".done"):
        synth.make_db(path)
        open(path + ".done", "w").write("ok")
    out = {"database": "igd_synth_db(files=1900, per_file=26316, seed=1000, nbp_log=14, HG38)", "workloads": {}}
    base = synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=True)
    work = {
        "config2_sorted_q1000000": (base, (0, 500)),
        "config4_share_q12500000": (synth.make_queries_slab(PER_GPU, 0, PER_GPU, seed=7, genome=synth.HG38), (0, 500)),
        "config4_slab0_of_8": (synth.make_queries_slab(8 * PER_GPU, 0, PER_GPU, seed=7, genome=synth.HG38), (0, 500)),
        # config 4 is not pinned on slab 0 alone: a slab from the middle and the last one of the 10^8-query set
        "config4_slab3_of_8": (synth.make_queries_slab(8 * PER_GPU, 3 * PER_GPU, 4 * PER_GPU, seed=7, genome=synth.HG38), (0,)),
        "config4_slab7_of_8": (synth.make_queries_slab(8 * PER_GPU, 7 * PER_GPU, 8 * PER_GPU, seed=7, genome=synth.HG38), (0, 500)),
        # a query FILE beyond one device batch (2^24): what `bin/igd search -q` loops over (tests/test_gpu_batches.py)
        "cli_sorted_q20000000": (synth.make_queries(20000000, seed=11, genome=synth.HG38, sorted_=True), (0,)),
        "config4_slab0_of_2": (synth.make_queries_slab(2 * PER_GPU, 0, PER_GPU, seed=7, genome=synth.HG38), (0,)),
        "config4_slab1_of_2": (synth.make_queries_slab(2 * PER_GPU, PER_GPU, 2 * PER_GPU, seed=7, genome=synth.HG38), (0,)),
        "config4_slab0_of_4": (synth.make_queries_slab(4 * PER_GPU, 0, PER_GPU, seed=7, genome=synth.HG38), (0,)),
        # queries of 6 .. 13 tiles: coverage difference arrays + exact walk of the last tile
        "long_sorted_q100000": (synth.make_queries(100000, seed=7, genome=synth.HG38, min_len=100000, max_len=200000, sorted_=True), (0, 500)),
        # round 5: every row of bench.py's extra_configs is checked -- the small batches, the skew rows ...
        "small_sorted_q1000": (synth.make_queries(1000, seed=7, genome=synth.HG38, sorted_=True), (0,)),
        "small_sorted_q100000": (synth.make_queries(100000, seed=7, genome=synth.HG38, sorted_=True), (0,)),
        "piled_1tile_q1000000": (stress_piled(1), (0,)),
        "piled_10tiles_q1000000": (stress_piled(10), (0,)),
        # ... and the stress databases (`@name`: OTHER_DBS), each with the headline's 10^6 position-sorted queries
        "clustered_q1000000@cl300x40000": (base, (0,)),
        "clustered_roadmap_q1000000@clrm1900x26316": (base, (0, 500)),
        # (sparse: 59 % of the tiles are empty and the queries reach up to three tiles on -- rule NEST drops a third of what rule
        #  FLAT, `-v 1`, counts: quirk #1 of SURVEY.md at scale)
        "sparse_q1000000@sparse100x1000": (synth.make_queries(1000000, seed=7, genome=synth.HG38, min_len=100, max_len=40000, sorted_=True), (0, 1)),
    }
    dst = os.path.join(ROOT, "tests", "golden", "bench_checksums.json")
    only = sys.argv[sys.argv.index("--only") + 1:] if "--only" in sys.argv else None
    if only:
        out = json.load(open(dst))
        work = {k: w for k, w in work.items() if k in only}
    out["other_databases"] = {k: "igd_synth_db(%s)" % ", ".join("%s=%s" % kv for kv in sorted(v.items())) for k, v in OTHER_DBS.items()}
    oracles = {}
    for name, ((ichr, qs, qe), vs) in work.items():
        dbn = name.split("@")[1] if "@" in name else ""
        if dbn not in oracles:
            p = path
            if dbn:
                p = "/tmp/igdb/%s.igd" % dbn
                if not os.path.exists(p + ".done"):
                    synth.make_db(p, **OTHER_DBS[dbn])
                    open(p + ".done", "w").write("ok")
            oracles[dbn] = Oracle(p)
        orc = oracles[dbn]
        for v in vs:
            h, tot = orc.search(ichr, qs, qe, v)
            assert int(h.sum()) == tot
            out["workloads"]["%s_v%d" % (name.split("@")[0], v)] = {"queries": len(qs), "v": v, "total": int(tot), "checksum": checksum(h),
                                                                     "database": dbn or "rm1900x26316"}
            print(name, v, tot, flush=True)
    for orc in oracles.values():
        orc.close()
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    main()
