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


def main():
    path = "/tmp/igdb/rm1900x26316.igd"
    if not os.path.exists(path + ".done"):
        os.makedirs("/tmp/igdb", exist_ok=True)
        synth.make_db(path)
        open(path + ".done", "w").write("ok")
    orc = Oracle(path)
    out = {"database": "igd_synth_db(files=1900, per_file=26316, seed=1000, nbp_log=14, HG38)", "workloads": {}}
    work = {
        "config2_sorted_q1000000": (synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=True), (0, 500)),
        "config4_share_q12500000": (synth.make_queries_slab(PER_GPU, 0, PER_GPU, seed=7, genome=synth.HG38), (0, 500)),
        "config4_slab0_of_8": (synth.make_queries_slab(8 * PER_GPU, 0, PER_GPU, seed=7, genome=synth.HG38), (0, 500)),
        # config 4 is not pinned on slab 0 alone: a slab from the middle and the last one of the 10^8-query set
        "config4_slab3_of_8": (synth.make_queries_slab(8 * PER_GPU, 3 * PER_GPU, 4 * PER_GPU, seed=7, genome=synth.HG38), (0,)),
        "config4_slab7_of_8": (synth.make_queries_slab(8 * PER_GPU, 7 * PER_GPU, 8 * PER_GPU, seed=7, genome=synth.HG38), (0, 500)),
        # a query FILE beyond one device batch (2^24): what `bin/igd search -q` loops over (tests/test_gpu_batches.py)
        "cli_sorted_q20000000": (synth.make_queries(20000000, seed=11, genome=synth.HG38, sorted_=True), (0,)),
        "config4_slab0_of_2": (synth.make_queries_slab(2 * PER_GPU, 0, PER_GPU, seed=7, genome=synth.HG38), (0,)),
        "config4_slab1_of_2": (synth.make_queries_slab(2 * PER_GPU, PER_GPU, 2 * PER_GPU, seed=7, genome=synth.HG38), (0,)),
        # queries of 6 .. 13 tiles: coverage difference arrays + exact walk of the last tile
        "long_sorted_q100000": (synth.make_queries(100000, seed=7, genome=synth.HG38, min_len=100000, max_len=200000, sorted_=True), (0, 500)),
    }
    dst = os.path.join(ROOT, "tests", "golden", "bench_checksums.json")
    only = sys.argv[sys.argv.index("--only") + 1:] if "--only" in sys.argv else None
    if only:
        out = json.load(open(dst))
        work = {k: w for k, w in work.items() if k in only}
    for name, ((ichr, qs, qe), vs) in work.items():
        for v in vs:
            h, tot = orc.search(ichr, qs, qe, v)
            assert int(h.sum()) == tot
            out["workloads"]["%s_v%d" % (name, v)] = {"queries": len(qs), "v": v, "total": int(tot), "checksum": checksum(h)}
            print(name, v, tot, flush=True)
    orc.close()
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    main()
