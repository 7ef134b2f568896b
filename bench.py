#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json: query-intervals/s of the overlap
search (igd search -q, hits-only) on a roadmap-scale synthetic .igd, on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of Q device-resident queries per GPU:
bucket (count/scan/scatter) -> igd_scan_tiles -> slab reduce, then (N>1) ONE RCCL all-reduce of
the nFiles-long int64 hits vector.  Weak scaling: every rank owns the same database and its
own Q queries.  Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     : dominant kernel (igd_scan_tiles) -- algorithmic bytes per launch (SURVEY.md 8d
                 terms, computed exactly on the GPU by igd_hip_batch_stats) / its HIP-event time
  cpu_baseline : the REAL reference `igd search -q` (oracle/_ref/igd, kind "reference") on the
                 same .igd and the same queries as BED text, 1 thread; falls back to the oracle
                 port (kind "port") when the prebuilt reference binary did not travel.
The oracle / reference are used here ONLY for that baseline and to check the GPU totals.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy rate


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def ensure_db(path, files, per_file, rank, world, barrier):
    """rank 0 generates the .igd (deterministic, ~20 s for the roadmap scale); others wait."""
    from igd_amd import synth
    done = path + ".done"
    if rank == 0 and not (os.path.exists(path) and os.path.exists(done)):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        t = time.time()
        synth.make_db(path, files=files, per_file=per_file, seed=1000, nbp_log=14, genome=synth.HG38)
        open(done, "w").write("ok")
        log("[bench] generated %s in %.1f s" % (path, time.time() - t))
    barrier()
    while not os.path.exists(done):
        time.sleep(0.2)


def cpu_baseline_allcores(exe, igd_path, bed_path, nq, expect_total, extra, repeats=3):
    """the reference has no threads and keeps its state in globals, so the faithful way to
    use more cores is one process per contiguous shard of the query file; wall time = slowest shard."""
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = max(1, min(ncores, 64))
    lines = open(bed_path, "rb").read().splitlines(keepends=True)
    shards = []
    for k in range(n):
        path = "%s.shard%d_of_%d" % (bed_path, k, n)
        with open(path, "wb") as fh:
            fh.writelines(lines[len(lines) * k // n:len(lines) * (k + 1) // n])
        shards.append(path)
    best, best_cpu, total = None, None, None
    for _ in range(repeats):
        outs = [open(sh + ".out", "wb") for sh in shards]
        t = time.perf_counter()
        procs = [subprocess.Popen([exe, "search", igd_path, "-q", sh] + extra, stdout=o, stderr=subprocess.DEVNULL)
                 for sh, o in zip(shards, outs)]
        cpu = 0.0
        for p in procs:                                     # rusage of each child: its own user+sys seconds
            _, _, ru = os.wait4(p.pid, 0)
            p.returncode = 0
            cpu = max(cpu, ru.ru_utime + ru.ru_stime)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
        best_cpu = cpu if best_cpu is None else min(best_cpu, cpu)
        total = 0
        for sh, o in zip(shards, outs):
            o.close()
            for line in open(sh + ".out").read().splitlines()[-2:]:
                if line.startswith("Total:"):
                    total += int(line.split(":")[1])
            os.unlink(sh + ".out")
    for sh in shards:
        os.unlink(sh)
    return {"value": nq / best, "unit": "query-intervals/s", "cores": n, "processes": n, "seconds": best,
            "slowest_shard_cpu_seconds": best_cpu, "value_if_spawn_were_free": nq / best_cpu if best_cpu else None,
            "totals_match_gpu": total == expect_total,
            "sample": "the same %d queries cut into %d contiguous shards, one `%s search -q` process per shard; "
                      "`seconds` = wall time from first spawn to last exit (dominated by spawning %d processes for a "
                      "0.3 s job), `slowest_shard_cpu_seconds` = largest user+sys of a shard; best of %d"
                      % (nq, n, os.path.basename(exe), n, repeats)}


def cpu_baseline(igd_path, bed_path, nq, expect_total, repeats=5, extra=()):
    """Reference CLI on the host, single thread, page cache warm, best of `repeats`."""
    ref = os.path.join(ROOT, "oracle", "_ref", "igd")
    port = os.path.join(ROOT, "oracle", "_build", "igd_oracle")
    if os.path.exists(ref) and len(igd_path) < 60:
        exe, kind = ref, "reference"
    else:
        if not os.path.exists(port):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        exe, kind = port, "port"
    best, total = None, None
    for _ in range(repeats):
        t = time.perf_counter()
        out = subprocess.run([exe, "search", igd_path, "-q", bed_path] + list(extra), stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, check=True).stdout
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
        for line in out.decode().splitlines()[-2:]:
            if line.startswith("Total:"):
                total = int(line.split(":")[1])
    ok = (total == expect_total)
    res = {"value": nq / best, "unit": "query-intervals/s", "cores": 1, "kind": kind,
           "sample": "all %d queries of the workload as BED text through `%s search -q%s`, "
                     "end to end (parse+search+print), best of %d, page cache warm; Total %s GPU (%s)"
                     % (nq, os.path.basename(exe), " " + " ".join(extra) if extra else "", repeats, "==" if ok else "!=", total),
           "seconds": best, "totals_match_gpu": ok}
    try:
        res["all_cores"] = cpu_baseline_allcores(exe, igd_path, bed_path, nq, expect_total, list(extra))
    except Exception as e:                                  # never lose the bench line to the side measurement
        res["all_cores"] = {"error": str(e)}
    return res


def box_copy_rate():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r01", "hbm_peak.json")))["copy_GBps_read_plus_write"]
    except Exception:
        return None


def measured_traffic(mode, args):
    """HBM bytes per launch of the dominant kernel from the PMC passes of tools/profile.sh
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of THIS command; FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for gfx950 streaming reads).  Counters cannot be collected
    from inside the benchmark process, so the committed summary is reported when it was taken with
    the same workload; otherwise null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        key = "%s/%s/q%d%s" % (mode, "shuffled" if args.shuffled else "sorted", args.queries, "/exact" if args.exact_arrays else "")
        return t.get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--queries", type=int, default=1000000, help="queries per GPU per step")
    ap.add_argument("--files", type=int, default=1900)
    ap.add_argument("--per-file", type=int, default=26316)
    ap.add_argument("--shuffled", action="store_true", help="queries in generation order, not position-sorted")
    ap.add_argument("--v", type=int, default=0, help="`-v N` signal filter (config 3)")
    ap.add_argument("--dir", default="/tmp/igdb")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--exact-arrays", action="store_true", help="read the 12-byte exact arrays, not the compact image")
    ap.add_argument("--grouping", choices=["default", "auto", "sorted", "bucket"], default="default",
                    help="how the engine groups queries by tile.  auto: the device checks the query order and picks "
                         "merge-join or bucketing (no assumption, ~5 gated no-op launches extra); sorted: the caller "
                         "promises (contig,start) order -- what a position-sorted BED is -- and the device VERIFIES it "
                         "in the timed region (a broken promise is an error, never a wrong count); bucket: always "
                         "counting-sort.  default = sorted for the position-sorted workload, auto with --shuffled")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from igd_amd import Database, synth
    from igd_amd.dist import allreduce_hits, init_from_env

    rank, world, local = init_from_env()
    if world != args.gpus:
        log("[bench] note: WORLD_SIZE=%d, --gpus=%d (using WORLD_SIZE)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: igd_amd has no CPU search path")
    if os.environ.get("IGD_BENCH_ONE_GPU"):      # N>1 control-flow test on a 1-GPU box (with IGD_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    def barrier():
        if world > 1:
            dist.barrier()

    igd_path = os.path.join(args.dir, "rm%dx%d.igd" % (args.files, args.per_file))
    ensure_db(igd_path, args.files, args.per_file, rank, world, barrier)

    t = time.time()
    db = Database(igd_path, device=local)
    open_s = time.time() - t
    Q = args.queries
    ichr, qs, qe = synth.make_queries(Q, seed=7 + rank, genome=synth.HG38, sorted_=not args.shuffled)
    d_ichr = torch.from_numpy(ichr).to(dev)
    d_qs = torch.from_numpy(qs).to(dev)
    d_qe = torch.from_numpy(qe).to(dev)
    d_hits = torch.zeros(max(db.nfiles, 1), dtype=torch.int64, device=dev)
    # One explicit (non-default) stream carries everything: torch's memsets/adds, the engine's
    # kernels (it enqueues on the hipStream_t it is given) and the RCCL all-reduce.
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    if args.grouping == "default":
        args.grouping = "auto" if args.shuffled else "sorted"
    gflags = {"auto": 0, "sorted": 1, "bucket": 2}[args.grouping] | (4 if args.exact_arrays else 0)

    # A job = K batches (steps) whose per-file counts ACCUMULATE in hits[] -- the reference's hits[] is
    # one accumulator over the whole query file (src/igd_search.c:925,1032-1039) and the engine adds --
    # followed by the path's ONE exchange: a SUM all-reduce of hits[nFiles] (no-op at N=1).  Both are
    # inside the timed region.
    def step():
        db.search_dev(d_ichr.data_ptr(), d_qs.data_ptr(), d_qe.data_ptr(), Q, d_hits.data_ptr(), None,
                      v=args.v, stream=stream, flags=gflags)

    for _ in range(args.warmup):
        step()
    allreduce_hits(d_hits)                  # also warms RCCL up (first call builds the communicator)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    db.profile_begin(args.steps)
    t0 = time.perf_counter()
    d_hits.zero_()
    for _ in range(args.steps):
        step()
    allreduce_hits(d_hits)                  # the one collective of the path
    torch.cuda.synchronize(dev)
    barrier()
    t1 = time.perf_counter()
    db.sync(stream)                      # surfaces a broken --grouping sorted promise
    prof = db.profile_end()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    # exact algorithmic work of one launch on this rank's batch (outside the timed region)
    st = db.batch_stats(d_ichr.data_ptr(), d_qs.data_ptr(), d_qe.data_ptr(), Q, v=args.v)
    mode = "v" if (args.v > 0 and db.gtype == 1) else "hits"
    algo_bytes = db.algorithmic_bytes(st, Q, mode)
    hits_job = d_hits.cpu().numpy()       # K identical batches per rank, summed over ranks
    assert args.steps > 0 and (hits_job % args.steps == 0).all(), "hits[] is not K times one batch"
    hits_one = hits_job // args.steps     # one batch per rank, summed over ranks

    if rank == 0:
        value = world * Q * args.steps / elapsed
        scan_s = prof["scan_ms"] * 1e-3
        achieved = algo_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
        traffic = measured_traffic(mode, args)
        line = {
            "metric": "query-intervals/sec (igd search -q, hits-only) on roadmap-scale synthetic .igd",
            "value": value, "unit": "query-intervals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "roadmap-scale synthetic IGD: %d files x %d intervals (%d tile records, "
                                   "%d tiles of 16384 bp, hg38 contigs) + %d %s queries per GPU per step, "
                                   "%s, queries and DB resident in HBM"
                                   % (args.files, args.per_file, db.nrecords, db.ntiles, Q,
                                      "generation-order" if args.shuffled else "position-sorted",
                                      "-v %d signal filter" % args.v if mode == "v" else "hits-only"),
                       "queries_per_gpu": Q, "nfiles": db.nfiles, "parallelism": "query-sharded x%d" % world, "grouping": args.grouping,
                       "collective": ("ONE sum all-reduce of int64[%d] per job (after the %d batches), inside the timed region"
                                      % (db.nfiles, args.steps)) if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "igd_scan_tiles", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # the same kernel time against the MEASURED HBM bytes (PMC), next to the algorithmic figure
                         "traffic_rate": (traffic / scan_s / 1e9) if (traffic and scan_s > 0) else None,
                         "traffic_frac": (traffic / scan_s / 1e9 / HBM_PEAK_GBS) if (traffic and scan_s > 0) else None,
                         "box_copy_rate": box_copy_rate(),   # GB/s a device-to-device copy reaches on this class of box (tools/hbm_peak.py)
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "bytes_per_query": algo_bytes / Q, "kernel_ms": prof["scan_ms"],
                         "pipeline_ms": prof["pipeline_ms"], "launches_timed": prof["launches"],
                         "work": st},
            "hits_per_step_total": int(hits_one.sum()),
            "db_open_s": open_s,
        }
        if world == 1 and not args.no_cpu:
            bed = os.path.join(args.dir, "q%d_%s.bed" % (Q, "shuf" if args.shuffled else "sorted"))
            if not os.path.exists(bed):
                synth.write_bed(bed, synth.HG38, ichr, qs, qe)
            extra = ["-v", str(args.v)] if mode == "v" else []
            line["cpu_baseline"] = cpu_baseline(igd_path, bed, Q, int(hits_one.sum()), extra=extra)
        print(json.dumps(line), flush=True)
    db.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
