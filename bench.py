#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json: query-intervals/s of the overlap
search (igd search -q, hits-only) on a roadmap-scale synthetic .igd, on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of Q device-resident queries per GPU
(k_query_bounds -> igd_scan_sorted -> k_reduce_slabs; the bucket kernels and igd_scan_tiles when the
batch is not ordered); a job = K steps accumulating into hits[] + (N>1) ONE RCCL all-reduce of the
nFiles-long int64 hits vector, all inside the timed region.

Workloads (BASELINE.json `configs`):
  N = 1  config 2: the roadmap-scale .igd + 10^6 position-sorted queries (seed 7);
  N > 1  config 4: ONE position-sorted set of N x 1.25e7 queries (seed 7), rank r takes the r-th
         contiguous slab (igd_amd.dist.shard_bounds) -- at N = 8 that is 10^8 queries; weak scaling.
N > 1 is one process per GPU.  Started by the driver through torch.distributed.run the ranks find
RANK/WORLD_SIZE/MASTER_* in the environment; started as plain `python bench.py --gpus N` this
process spawns the N ranks itself -- BEFORE it touches the GPU or imports torch -- and relays
rank 0's JSON line.

Prints ONE compact JSON line on rank 0 (contract in the task statement; at most LINE_CAP = 8 KB -- the driver keeps the last ~8 KB of
stdout -- asserted here and in tests/test_gpu_bench_line.py) and writes the whole record, every side measurement in full, to
bench_extra.json (--extra-out).  Both carry
  roofline     : dominant kernel (igd_scan_sorted; igd_scan_tiles on the bucket path).  `achieved` / `frac` price its HIP-event time
                 against the COMPULSORY bytes of the batch, computed in this run by the engine
                 (igd_hip_batch_traffic: every visited unit's records once in the bytes of the image
                 read + descriptors + queries + counter rows) -- a fraction that cannot exceed 1.  The
                 algorithmic bytes of SURVEY.md 8(d) (what the REFERENCE's algorithm touches, computed
                 exactly on the GPU by igd_hip_batch_stats) are reported next to it as algorithmic_*;
                 `box` = what a float4 copy / read kernel and pinned PCIe copies reach on this box now.
  cpu_baseline : the REAL reference `igd search -q` (oracle/_ref/igd, kind "reference") on the
                 same .igd and the same queries as BED text, 1 thread; falls back to the oracle
                 port (kind "port") when the prebuilt reference binary did not travel.
  extra_configs: short timed runs of configs 3 (`-v 500`), shuffled input, config 4's per-GPU share
                 (1.25e7 queries on one GPU), one GPU's slab of the 2- / 4- / 8-GPU job, two small batches, the stress shapes
                 (skew, long queries, clustered and sparse databases) and config 5 (`-f`, 8 and 16 bytes per overlap)
                 in the same process (N = 1 only), each checked against the oracle's committed counts; the compact line
                 keeps key, ms_per_step, kernel_ms, frac and matches_oracle of every row.
  cli_end_to_end: the product command line on the workload's files, every size on BOTH routes (engine = MI355X, host = CPU
                 threads for small files), each labelled with who counted.
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

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the box's own copy rate is measured below
CONFIG4_PER_GPU = 12500000


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--queries", type=int, default=0,
                    help="queries per GPU per step (default: 10^6 at N=1 = config 2; 1.25e7 at N>1 = config 4)")
    ap.add_argument("--files", type=int, default=1900)
    ap.add_argument("--per-file", type=int, default=26316)
    ap.add_argument("--shuffled", action="store_true", help="queries in generation order, not position-sorted (N=1)")
    ap.add_argument("--v", type=int, default=0, help="`-v N` signal filter (config 3)")
    ap.add_argument("--dir", default="/tmp/igdb")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--extra-out", default="", help="where the whole record goes (default: bench_extra.json beside bench.py); stdout carries "
                                                     "ONE compact line of at most 8 KB")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs runs")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-cache runs of the headline batch (profiling runs: their launches would be averaged in)")
    ap.add_argument("--slab-of", type=int, default=0, metavar="G",
                    help="N=1 only: this GPU's step is slab 0 of a G-GPU config-4 job (G x --queries position-sorted queries) -- "
                         "what one of G GPUs would run, measured without the other G-1")
    ap.add_argument("--exact-arrays", action="store_true", help="read the 12-byte exact arrays, not the compact image")
    ap.add_argument("--long-queries", action="store_true", help="do not pass IGD_HIP_FLAG_SHORT (A/B: dense sorted batches then take the ordinary step, with k_query_bounds)")
    ap.add_argument("--query-layout", choices=["runs", "ichr"], default="ichr",
                    help="how a position-sorted batch under the order promise is resident: one contig number per query (12 B/query, the "
                         "default) or contig runs + starts + ends (8 B/query, igd_hip_search_runs_dev: measured no faster -- the grouping "
                         "kernel is not bound by the bytes it reads, DESIGN.md section 4)")
    ap.add_argument("--grouping", choices=["default", "auto", "sorted", "bucket"], default="default",
                    help="how the engine groups queries by tile.  auto: the device checks the query order and picks "
                         "merge-join or bucketing (no assumption, ~5 gated no-op launches extra); sorted: the caller "
                         "promises (contig,start) order -- what a position-sorted BED is -- and the device VERIFIES it "
                         "in the timed region (a broken promise is an error, never a wrong count); bucket: always "
                         "counting-sort, no order check (what the CLI passes when its parser saw a line out of order).  default = sorted for "
                         "the position-sorted workload, bucket with --shuffled")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: this process only spawns (no torch, no HIP call here, ever)
def spawn_ranks(n, argv):
    """Start the N ranks as fresh child processes and watch ALL of them: when one exits non-zero (no device, out of memory,
    a failed RCCL init) the others would sit in dist.barrier() until the collective's timeout -- they are terminated at once
    and this process exits non-zero with the failed rank's stderr tail, so that the caller sees an error, not a hang."""
    import socket
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        ef = tempfile.TemporaryFile(mode="w+b")
        errs.append(ef)
        # rank 0 owns stdout (the JSON line); the other ranks' stdout goes to stderr; every rank's stderr is kept for the report
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr, stderr=ef))
    failed, rc = None, 0
    deadline = time.time() + float(os.environ.get("IGD_BENCH_TIMEOUT_S", "1500"))
    while True:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed, rc = bad[0], codes[bad[0]]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            failed, rc = [r for r, c in enumerate(codes) if c is None][0], 124
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    for r, ef in enumerate(errs):
        ef.seek(0)
        text = ef.read().decode(errors="replace")
        if failed is None or r == failed:
            sys.stderr.write(text if failed is None else "".join(text.splitlines(True)[-40:]))
        ef.close()
    if failed is not None:
        log("[bench] rank %d %s (code %s); the other %d rank(s) were stopped" %
            (failed, "timed out" if rc == 124 else "failed", rc, n - 1))
        return abs(rc) or 1
    return 0


# ------------------------------------------------------------------------------------------------
def ensure_db(path, files, per_file, rank, barrier):
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


def cpu_info():
    """CPU model and core counts of the host the baselines run on (state N and the CPU model)."""
    model, phys = None, set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid)); pid = cid = None
        if pid is not None:
            phys.add((pid, cid))
    except Exception:
        pass
    return {"cpu_model": model, "physical_cores": len(phys) or None, "logical_cpus": os.cpu_count(),
            "cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}


def cli_end_to_end(igd_path, bed_path, expect_total, repeats=5):
    """Wall time of the PRODUCT command line -- load the .igd, upload, parse, kernels, print -- on the headline's files,
    next to the reference CLI's seconds in cpu_baseline (report it separately): best of `repeats`."""
    exe = os.path.join(ROOT, "bin", "igd")
    limit = host_route_limit()
    nq = sum(1 for _ in open(bed_path, "rb"))
    out = {"command": "bin/igd search <db> -q <the headline's 10^6-query BED> [..]", "repeats": repeats, "host_route_limit": limit,
           "default_route": "host" if (limit and nq <= limit) else "engine",
           "routes": "every size is timed on BOTH routes: *_seconds = the tool's own routing (default_route says who counted: `host` = "
                     "CPU threads, igd_hostpath.c -- NOT an MI355X number), *_engine_seconds / q_engine_only_* = the same command with "
                     "IGD_HOST_MAX_QUERIES=0 (the GPU engine whatever the size)"}
    eng_env = dict(os.environ, IGD_HOST_MAX_QUERIES="0")
    for name, extra in (("q", []), ("q_v500", ["-v", "500"]), ("q_f", ["-f"])):
        for route, env in (("", None), ("_engine", eng_env)):
            if name == "q" and route:
                continue                                    # (timed below with the phase table)
            best, worst, ok = None, None, None
            for _ in range(repeats if name != "q_f" else 3):
                t = time.perf_counter()
                p = subprocess.run([exe, "search", igd_path, "-q", bed_path] + extra, env=env,
                                   stdout=subprocess.DEVNULL if name == "q_f" else subprocess.PIPE, stderr=subprocess.DEVNULL)
                dt = time.perf_counter() - t
                if p.returncode != 0:
                    best = None
                    break
                best = dt if best is None else min(best, dt)
                worst = dt if worst is None else max(worst, dt)
                if name == "q":
                    tot = [int(l.split(":")[1]) for l in p.stdout.decode().splitlines()[-2:] if l.startswith("Total:")]
                    ok = bool(tot) and tot[0] == expect_total
            out[name + route + "_seconds"] = best
            out[name + route + "_seconds_slowest"] = worst
            if ok is not None:
                out["q_total_matches_gpu"] = ok
    out["q_f_route"] = "engine" if not (limit is not None and 0 < nq <= host_route_limit_enum()) else "host"
    # the same file sent to the GPU whatever its size, with the tool's own phase table (IGD_TIMING) of the fastest and the
    # slowest of the repeats: what a search costs before its first kernel varies from run to run on one box
    try:
        runs = []
        for _ in range(repeats):
            t = time.perf_counter()
            p = subprocess.run([exe, "search", igd_path, "-q", bed_path], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, IGD_HOST_MAX_QUERIES="0", IGD_TIMING="1"))
            dt = time.perf_counter() - t
            if p.returncode == 0:
                runs.append((dt, [l for l in p.stderr.decode().splitlines() if l.startswith("[igd timing]")]))
        if runs:
            runs.sort(key=lambda r: r[0])
            out["q_engine_only_seconds"] = runs[0][0]
            out["q_engine_only_seconds_slowest"] = runs[-1][0]
            out["q_engine_only_seconds_median"] = runs[len(runs) // 2][0]
            out["q_engine_only_phases_fastest"] = runs[0][1]
            out["q_engine_only_phases_slowest"] = runs[-1][1]
        try:
            pc = open("/proc/meminfo").read()
            out["page_cache_note"] = "the .igd (%d MB) was written and read by this process before: served from the page cache; %s" % (
                os.path.getsize(igd_path) >> 20, [l for l in pc.splitlines() if l.startswith("Cached:")][0])
        except Exception:
            pass
    except Exception as e:
        out["q_engine_only_error"] = str(e)
    # Small query files: the reference starts cheaply (header only, then the tiles the queries touch); files of at most
    # IGD_HOST_MAX_QUERIES lines are counted on the host by product code (igd_hostpath.c), larger ones go to the engine.
    # Wall time of `search -q` at every size, the reference binary's beside it, stdout compared byte for byte.
    try:
        ref = os.path.join(ROOT, "oracle", "_ref", "igd")
        synth_exe = os.path.join(ROOT, "bin", "igd_synth")
        rows = []
        for n in (1000, 10000, 100000, 300000, 1000000, 3000000):
            q = os.path.join(os.path.dirname(bed_path), "q%d.bed" % n)
            if not os.path.exists(q):
                subprocess.check_call([synth_exe, "queries", q, "--n", str(n)], stdout=subprocess.DEVNULL)
            row = {"queries": n, "product_route": "host" if (limit and n <= limit) else "engine"}
            outs = {}
            for who, cmd, env in (("reference", [ref], None), ("product", [exe], None),
                                  ("product_engine_only", [exe], dict(os.environ, IGD_HOST_MAX_QUERIES="0"))):
                if who == "reference" and not os.path.exists(ref):
                    continue
                best = None
                for _ in range(5 if n <= 1000000 else 3):
                    t = time.perf_counter()
                    p = subprocess.run(cmd + ["search", igd_path, "-q", q], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
                    dt = time.perf_counter() - t
                    if p.returncode != 0:
                        best = None
                        break
                    best = dt if best is None else min(best, dt)
                    outs[who] = p.stdout
                row[who + "_seconds"] = best
            row["stdout_identical"] = len(set(outs.values())) == 1 and len(outs) >= 2
            rows.append(row)
        out["small_files"] = rows
        out["small_files_note"] = ("files of at most igdc_host_limit() queries (25 000 per usable host thread; IGD_HOST_MAX_QUERIES overrides) are "
                                   "counted on the host (igd_hostpath.c, product code): product_route says who counted product_seconds; "
                                   "product_engine_only = the same file sent to the GPU")
    except Exception as e:
        out["small_files"] = {"error": str(e)}
    return out


def cpu_baseline_allcores_large(exe, igd_path, genome, queries, expect_total, workdir):
    """The host's cores on a batch large enough that starting the processes is a small part of the wall time: config 4's
    per-GPU share (1.25e7 position-sorted queries) cut into 256 contiguous shards, one reference process per shard, P of
    them at a time (ONE xargs -P: the benchmark process itself -- torch loaded -- forks slowly) for P = 16, 32, 64 and the
    number of physical cores; the best P is the reported value, the whole table is kept (on the pool's hosts more processes
    than ~32 at a time run SLOWER: they share one page cache / one container's CPU quota)."""
    from igd_amd import synth
    ichr, qs, qe = queries
    n = len(qs)
    info = cpu_info()
    allowed = info["cpus_allowed"] or info["logical_cpus"] or 1
    nshards = 256
    shards = []
    for r in range(nshards):
        lo, hi = n * r // nshards, n * (r + 1) // nshards
        path = os.path.join(workdir, "ac%d_%d.bed" % (r, nshards))
        synth.write_bed(path, genome, ichr[lo:hi], qs[lo:hi], qe[lo:hi])
        shards.append(path)
    listing = ("\n".join(shards) + "\n").encode()

    def run(k, what):
        t = time.perf_counter()
        subprocess.run(["xargs", "-P", str(k), "-I", "{}", "sh", "-c", what], input=listing, check=True, stderr=subprocess.DEVNULL)
        return time.perf_counter() - t

    table, best = [], None
    for k in sorted(set(min(allowed, x) for x in (16, 32, 64, info["physical_cores"] or 64))):
        dt = min(run(k, "'%s' search '%s' -q {} > {}.out" % (exe, igd_path)) for _ in range(2))
        total = 0
        for sh in shards:
            for line in open(sh + ".out").read().splitlines()[-2:]:
                if line.startswith("Total:"):
                    total += int(line.split(":")[1])
        row = {"processes_at_a_time": k, "seconds": dt, "value": n / dt, "spawn_seconds": min(run(k, "true {}") for _ in range(2)),
               "totals_match_oracle_fixture": total == expect_total}
        table.append(row)
        if best is None or dt < best["seconds"]:
            best = row
    for sh in shards:
        for f in (sh, sh + ".out"):
            try:
                os.unlink(f)
            except OSError:
                pass
    return {"value": best["value"], "unit": "query-intervals/s", "cores": best["processes_at_a_time"], "processes": nshards,
            "seconds": best["seconds"], "queries": n, "spawn_seconds": best["spawn_seconds"],
            "spawn_fraction": best["spawn_seconds"] / best["seconds"], "totals_match_oracle_fixture": best["totals_match_oracle_fixture"],
            "by_processes_at_a_time": table, "host": info,
            "sample": "config 4's per-GPU share (%d position-sorted queries) cut into %d contiguous shards, one `%s search -q` process "
                      "per shard, P at a time under one xargs -P; wall time from start to the last exit, best of 2 per P; value = the "
                      "best P; spawn_seconds = the same xargs starting `true`" % (n, nshards, os.path.basename(exe))}


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
    res = {"value": nq / best, "unit": "query-intervals/s", "cores": 1, "kind": kind, "exe": exe, "host": cpu_info(),
           "sample": "all %d queries of the workload as BED text through `%s search -q%s`, "
                     "end to end (parse+search+print), best of %d, page cache warm; Total %s GPU (%s)"
                     % (nq, os.path.basename(exe), " " + " ".join(extra) if extra else "", repeats, "==" if ok else "!=", total),
           "seconds": best, "totals_match_gpu": ok}
    try:
        res["all_cores"] = cpu_baseline_allcores(exe, igd_path, bed_path, nq, expect_total, list(extra))
    except Exception as e:                                  # never lose the bench line to the side measurement
        res["all_cores"] = {"error": str(e)}
    return res


def golden_counts(key):
    """Oracle-computed (total, checksum) of a bench workload: tests/golden/bench_checksums.json, written by
    tools/make_bench_checksums.py from the CPU oracle.  A data fixture -- nothing of oracle/ runs here."""
    try:
        w = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_checksums.json")))["workloads"].get(key)
        return (w["total"], w["checksum"]) if w else None
    except Exception:
        return None


def hits_checksum(hits_one):
    import numpy as np
    return int((hits_one.astype(np.uint64) * (np.arange(len(hits_one), dtype=np.uint64) + np.uint64(1))).sum() & np.uint64((1 << 63) - 1))


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/profile.sh:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950 streaming reads).  Counters cannot be collected from inside the benchmark
    process: this is a CONSTANT of a committed profile of the same workload, tagged with its source."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path)).get(key)
        if t:
            return t.get("hbm_bytes_per_launch"), "profiles/traffic.json[%s] (%s)" % (key, t.get("source", "committed rocprofv3 --pmc run"))
    except Exception:
        pass
    return None, None


class Job:
    """One resident batch + the timed loop over it (shared by the headline run and extra_configs)."""

    def __init__(self, db, dev, stream, ichr, qs, qe, v, gflags, layout="ichr"):
        import torch
        self.db, self.dev, self.stream, self.v, self.gflags = db, dev, stream, v, gflags
        self.Q = len(qs)
        # A position-sorted batch under the order promise is resident as (contig runs, starts, ends): run_start[nCtg + 1]
        # instead of one contig number per query -- what a position-sorted BED is, and 4 bytes per query less for the
        # grouping kernel to read (igd_hip_search_runs_dev).  The ichr[] form stays resident for the instrumentation calls.
        self.d_runs = None
        if layout == "runs" and (gflags & 1) and not (gflags & 2):
            try:
                self.d_runs = torch.from_numpy(db.contig_runs(ichr, db.nctg)).to(dev)
            except Exception:
                self.d_runs = None                          # (unknown contigs in the batch: no run table)
        self.d_ichr = torch.from_numpy(ichr).to(dev)
        self.d_qs = torch.from_numpy(qs).to(dev)
        self.d_qe = torch.from_numpy(qe).to(dev)
        self.d_hits = torch.zeros(max(db.nfiles, 1), dtype=torch.int64, device=dev)

    def step(self, zero_first=False):
        """One pass of the hot path over the resident batch; zero_first: the batch's first kernel clears hits[]
        (IGD_HIP_FLAG_ZERO_FIRST) -- the job's accumulator starts from zero without a launch of its own."""
        if self.d_runs is not None:
            self.db.search_runs_dev(self.d_runs.data_ptr(), self.d_qs.data_ptr(), self.d_qe.data_ptr(), self.Q,
                                    self.d_hits.data_ptr(), None, v=self.v, stream=self.stream, flags=self.gflags | (8 if zero_first else 0))
            return
        self.db.search_dev(self.d_ichr.data_ptr(), self.d_qs.data_ptr(), self.d_qe.data_ptr(), self.Q,
                           self.d_hits.data_ptr(), None, v=self.v, stream=self.stream, flags=self.gflags | (8 if zero_first else 0))

    def run(self, steps, warmup, barrier=lambda: None, collective=lambda t: None):
        """W untimed steps, then exactly K steps + the path's one collective between barrier+synchronize."""
        import torch
        for _ in range(warmup):
            self.step()
        collective(self.d_hits)                 # also warms RCCL up (first call builds the communicator)
        torch.cuda.synchronize(self.dev)
        barrier()
        torch.cuda.synchronize(self.dev)
        # the dominant kernel is timed with HIP events on its own stream on every 4th step of the timed region (an event
        # is one more packet between two kernels: timing every step costs the job ~4 us per step)
        self.db.profile_begin(steps, every=4 if steps >= 16 else 1)
        # (polling the stream with hipStreamQuery instead of sleeping on its signal was measured and is OFF: 89-94 us per step
        # against 87-88 at --steps 20 -- the polling thread gets in the way of the runtime's own completion handling)
        spin = bool(os.environ.get("IGD_BENCH_SPIN"))
        t0 = time.perf_counter()
        for k in range(steps):
            self.step(zero_first=(k == 0))      # hits[] of the job starts from zero: cleared by the first batch's first kernel
        collective(self.d_hits)                 # the one collective of the path
        if spin:
            self.db.sync(self.stream, spin=True)
        torch.cuda.synchronize(self.dev)
        barrier()
        t1 = time.perf_counter()
        self.db.sync(self.stream)               # surfaces a broken --grouping sorted promise
        prof = self.db.profile_end()
        return t1 - t0, prof

    def run_cold(self, steps=12, dirty=False):
        """The same batch with the last-level cache emptied before every launch: what ONE `igd search` sees (the timed
        loop re-streams the same 319 MB image against a 256 MiB Infinity Cache).  Between two launches, on the same
        stream and outside the kernel's event pair, 512 MiB of other memory is read."""
        import torch
        evict = torch.zeros(128 << 20, dtype=torch.int32, device=self.dev)     # 512 MiB
        sink = torch.zeros(1, dtype=torch.int64, device=self.dev)
        def flush():
            nonlocal sink
            if dirty:
                evict.add_(1)                               # read AND written: the cache is left full of dirty lines, whose
            else:                                           # write-back then runs while the kernel under test does
                sink += evict.sum()                         # read only: clean lines
        for _ in range(2):
            flush()
            self.step()
        torch.cuda.synchronize(self.dev)
        self.db.profile_begin(steps, every=1)
        for _ in range(steps):
            flush()
            self.step()
        torch.cuda.synchronize(self.dev)
        self.db.sync(self.stream)
        prof = self.db.profile_end()
        del evict
        return prof

    def roofline(self, prof, traffic_key=None):
        db = self.db
        p = (self.d_ichr.data_ptr(), self.d_qs.data_ptr(), self.d_qe.data_ptr())
        st = db.batch_stats(*p, self.Q, v=self.v)
        mode = "v" if (self.v > 0 and db.gtype == 1) else "hits"
        algo = db.algorithmic_bytes(st, self.Q, mode)
        tr = db.batch_traffic(*p, self.Q, v=self.v, flags=self.gflags)
        kernel = db.last_scan_kernel()          # of the traffic batch just run: same arrays, same flags as the timed steps
        scan_s = prof["scan_ms"] * 1e-3
        ach = tr["total"] / scan_s / 1e9 if scan_s > 0 else 0.0
        pmc, src = pmc_traffic(traffic_key) if traffic_key else (None, None)
        return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "bytes_per_launch": tr["total"], "bytes_source": "compulsory traffic computed in this run (igd_hip_batch_traffic)",
                "bytes_breakdown": tr,
                "traffic": pmc, "traffic_source": src,
                "algorithmic_bytes_per_launch": algo, "algorithmic_achieved": algo / scan_s / 1e9 if scan_s > 0 else 0.0,
                "algorithmic_frac": algo / scan_s / 1e9 / HBM_PEAK_GBS if scan_s > 0 else 0.0,
                "algorithmic_bytes_per_query": algo / self.Q,
                "kernel_ms": prof["scan_ms"], "pipeline_ms": prof["pipeline_ms"], "launches_timed": prof["launches"],
                "work": st}


def extra_configs(db, dev, stream, args, box):
    """Short driver-visible runs of the other single-GPU configurations (BASELINE configs 3, 4-share, 5)."""
    import numpy as np
    from igd_amd import synth
    out = []
    Q = 1000000
    # config 4's batches go in under BOTH verified promises -- position-sorted (1) and no query longer than a tile (16: the
    # generator's queries are 100 .. 1999 bp) -- and, being dense, take the engine's DIRECT step (no per-query pre-pass)
    SHORT = 1 | 16
    base = synth.make_queries(Q, seed=7, genome=synth.HG38, sorted_=True)
    shuf = synth.make_queries(Q, seed=7, genome=synth.HG38, sorted_=False)
    dense = synth.make_queries_slab(CONFIG4_PER_GPU, 0, CONFIG4_PER_GPU, seed=7, genome=synth.HG38)
    slab8 = synth.make_queries_slab(8 * CONFIG4_PER_GPU, 0, CONFIG4_PER_GPU, seed=7, genome=synth.HG38)
    cases = [("config 3: -v 500, 10^6 position-sorted queries", base, 500, 1, 100, "config2_sorted_q1000000_v500"),
             ("10^6 queries in generation order (flags = IGD_HIP_FLAG_BUCKET, what the command line tool's parser passes for a file it has seen out of order)",
              shuf, 0, 2, 100, "shuffled_bucket_q1000000_v0=config2_sorted_q1000000_v0"),
             ("the same 10^6 queries in generation order with flags = 0: the device checks the order itself and then takes the bucket path", shuf, 0, 0, 100,
              "shuffled_auto_q1000000_v0=config2_sorted_q1000000_v0"),
             ("config 4 per-GPU share: 1.25e7 position-sorted queries in one batch on one GPU", dense, 0, SHORT, 30, "config4_share_q12500000_v0"),
             ("config 4 as one of 8 GPUs sees it: slab 0 (1.25e7 queries) of the 10^8 position-sorted queries", slab8, 0, SHORT, 30,
              "config4_slab0_of_8_v0"),
             ("config 4 as one of 4 GPUs sees it: slab 0 (1.25e7 queries) of 5e7 position-sorted queries",
              synth.make_queries_slab(4 * CONFIG4_PER_GPU, 0, CONFIG4_PER_GPU, seed=7, genome=synth.HG38), 0, SHORT, 30, "config4_slab0_of_4_v0"),
             ("config 4 as one of 2 GPUs sees it: slab 0 (1.25e7 queries) of 2.5e7 position-sorted queries",
              synth.make_queries_slab(2 * CONFIG4_PER_GPU, 0, CONFIG4_PER_GPU, seed=7, genome=synth.HG38), 0, SHORT, 30, "config4_slab0_of_2_v0"),
             ("small batch: 10^3 position-sorted queries per step (latency of one pass)",
              synth.make_queries(1000, seed=7, genome=synth.HG38, sorted_=True), 0, 1, 200, "small_sorted_q1000_v0"),
             ("small batch: 10^5 position-sorted queries per step", synth.make_queries(100000, seed=7, genome=synth.HG38, sorted_=True),
              0, 1, 200, "small_sorted_q100000_v0")]
    # stress shapes: queries piled up in one / ten tiles of the same database (the skew valves' work) ...
    rng = np.random.default_rng(5)
    for span in (1, 10):
        ps = np.sort((50000000 + rng.integers(0, 16384 * span, Q)).astype(np.int32))
        cases.append(("stress: 10^6 position-sorted queries inside %d tile%s of chr1 (skew valve)" % (span, "" if span == 1 else "s"),
                      (np.zeros(Q, np.int32), ps, (ps + rng.integers(100, 2000, Q)).astype(np.int32)), 0, 1, 10,
                      "piled_%s_q1000000_v0" % ("1tile" if span == 1 else "10tiles")))
    # ... queries of 6 .. 13 tiles each (difference arrays over the tiles they cover whole + an exact walk of the last tile) ...
    cases.append(("stress: 10^5 position-sorted queries of 100-200 kbp (6-13 tiles each: 2.4e8 overlaps per step)",
                  synth.make_queries(100000, seed=7, genome=synth.HG38, min_len=100000, max_len=200000, sorted_=True), 0, 1, 10,
                  "long_sorted_q100000_v0"))
    dbs = [db] * len(cases)
    # ... a clustered database (half of the intervals around 2000 hot spots: tiles of 10^3 .. 10^4 records, many chunks each)
    # and a sparse one (10^5 intervals: 59 % of the tiles are empty; queries of up to 40 kbp reach up to three tiles on, so
    # rule NEST -- an empty first tile ends the query, src/igd_search.c:468 -- drops a third of what `-v 1`, rule FLAT, counts)
    others = []
    from igd_amd import Database
    for tag, kw, rows in (("cl300x40000", dict(files=300, per_file=40000, seed=77, genome=synth.HG38, clustered=True),
                           [("stress: clustered database (300 files x 40 000 intervals, half around 2000 hot spots: %d tile records) "
                             "+ 10^6 position-sorted queries", base, 0, 1, 30, "clustered_q1000000_v0")]),
                          ("clrm1900x26316", dict(files=1900, per_file=26316, seed=1000, genome=synth.HG38, clustered=True),
                           [("stress: the roadmap-scale database with real-data clustering (1900 files x 26 316 intervals, half of them around 2000 "
                             "hot spots: %d tile records, tiles of 10^2 .. 10^4 records) + the headline's 10^6 position-sorted queries", base, 0, 1, 30,
                             "clustered_roadmap_q1000000_v0"),
                            ("stress: the same clustered roadmap-scale database, -v 500 (%d tile records)", base, 500, 1, 30, "clustered_roadmap_q1000000_v500")]),
                          ("sparse100x1000", dict(files=100, per_file=1000, seed=31, genome=synth.HG38),
                           [("stress: sparse database (100 files x 1000 intervals: %d tile records, most tiles empty) + 10^6 position-sorted "
                             "queries of 100-40 000 bp, rule NEST (quirk #1 at scale)", None, 0, 1, 30, "sparse_q1000000_v0"),
                            ("stress: the same sparse database and queries under `-v 1` (rule FLAT: later tiles count behind an empty "
                             "first tile; %d tile records)", None, 1, 1, 30, "sparse_q1000000_v1")])):
        if tag == "clrm1900x26316" and not (args.files == 1900 and args.per_file == 26316):
            continue                                        # (a row of the full-size run only)
        try:
            op = os.path.join(args.dir, tag + ".igd")
            if not os.path.exists(op + ".done"):
                synth.make_db(op, **kw)
                open(op + ".done", "w").write("ok")
            odb = Database(op, device=dev.index or 0)
            others.append(odb)
            for (nm, qq, v_, gf, st_, gk) in rows:
                if qq is None:
                    qq = synth.make_queries(Q, seed=7, genome=synth.HG38, min_len=100, max_len=40000, sorted_=True)
                cases.append((nm % odb.nrecords, qq, v_, gf, st_, gk))
                dbs.append(odb)
        except Exception as e:
            out.append({"workload": "stress: database " + tag, "error": str(e)})
    for (name, (ichr, qs, qe), v, gflags, steps, gkey), db in zip(cases, dbs):
        # "row=fixture": the row's short name and the oracle fixture its counts are compared with (a shuffled batch has the
        # counts of the same queries position-sorted)
        rkey, _, gkey = gkey.partition("=")
        gkey = gkey or rkey
        try:
            job = Job(db, dev, stream, ichr, qs, qe, v, gflags, args.query_layout)
            el, prof = job.run(steps, 3)
            rl = job.roofline(prof)
            hj = job.d_hits.cpu().numpy()
            assert (hj % steps == 0).all(), "hits[] is not K times one batch"
            ent = {"key": rkey, "workload": name, "flags": gflags, "value": len(qs) * steps / el, "unit": "query-intervals/s", "steps": steps,
                   "ms_per_step": 1e3 * el / steps, "kernel": rl["kernel"], "kernel_ms": prof["scan_ms"], "pipeline_ms": prof["pipeline_ms"],
                   "roofline_frac": rl["frac"],
                   "bytes_per_launch": rl["bytes_per_launch"], "algorithmic_frac": rl["algorithmic_frac"],
                   "hits_per_step": int(hj.sum()) // steps, "hits_checksum": hits_checksum(hj // steps)}
            g = golden_counts(gkey) if (gkey and args.files == 1900 and args.per_file == 26316) else None
            if g:                                           # the oracle's counts of the same workload (committed fixture)
                ent["matches_oracle"] = (ent["hits_per_step"], ent["hits_checksum"]) == g
                if not ent["matches_oracle"]:
                    ent["error"] = "per-file counts differ from the oracle's (tests/golden/bench_checksums.json[%s])" % gkey
            elif gkey:
                ent["matches_oracle"] = None                # (another database size than the fixture's: nothing to compare with)
            out.append(ent)
            del job
        except Exception as e:                              # a side measurement must not lose the line
            out.append({"key": rkey, "workload": name, "error": str(e)})
    db = dbs[0]
    for odb in others:
        odb.close()
    # config 5: -f through the C API (count + scan + chunked fill + pinned D2H, result in host memory): the packed stream (8 bytes
    # per overlap, round 6 -- what the command line tool moves) and the 16-byte igd_hip_hit stream beside it
    try:
        ichr, qs, qe = base
        d2h = box.get("d2h_GBps") or 0.0
        for key, fn, per, what in (("config5_f_q1000000", db.enumerate_stream8, 8, "igd_hip_enumerate_stream8: 8 bytes per overlap (start | length | idx)"),
                                   ("config5_f_hit16_q1000000", db.enumerate_stream, 16, "igd_hip_enumerate_stream: 16-byte igd_hip_hit records")):
            if per == 8 and db.hit8_idx_bits() < 0:
                continue
            fn(ichr, qs, qe)                                # warm: workspace + pinned buffers
            best, tot = None, 0
            for _ in range(3):
                t = time.perf_counter()
                _, tot = fn(ichr, qs, qe)
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            nbytes = per * tot
            ent = {"key": key, "workload": "config 5: -f enumeration of 10^6 position-sorted queries, overlaps streamed to pinned host memory "
                                           "(%s, H2D of the queries included)" % what,
                   "value": Q / best, "unit": "query-intervals/s", "ms_per_call": 1e3 * best, "overlaps": int(tot),
                   "records_per_s": tot / best, "output_bytes": int(nbytes),
                   "roofline": {"bound": "pcie-d2h", "achieved": nbytes / best / 1e9, "peak": d2h, "unit": "GB/s",
                                "frac": (nbytes / best / 1e9 / d2h) if d2h else None,
                                "peak_source": "pinned device->host hipMemcpyAsync of 256 MiB measured in this run"}}
            g = golden_counts("config2_sorted_q1000000_v0") if (args.files == 1900 and args.per_file == 26316) else None
            ent["matches_oracle"] = (int(tot) == g[0]) if g else None      # (the number of overlaps = the counted total; the records: tests/test_gpu_fullsize.py)
            out.append(ent)
    except Exception as e:
        out.append({"key": "config5_f_q1000000", "workload": "config 5: -f", "error": str(e)})
    return out


def scale_anchor(extras):
    """The weak-scaling curve of BASELINE config 4 as ONE GPU can measure it: rank r of an N-GPU job runs slab r of one
    position-sorted set of N x 1.25e7 queries, so its step is what `--slab-of N` times here; N = 1 is the unsharded 1.25e7-query
    batch.  (The default N = 1 line is BASELINE config 2 -- 10^6 queries per step -- and no anchor for that curve.)"""
    def find(word):
        for e in extras:
            if word in e.get("workload", "") and e.get("ms_per_step"):
                return e
        return None
    pts = {1: find("config 4 per-GPU share"), 2: find("one of 2 GPUs"), 4: find("one of 4 GPUs"), 8: find("one of 8 GPUs")}
    if not pts[1]:
        return {"error": "the config-4 share was not measured"}
    out = {"per_gpu_queries": CONFIG4_PER_GPU,
           "what": "ms per step of ONE GPU on its share of an N-GPU config-4 job (slab 0 of N x 1.25e7 position-sorted queries); "
                   "the all-reduce of int64[nFiles] (15 KB, once per job) is not in these single-GPU figures",
           "step_ms": {str(n): e["ms_per_step"] for n, e in pts.items() if e},
           "predicted_value": {str(n): n * CONFIG4_PER_GPU / (e["ms_per_step"] * 1e-3) for n, e in pts.items() if e},
           "predicted_efficiency_vs_n1": {str(n): pts[1]["ms_per_step"] / e["ms_per_step"] for n, e in pts.items() if e},
           "note": "efficiency above 1 is expected: a slab of a larger sorted set touches fewer tiles (N = 8: one tile in eight, "
                   "530 queries per tile), so a rank's step gets SHORTER as N grows"}
    return out


LINE_CAP = 8192      # the driver keeps the last ~8 KB of stdout: the final line must fit whole


def _r(x, sig=6):
    """floats to `sig` significant digits (the compact line's numbers; the side file keeps them whole)"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    return x


def host_route_limit():
    """igdc_host_limit() of the product library: query files of at most this many lines are counted by the host's threads"""
    try:
        import ctypes
        lib = ctypes.CDLL(os.path.join(ROOT, "igd_amd", "lib", "libigd.so"))
        lib.igdc_host_limit.restype = ctypes.c_int64
        return int(lib.igdc_host_limit())
    except Exception:
        return None


def host_route_limit_enum():
    try:
        import ctypes
        lib = ctypes.CDLL(os.path.join(ROOT, "igd_amd", "lib", "libigd.so"))
        lib.igdc_host_limit_enum.restype = ctypes.c_int64
        return int(lib.igdc_host_limit_enum())
    except Exception:
        return 0


def compact_line(full, extra_path):
    """The ONE stdout line: the contract keys, the roofline and cpu_baseline objects and a few numbers per side measurement --
    everything else (phase tables, per-P sweeps, byte breakdowns, long workload descriptions) lives in the side file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    out = {k: (full[k] if k in ("value", "ms_per_step") else _r(full[k])) for k in keep}    # (value = queries / time exactly)
    cfg = full["config"]
    out["config"] = {k: cfg[k] for k in ("workload", "queries_per_gpu", "queries_per_step_all_gpus", "nfiles", "parallelism",
                                         "grouping", "collective") if k in cfg}
    rl = full["roofline"]
    o = {k: _r(rl.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_launch", "traffic", "kernel_ms",
                                    "pipeline_ms", "launches_timed", "frac_pmc", "step_frac", "algorithmic_bytes_per_launch",
                                    "algorithmic_frac")}
    o["bytes_source"] = "compulsory bytes of the launch, computed in this run (igd_hip_batch_traffic)"
    if rl.get("traffic_source"):
        o["traffic_source"] = rl["traffic_source"][:90]
    for c in ("cold", "cold_after_writes"):
        if isinstance(rl.get(c), dict) and rl[c].get("kernel_ms"):
            o[c] = {"kernel_ms": _r(rl[c]["kernel_ms"]), "frac": _r(rl[c].get("frac"))}
    if isinstance(rl.get("box"), dict):
        o["box"] = {k: _r(v, 4) for k, v in rl["box"].items() if isinstance(v, (int, float))}
    out["roofline"] = o
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind", "seconds", "totals_match_gpu")}
        c["sample"] = (cb.get("sample") or "")[:200]
        c["cpu_model"] = (cb.get("host") or {}).get("cpu_model")
        ac = cb.get("all_cores")
        if isinstance(ac, dict) and ac.get("value"):
            c["all_cores"] = {"value": _r(ac["value"]), "cores": ac.get("cores"), "seconds": _r(ac.get("seconds")),
                              "totals_match_gpu": ac.get("totals_match_gpu")}
        al = cb.get("all_cores_large")
        if isinstance(al, dict) and al.get("value"):
            c["all_cores_large"] = {"value": _r(al["value"]), "cores": al.get("cores"), "queries": al.get("queries"),
                                    "seconds": _r(al.get("seconds")), "totals_match_oracle_fixture": al.get("totals_match_oracle_fixture")}
        out["cpu_baseline"] = c
    for k in ("matches_oracle", "n_ranks_seen", "dist_backend", "hits_per_step_total", "hits_checksum", "efficiency_vs_anchor"):
        if k in full:
            out[k] = _r(full[k])
    out["db_open_s"] = _r(full.get("db_open_s"), 3)
    if "devices" in full:
        out["devices"] = [d[:60] for d in full["devices"]][:8]
    if "extra_configs" in full:                 # one short row per side measurement: key, step, dominant kernel, fraction, parity
        rows = []
        for e in full["extra_configs"]:
            row = {"workload": e.get("key") or e.get("workload", "")[:40]}
            if "error" in e:
                row["error"] = str(e["error"])[:80]
            else:
                row["ms_per_step"] = _r(e.get("ms_per_step", e.get("ms_per_call")), 4)
                if e.get("kernel_ms") is not None:
                    row["kernel"] = (e.get("kernel") or "").replace("igd_scan_", "")
                    row["kernel_ms"] = _r(e["kernel_ms"], 4)
                fr = e.get("roofline_frac", (e.get("roofline") or {}).get("frac"))
                row["frac"] = _r(fr, 3)
                if "matches_oracle" in e:
                    row["matches_oracle"] = e["matches_oracle"]
            rows.append(row)
        out["extra_configs"] = rows
    sa = full.get("scale_anchor")
    if isinstance(sa, dict):
        if "step_ms" in sa:
            out["scale_anchor"] = {"step_ms": {k: _r(v, 4) for k, v in sa["step_ms"].items()}}
        elif "ms_per_step" in sa:
            out["scale_anchor"] = {"ms_per_step": _r(sa["ms_per_step"], 4), "value": _r(sa.get("value"))}
        else:
            out["scale_anchor"] = {"error": str(sa.get("error"))[:80]}
    sp = full.get("scale_prediction")
    if isinstance(sp, dict) and "step_ms" in sp:
        out["scale_prediction"] = {"step_ms": sp["step_ms"], "source": str(sp.get("source"))[:120], "this_run_vs_predicted_step": _r(sp.get("this_run_vs_predicted_step"), 4)}
    ce = full.get("cli_end_to_end")
    if isinstance(ce, dict) and "error" not in ce:
        out["cli_end_to_end"] = {"what": "wall seconds of `bin/igd search <db> -q <the workload's BED>`, best of N; route = who counts: "
                                         "engine (MI355X) or host (CPU threads, files below igdc_host_limit)",
                                 "default_route": ce.get("default_route"), "default_seconds": _r(ce.get("q_seconds"), 4),
                                 "engine_seconds": _r(ce.get("q_engine_only_seconds"), 4),
                                 "engine_seconds_median": _r(ce.get("q_engine_only_seconds_median"), 4),
                                 "engine_seconds_slowest": _r(ce.get("q_engine_only_seconds_slowest"), 4),
                                 "engine_v500_seconds": _r(ce.get("q_v500_engine_seconds"), 4),
                                 "engine_f_seconds": _r(ce.get("q_f_engine_seconds"), 4),
                                 "reference_seconds": _r(ce.get("reference_q_seconds"), 4),
                                 "total_matches_gpu": ce.get("q_total_matches_gpu")}
    out["extra_file"] = extra_path
    text = json.dumps(out, separators=(",", ":"))
    # never lose the line to its size: drop the optional parts, largest first, until it fits
    for k in ("devices", "cli_end_to_end", "scale_prediction", "extra_configs", "scale_anchor"):
        if len(text) <= LINE_CAP - 200:
            break
        out.pop(k, None)
        out["dropped_for_size"] = out.get("dropped_for_size", []) + [k]
        text = json.dumps(out, separators=(",", ":"))
    return text


def write_extra(full, path):
    """The whole record (every side measurement in full) as a file beside bench.py; returns the path it was written to."""
    for cand in (path, os.path.join("/tmp", os.path.basename(path))):
        try:
            os.makedirs(os.path.dirname(cand) or ".", exist_ok=True)
            with open(cand, "w") as fh:
                json.dump(full, fh, indent=1)
                fh.write("\n")
            return cand
        except OSError:
            continue
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist
    from igd_amd import Database, synth
    from igd_amd.database import measure_rates
    from igd_amd.dist import allreduce_hits, gather_strings, init_from_env, ranks_seen, shard_bounds

    if os.environ.get("IGD_BENCH_DIE_RANK") == os.environ.get("RANK", "0") and "WORLD_SIZE" in os.environ:
        raise SystemExit("bench.py: rank %s told to die at start-up (IGD_BENCH_DIE_RANK: the launcher's fail-fast test)" % os.environ["RANK"])
    rank, world, local = init_from_env()
    if world != args.gpus:
        log("[bench] note: WORLD_SIZE=%d, --gpus=%d (using WORLD_SIZE)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: igd_amd has no CPU search path")
    if os.environ.get("IGD_BENCH_ONE_GPU"):      # N>1 control-flow test on a 1-GPU box (with IGD_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    grouped = dist.is_initialized()                 # N > 1, or one rank forced through the collective path (IGD_DIST_FORCE=1)

    def barrier():
        if grouped:
            dist.barrier()

    igd_path = os.path.join(args.dir, "rm%dx%d.igd" % (args.files, args.per_file))
    ensure_db(igd_path, args.files, args.per_file, rank, barrier)

    t = time.time()
    db = Database(igd_path, device=local)
    open_s = time.time() - t
    if world == 1 and args.slab_of > 1:
        Q = args.queries or CONFIG4_PER_GPU
        ichr, qs, qe = synth.make_queries_slab(args.slab_of * Q, 0, Q, seed=7, genome=synth.HG38)
        args.shuffled = False
        wl = ("slab 0 (%d queries) of ONE position-sorted set of %d x %d queries (seed 7): one GPU's share of a %d-GPU config-4 job"
              % (Q, args.slab_of, Q, args.slab_of))
    elif world == 1:
        Q = args.queries or 1000000
        ichr, qs, qe = synth.make_queries(Q, seed=7, genome=synth.HG38, sorted_=not args.shuffled)
        wl = ("%d %s queries per step (seed 7)" % (Q, "generation-order" if args.shuffled else "position-sorted"))
    else:
        # config 4: ONE position-sorted set, contiguous slabs
        Q = args.queries or CONFIG4_PER_GPU
        lo, hi = shard_bounds(world * Q, world, rank)
        ichr, qs, qe = synth.make_queries_slab(world * Q, lo, hi, seed=7, genome=synth.HG38)
        args.shuffled = False
        wl = ("ONE position-sorted set of %d x %d = %d queries (seed 7), rank r owns the r-th contiguous slab of %d "
              "(one batch per step)" % (world, Q, world * Q, Q))
    # One explicit (non-default) stream carries everything: torch's memsets/adds, the engine's
    # kernels (it enqueues on the hipStream_t it is given) and the RCCL all-reduce.
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    if args.grouping == "default":
        args.grouping = "bucket" if args.shuffled else "sorted"    # (igdc_queries_flags: what the CLI passes for an ordered / an unordered file)
    gflags = {"auto": 0, "sorted": 1, "bucket": 2}[args.grouping] | (4 if args.exact_arrays else 0)
    if args.grouping == "sorted" and not args.long_queries:
        gflags |= 16                                # IGD_HIP_FLAG_SHORT: the generator's queries are 100 .. 1999 bp (verified on the device like the order)

    # A job = K batches (steps) whose per-file counts ACCUMULATE in hits[] -- the reference's hits[] is
    # one accumulator over the whole query file (src/igd_search.c:925,1032-1039) and the engine adds --
    # followed by the path's ONE exchange: a SUM all-reduce of hits[nFiles] (no-op at N=1).  Both are
    # inside the timed region.
    job = Job(db, dev, stream, ichr, qs, qe, args.v, gflags, args.query_layout)
    elapsed, prof = job.run(args.steps, args.warmup, barrier, allreduce_hits)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if grouped:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # who took part: world size as the process group saw it, and each rank's device (ordinal, PCI bus id, name)
    try:
        props = torch.cuda.get_device_properties(local)
        me = "rank %d: cuda:%d %s pci %s" % (rank, local, props.name, getattr(props, "pci_bus_id", "?"))
    except Exception as e:
        me = "rank %d: cuda:%d (%s)" % (rank, local, e)
    devices = gather_strings(me)
    mode = "v" if (args.v > 0 and db.gtype == 1) else "hits"
    hits_job = job.d_hits.cpu().numpy()      # K identical batches per rank, summed over ranks
    assert args.steps > 0 and (hits_job % args.steps == 0).all(), "hits[] is not K times one batch"
    hits_one = hits_job // args.steps        # one batch per rank, summed over ranks
    tkey = "%s/%s/q%d%s%s" % (mode, "shuffled" if args.shuffled else "sorted", Q, "/exact" if args.exact_arrays else "",
                              "/slab0of%d" % args.slab_of if (world == 1 and args.slab_of > 1) else "")
    rl = job.roofline(prof, tkey if world == 1 else None)     # outside the timed region

    anchor = None
    if rank == 0 and world > 1:
        try:                                                # the N = 1 anchor of the weak-scaling curve: the same per-GPU batch size, unsharded
            a_q = synth.make_queries_slab(Q, 0, Q, seed=7, genome=synth.HG38)
            a_job = Job(db, dev, stream, a_q[0], a_q[1], a_q[2], args.v, gflags, args.query_layout)
            a_el, _ = a_job.run(30, 3)
            anchor = {"workload": "ONE GPU, %d position-sorted queries per step (seed 7): what N = 1 of this weak-scaling job is" % Q,
                      "steps": 30, "ms_per_step": 1e3 * a_el / 30, "value": Q * 30 / a_el, "unit": "query-intervals/s",
                      "measured": "by rank 0 on its GPU after the job's timed region"}
            del a_job
        except Exception as e:
            anchor = {"error": str(e)}
    if rank == 0:
        box = measure_rates(local)
        # what `frac` is a fraction of, three ways: compulsory bytes over the kernel's time (frac), the bytes the PMC counters saw
        # over the same time (frac_pmc), and compulsory bytes over the whole step -- all three kernels and their launch gaps
        rl["frac_of"] = "compulsory bytes of the launch (igd_hip_batch_traffic) / kernel_ms / 8 TB/s"
        rl["frac_pmc"] = (rl["traffic"] / (rl["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if (rl.get("traffic") and rl["kernel_ms"] > 0) else None
        rl["step_frac"] = rl["bytes_per_launch"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS
        rl["box"] = dict(box, note="measured in this run by igd_hip_measure_rates: float4 copy kernel (read+write bytes), "
                                   "float4 read-only kernel, pinned D2H / H2D copies")
        value = world * Q * args.steps / elapsed
        line = {
            "metric": "query-intervals/sec (igd search -q, hits-only) on roadmap-scale synthetic .igd",
            "value": value, "unit": "query-intervals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: roadmap-scale synthetic IGD, %d files x %d intervals (%d tile records, "
                                   "%d tiles of 16384 bp, hg38 contigs) replicated per GPU + %s, %s, queries and DB resident in HBM"
                                   % ("BASELINE config 2" if world == 1 else "BASELINE config 4", args.files, args.per_file,
                                      db.nrecords, db.ntiles, wl,
                                      "-v %d signal filter" % args.v if mode == "v" else "hits-only"),
                       "queries_per_gpu": Q, "queries_per_step_all_gpus": world * Q, "nfiles": db.nfiles,
                       "parallelism": "query-sharded x%d" % world, "grouping": args.grouping,
                       "query_layout": ("contig runs + starts + ends (8 B/query; igd_hip_search_runs_dev)" if job.d_runs is not None
                                        else "contig numbers + starts + ends (12 B/query; igd_hip_search_dev)"),
                       "collective": ("ONE sum all-reduce of int64[%d] per job (after the %d batches), inside the timed region"
                                      % (db.nfiles, args.steps)) if grouped else "none"},
            "roofline": rl,
            "n_ranks_seen": ranks_seen(), "devices": devices,
            "dist_backend": (dist.get_backend() if dist.is_initialized() else None),
            "hits_per_step_total": int(hits_one.sum()),
            "hits_checksum": hits_checksum(hits_one),
            "db_open_s": open_s,
        }
        # the oracle's counts of the same workload (committed fixture): a fast kernel with other counts is not a result
        gkey = None
        if args.files == 1900 and args.per_file == 26316 and not args.queries:
            if world == 1 and args.slab_of == 8:
                gkey = "config4_slab0_of_8_v%d" % (args.v if mode == "v" else 0)
            elif world == 1 and args.slab_of <= 1:
                gkey = "config2_sorted_q1000000_v%d" % (args.v if mode == "v" else 0)
        elif args.files == 1900 and args.per_file == 26316 and world == 1 and args.queries == CONFIG4_PER_GPU and not args.shuffled and args.slab_of <= 1:
            gkey = "config4_share_q12500000_v%d" % (args.v if mode == "v" else 0)
        g = golden_counts(gkey) if gkey else None
        if g:
            line["matches_oracle"] = (line["hits_per_step_total"], line["hits_checksum"]) == g
            if not line["matches_oracle"]:
                raise SystemExit("bench.py: per-file counts differ from the oracle's (tests/golden/bench_checksums.json[%s]): %s vs %s"
                                 % (gkey, (line["hits_per_step_total"], line["hits_checksum"]), g))
        if world == 1 and not args.no_cold:
            try:                                            # the headline batch with the last-level cache emptied before every launch
                cold = job.run_cold()
                rl["cold"] = {"kernel_ms": cold["scan_ms"], "pipeline_ms": cold["pipeline_ms"], "launches_timed": cold["launches"],
                              "achieved": rl["bytes_per_launch"] / (cold["scan_ms"] * 1e-3) / 1e9 if cold["scan_ms"] > 0 else None,
                              "frac": rl["bytes_per_launch"] / (cold["scan_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if cold["scan_ms"] > 0 else None,
                              "how": "512 MiB of other memory read (not written: clean lines) on the same stream between two launches, outside "
                                     "the kernel's HIP-event pair: no launch finds the image in the 256 MiB Infinity Cache"}
                cd = job.run_cold(dirty=True)
                rl["cold_after_writes"] = {"kernel_ms": cd["scan_ms"], "pipeline_ms": cd["pipeline_ms"], "launches_timed": cd["launches"],
                                           "frac": rl["bytes_per_launch"] / (cd["scan_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if cd["scan_ms"] > 0 else None,
                                           "how": "as `cold`, but the 512 MiB are read AND written (what a memset would leave): the "
                                                  "cache is full of dirty lines and their write-back competes with the kernel's reads"}
            except Exception as e:
                rl["cold"] = {"error": str(e)}
        if world == 1 and not args.no_cpu:
            bed = os.path.join(args.dir, "q%d_%s.bed" % (Q, "shuf" if args.shuffled else "sorted"))
            if not os.path.exists(bed):
                synth.write_bed(bed, synth.HG38, ichr, qs, qe)
            extra = ["-v", str(args.v)] if mode == "v" else []
            line["cpu_baseline"] = cpu_baseline(igd_path, bed, Q, int(hits_one.sum()), extra=extra)
            try:
                line["cli_end_to_end"] = cli_end_to_end(igd_path, bed, int(hits_one.sum()))
                line["cli_end_to_end"]["reference_q_seconds"] = line["cpu_baseline"].get("seconds") if not extra else None
            except Exception as e:
                line["cli_end_to_end"] = {"error": str(e)}
            if args.files == 1900 and args.per_file == 26316 and not args.no_extra and line["cpu_baseline"].get("kind") == "reference":
                try:
                    g = golden_counts("config4_share_q12500000_v0")
                    line["cpu_baseline"]["all_cores_large"] = cpu_baseline_allcores_large(
                        line["cpu_baseline"]["exe"], igd_path, synth.HG38,
                        synth.make_queries_slab(CONFIG4_PER_GPU, 0, CONFIG4_PER_GPU, seed=7, genome=synth.HG38), g[0] if g else -1, args.dir)
                except Exception as e:
                    line["cpu_baseline"]["all_cores_large"] = {"error": str(e)}
        if world == 1 and not args.no_extra:
            line["extra_configs"] = extra_configs(db, dev, stream, args, box)
            line["scale_anchor"] = scale_anchor(line["extra_configs"])
        if world > 1 and anchor is not None:
            # like for like: N ranks x 1.25e7 queries against ONE GPU running 1.25e7 queries of the same generator (measured by rank 0
            # right after the job, outside its timed region) -- the N = 1 line of bench.py runs BASELINE config 2 (10^6 queries per
            # step), another workload, so a ratio of the two lines' `value` says nothing about scaling
            line["scale_anchor"] = anchor
            line["efficiency_vs_anchor"] = value / (world * anchor["value"]) if anchor.get("value") else None
            # what one GPU measured for a rank's step of an N-GPU job (slab 0 of N, committed with its source): the curve a
            # measured SCALE file can be checked against -- a rank's step gets SHORTER as N grows (fewer tiles per slab)
            try:
                line["scale_prediction"] = json.load(open(os.path.join(ROOT, "profiles", "scale_prediction.json")))
                pm = line["scale_prediction"]["step_ms"].get(str(world))
                if pm and Q == CONFIG4_PER_GPU:
                    line["scale_prediction"]["this_run_vs_predicted_step"] = (1e3 * elapsed / args.steps) / pm
            except Exception as e:
                line["scale_prediction"] = {"error": str(e)}
        extra_path = write_extra(line, args.extra_out or os.path.join(ROOT, "bench_extra.json"))
        log("[bench] the whole record (%d bytes as one JSON line) is in %s" % (len(json.dumps(line)), extra_path))
        text = compact_line(line, os.path.relpath(extra_path, ROOT) if extra_path and extra_path.startswith(ROOT) else extra_path)
        assert len(text) <= LINE_CAP and "\n" not in text, len(text)
        print(text, flush=True)
    del job
    db.close()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
