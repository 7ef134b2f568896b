"""The driver's contract for bench.py: ONE JSON line on stdout with the agreed keys, a roofline object for
the dominant kernel and a cpu_baseline object -- checked here on a small synthetic workload so that the
contract cannot rot unnoticed (the real workload is BASELINE.json's configs[1])."""
import json
import os
import shutil
import subprocess
import sys

import pytest

from helpers import ROOT, short_tmpdir

pytestmark = pytest.mark.gpu
LINE_CAP = 8192          # bench.py's LINE_CAP: what the driver's stdout tail holds


def test_bench_prints_one_json_line_with_the_contract_keys():
    d = short_tmpdir("igb")
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--files", "40", "--per-file", "3000", "--queries", "20000",
                            "--steps", "3", "--warmup", "1", "--dir", d, "--extra-out", os.path.join(d, "bench_extra.json")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert p.returncode == 0, p.stderr.decode()[-800:]
        lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
        assert len(lines) == 1
        # the driver keeps the last ~8 KB of stdout: the line must fit whole (round 5's 20 KB line was never parsed)
        assert len(lines[0]) <= LINE_CAP, len(lines[0])
        c = json.loads(lines[0])
        assert "dropped_for_size" not in c
        # the compact line: the contract keys + roofline + cpu_baseline + one short row per side measurement
        for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
            assert isinstance(c[k], t), k
        assert c["n_gpus"] == 1 and c["steps"] == 3 and c["warmup"] == 1 and c["vs_baseline"] is None and "workload" in c["config"]
        assert abs(c["value"] - 20000 * 3 / (c["ms_per_step"] * 3e-3)) / c["value"] < 1e-4
        cr = c["roofline"]
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_launch", "traffic", "kernel_ms"):
            assert k in cr, k
        assert cr["bound"] == "hbm" and cr["peak"] == 8000.0 and 0 < cr["frac"] <= 1.0 and abs(cr["frac"] - cr["achieved"] / 8000.0) < 1e-4
        cc = c["cpu_baseline"]
        assert cc["kind"] in ("reference", "port") and cc["cores"] == 1 and cc["value"] > 0 and cc["seconds"] > 0 and cc["totals_match_gpu"] is True
        assert c["n_ranks_seen"] == 1 and "matches_oracle" not in c     # (not the fixture's database size)
        xr = c["extra_configs"]
        assert len(xr) == 17 and len(set(r["workload"] for r in xr)) == 17 and not any("error" in r for r in xr), xr
        assert all(r["ms_per_step"] > 0 and set(r) <= {"workload", "ms_per_step", "kernel", "kernel_ms", "frac", "matches_oracle"} for r in xr)
        ce = c["cli_end_to_end"]
        assert ce["default_route"] in ("host", "engine") and ce["engine_seconds"] > 0 and ce["default_seconds"] > 0
        assert set(c["scale_anchor"]["step_ms"]) == {"1", "2", "4", "8"}
        # ... and the whole record in the side file the line names
        assert c["extra_file"] and os.path.exists(os.path.join(d, "bench_extra.json"))
        j = json.load(open(os.path.join(d, "bench_extra.json")))
        assert abs(j["value"] - c["value"]) / j["value"] < 1e-5 and j["roofline"]["kernel"] == cr["kernel"]
        for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
            assert isinstance(j[k], t), k
        assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["vs_baseline"] is None
        assert j["higher_is_better"] is True and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
        assert abs(j["value"] - 20000 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-6
        r = j["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["achieved"] > 0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
        # the fraction prices the kernel time against bytes computed IN THE RUN, and is a fraction
        assert 0 < r["frac"] <= 1.0 and r["bytes_per_launch"] == r["bytes_breakdown"]["total"] > 0
        assert abs(r["achieved"] - r["bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
        assert r["algorithmic_bytes_per_launch"] > 0 and r["box"]["hbm_copy_GBps"] > 500 and r["box"]["d2h_GBps"] > 1
        c = j["cpu_baseline"]
        x = j["extra_configs"]
        assert len(x) == 17 and not any("error" in e for e in x), x
        assert [e["key"] for e in x] == [r["workload"] for r in xr]
        assert all(e["value"] > 0 for e in x) and all(e["roofline"]["bound"] == "pcie-d2h" and e["overlaps"] > 0 for e in x[-2:])
        assert x[-2]["key"] == "config5_f_q1000000" and x[-2]["output_bytes"] * 2 == x[-1]["output_bytes"]       # 8 against 16 bytes per overlap
        assert sum(e["workload"].startswith("stress:") for e in x) == 6
        # every row but `-f` carries the comparison with the oracle's fixture (None here: not the fixture's database size)
        assert all("matches_oracle" in e for e in x), [e["workload"] for e in x if "matches_oracle" not in e]
        assert r["cold"]["kernel_ms"] > 0 and 0 < r["cold"]["frac"] <= 1.0
        # what the fraction is a fraction of, and the like-for-like anchors of the weak-scaling curve
        assert r["frac_of"] and r["step_frac"] > 0 and r["step_frac"] <= r["frac"] and "frac_pmc" in r
        sa = j["scale_anchor"]
        assert set(sa["step_ms"]) == {"1", "2", "4", "8"} and all(v > 0 for v in sa["step_ms"].values()) and sa["predicted_value"]["8"] > 0
        assert "query_layout" in j["config"]
        e2e = j["cli_end_to_end"]
        small = e2e["small_files"]
        assert [row["queries"] for row in small] == [1000, 10000, 100000, 300000, 1000000, 3000000] and all(row["product_seconds"] > 0 for row in small)
        assert all(row["stdout_identical"] for row in small if "reference_seconds" in row)
        assert e2e["q_seconds"] > 0 and e2e["q_v500_seconds"] > 0 and e2e["q_f_seconds"] > 0 and e2e["q_total_matches_gpu"] is True
        # both routes at every size, each labelled with who counted
        assert e2e["default_route"] in ("host", "engine") and e2e["q_v500_engine_seconds"] > 0 and e2e["q_f_engine_seconds"] > 0
        assert all(row["product_route"] in ("host", "engine") and row["product_engine_only_seconds"] > 0 for row in small)
        # the engine route's own seconds with the tool's phase table of the fastest and the slowest repeat
        assert e2e["q_engine_only_seconds"] > 0 and e2e["q_engine_only_seconds_slowest"] >= e2e["q_engine_only_seconds"]
        assert any("HIP runtime init" in l for l in e2e["q_engine_only_phases_fastest"]) and e2e["q_engine_only_phases_slowest"]
        assert c["host"]["cpu_model"] and c["host"]["logical_cpus"] >= 1
        c = j["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == j["unit"] and c["sample"]
        assert c["totals_match_gpu"] is True
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _two_rank_expected(d, files, per_file, q):
    """hits of the whole 2 x q position-sorted set through Database.search on one GPU"""
    import numpy as np
    from igd_amd import Database, synth
    db = Database(os.path.join(d, "rm%dx%d.igd" % (files, per_file)))
    ichr, qs, qe = synth.make_queries(2 * q, seed=7, genome=synth.HG38, sorted_=True)
    hits, total = db.search(ichr, qs, qe)
    db.close()
    w = (hits.astype(np.uint64) * (np.arange(len(hits), dtype=np.uint64) + np.uint64(1))).sum() & np.uint64((1 << 63) - 1)
    return int(total), int(w)


@pytest.mark.parametrize("launcher", ["self-spawn", "torchrun"])
def test_bench_two_ranks_config4_slabs_and_allreduce(launcher):
    """N > 1: `bench.py --gpus 2` starts two ranks (by itself, or under torch.distributed.run as the driver does),
    each takes its contiguous slab of ONE sorted query set, and the all-reduced hits[] equals the unsharded search.
    Both ranks share GPU 0 here (IGD_BENCH_ONE_GPU) and the collective runs over gloo -- the box has one GPU."""
    d = short_tmpdir("igb")
    try:
        files, per_file, q = 40, 3000, 15000
        env = dict(os.environ, IGD_BENCH_ONE_GPU="1", IGD_DIST_BACKEND="gloo")
        env.pop("WORLD_SIZE", None)
        env.pop("RANK", None)
        tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--files", str(files), "--per-file", str(per_file),
                "--queries", str(q), "--steps", "3", "--warmup", "1", "--dir", d, "--extra-out", os.path.join(d, "bench_extra.json")]
        if launcher == "self-spawn":
            cmd = [sys.executable] + tail
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                   "--master-addr", "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300)] + tail
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        lines = [l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")]
        assert len(lines) == 1, p.stdout.decode()[-800:]
        assert len(lines[0]) <= LINE_CAP, len(lines[0])
        j = json.loads(lines[0])
        assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 3
        assert "config 4" in j["config"]["workload"] and "all-reduce" in j["config"]["collective"]
        assert j["config"]["queries_per_gpu"] == q and j["config"]["queries_per_step_all_gpus"] == 2 * q
        assert abs(j["value"] - 2 * q * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-6
        assert 0 < j["roofline"]["frac"] <= 1.0
        total, chk = _two_rank_expected(d, files, per_file, q)
        assert j["hits_per_step_total"] == total and j["hits_checksum"] == chk
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_rccl_world_size_one_int64_sum_on_the_engines_stream():
    """librccl loads next to libigd_hip.so in ONE process, the "nccl" backend initialises on device 0 and an int64 SUM
    all-reduce of a hits vector runs on the non-default stream the engine's kernels were enqueued on -- everything the
    N > 1 job needs from RCCL except a second GPU.  In a child process: the process group must not leak into pytest."""
    d = short_tmpdir("igb")
    try:
        code = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %r)
from igd_amd import Database, synth
from igd_amd.dist import allreduce_hits, init_from_env
path = os.path.join(%r, "s.igd")
synth.make_db(path, files=1900, per_file=60, seed=1000, genome=synth.SMALL)
rank, world, local = init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
db = Database(path, device=0)
assert db.nfiles == 1900
ichr, qs, qe = synth.make_queries(50000, seed=7, genome=synth.SMALL, sorted_=True)
want, wtot = db.search(ichr, qs, qe)
st = torch.cuda.Stream(device=dev)
torch.cuda.synchronize(dev)
with torch.cuda.stream(st):
    d_q = [torch.from_numpy(a).to(dev) for a in (ichr, qs, qe)]
    hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
    for _ in range(3):
        db.search_dev(d_q[0].data_ptr(), d_q[1].data_ptr(), d_q[2].data_ptr(), len(qs), hits.data_ptr(), None,
                      stream=st.cuda_stream, flags=1)
    allreduce_hits(hits)                       # RCCL: int64[1900] SUM on `st`
    st.synchronize()
db.sync(st.cuda_stream)
assert np.array_equal(hits.cpu().numpy(), 3 * want), "all-reduced hits differ"
maps = open("/proc/self/maps").read()
assert "librccl" in maps and "libigd_hip.so" in maps
dist.barrier(); dist.destroy_process_group(); db.close()
print("RCCL-OK", int(hits.sum().item()))
''' % (ROOT, d)
        env = dict(os.environ, IGD_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(29950 + os.getpid() % 40), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("IGD_DIST_BACKEND", None)
        p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
        assert p.returncode == 0 and b"RCCL-OK" in p.stdout, (p.stdout.decode()[-500:], p.stderr.decode()[-1500:])
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_bench_one_rank_forced_through_the_collective_path():
    """bench.py with WORLD_SIZE=1 and IGD_DIST_FORCE=1: process group over RCCL, barriers, the all-reduce of hits[] inside
    the timed region and the MAX of the elapsed times -- the N > 1 code path on one GPU -- and the line says so."""
    d = short_tmpdir("igb")
    try:
        env = dict(os.environ, IGD_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(29900 + os.getpid() % 40), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("IGD_DIST_BACKEND", None)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--files", "40", "--per-file", "3000", "--queries", "20000",
                            "--steps", "3", "--warmup", "1", "--dir", d, "--no-extra", "--no-cpu", "--extra-out", os.path.join(d, "bench_extra.json")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        j = json.loads([l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")][0])
        assert j["n_gpus"] == 1 and j["n_ranks_seen"] == 1 and j["dist_backend"] == "nccl"
        assert "all-reduce" in j["config"]["collective"] and len(j["devices"]) == 1 and "cuda:0" in j["devices"][0]
        assert j["hits_per_step_total"] > 0
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_bench_fails_fast_when_a_rank_dies():
    """`bench.py --gpus 2` where rank 1 dies at start-up (IGD_BENCH_DIE_RANK): the launcher stops rank 0 -- which would sit in
    the rendezvous / a barrier until the collective's timeout -- and exits non-zero within seconds, naming the rank."""
    import time
    d = short_tmpdir("igb")
    try:
        env = dict(os.environ, IGD_BENCH_ONE_GPU="1", IGD_DIST_BACKEND="gloo", IGD_BENCH_DIE_RANK="1")
        env.pop("WORLD_SIZE", None)
        env.pop("RANK", None)
        t = time.time()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--files", "40", "--per-file", "3000",
                            "--queries", "15000", "--steps", "3", "--warmup", "1", "--dir", d, "--extra-out", os.path.join(d, "bench_extra.json")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
        assert p.returncode != 0 and time.time() - t < 120
        assert b"rank 1 failed" in p.stderr, p.stderr.decode()[-800:]
        assert not [l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _expected_whole(d, files, per_file, n_total):
    """hits of the whole position-sorted set of n_total queries through Database.search on one GPU"""
    import numpy as np
    from igd_amd import Database, synth
    db = Database(os.path.join(d, "rm%dx%d.igd" % (files, per_file)))
    ichr, qs, qe = synth.make_queries(n_total, seed=7, genome=synth.HG38, sorted_=True)
    hits, total = db.search(ichr, qs, qe)
    db.close()
    w = (hits.astype(np.uint64) * (np.arange(len(hits), dtype=np.uint64) + np.uint64(1))).sum() & np.uint64((1 << 63) - 1)
    return int(total), int(w)


def test_bench_eight_ranks_control_flow_on_one_gpu():
    """Ready for the 8-GPU node: `bench.py --gpus 8` exactly as the driver starts it (torch.distributed.run,
    8 ranks) -- slab bounds of ONE sorted set, barriers, the all-reduce, the MAX of the elapsed times, rank 0's anchor run --
    with the 8 ranks sharing GPU 0 and the collective over gloo (the box has one GPU; with 8 GPUs the same code takes
    LOCAL_RANK and RCCL).  The all-reduced hits[] equals the unsharded search, the line names 8 ranks, carries the
    like-for-like anchor, its efficiency and the predicted per-rank step, and the whole job ends in minutes."""
    import time
    d = short_tmpdir("igb")
    try:
        files, per_file, q = 40, 3000, 6000
        env = dict(os.environ, IGD_BENCH_ONE_GPU="1", IGD_DIST_BACKEND="gloo")
        env.pop("WORLD_SIZE", None)
        env.pop("RANK", None)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
               "--master-addr", "127.0.0.1", "--master-port", str(29300 + os.getpid() % 250),
               os.path.join(ROOT, "bench.py"), "--gpus", "8", "--files", str(files), "--per-file", str(per_file),
               "--queries", str(q), "--steps", "3", "--warmup", "1", "--dir", d, "--extra-out", os.path.join(d, "bench_extra.json")]
        t = time.time()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        wall = time.time() - t
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        lines = [l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")]
        assert len(lines) == 1, p.stdout.decode()[-800:]
        assert len(lines[0]) <= LINE_CAP, len(lines[0])
        j = json.loads(lines[0])
        assert "dropped_for_size" not in j and j["roofline"]["kernel_ms"] > 0
        assert j["n_gpus"] == 8 and j["n_ranks_seen"] == 8 and len(j["devices"]) == 8 and j["scaling"] == "weak"
        assert all(("rank %d:" % r) in j["devices"][r] for r in range(8))
        assert j["config"]["queries_per_gpu"] == q and j["config"]["queries_per_step_all_gpus"] == 8 * q
        assert "config 4" in j["config"]["workload"] and "all-reduce" in j["config"]["collective"]
        assert abs(j["value"] - 8 * q * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-6
        assert j["scale_anchor"]["value"] > 0 and j["efficiency_vs_anchor"] > 0
        pr = j["scale_prediction"]                              # what a future SCALE file can be checked against
        assert set(pr["step_ms"]) == {"1", "2", "4", "8"} and pr["source"]
        total, chk = _expected_whole(d, files, per_file, 8 * q)
        assert j["hits_per_step_total"] == total and j["hits_checksum"] == chk
        assert wall < 600, wall
    finally:
        shutil.rmtree(d, ignore_errors=True)
