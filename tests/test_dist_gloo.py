"""Multi-GPU host logic on CPU: world_size 2 over gloo.  Each rank takes its contiguous slab of
the queries (igd_amd.dist.shard_bounds), produces its local hits vector (here by the oracle,
standing in for the device step) and the ONE collective of the path -- the SUM all-reduce of
hits[nFiles] (igd_amd.dist.allreduce_hits) -- must reproduce the unsharded vector on every rank."""
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN, Oracle


def _worker(rank, world, port, path, qfile, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from igd_amd.dist import allreduce_hits, init_from_env, shard_bounds
    r, w, _ = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    o = Oracle(path)
    ichr, qs, qe = o.read_queries(qfile)
    lo, hi = shard_bounds(len(qs), w, r)
    local, _ = o.search(ichr[lo:hi], qs[lo:hi], qe[lo:hi], 0)
    t = torch.from_numpy(local.copy())
    allreduce_hits(t)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_partition():
    from igd_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_allreduce_equals_unsharded(tmp_path):
    import torch.multiprocessing as mp
    path = os.path.join(GOLDEN, "smallrand", "db.igd")
    qfile = os.path.join(GOLDEN, "smallrand", "q.bed")
    port = 29500 + (os.getpid() % 400)
    mp.spawn(_worker, args=(2, port, path, qfile, str(tmp_path)), nprocs=2, join=True)
    o = Oracle(path)
    ichr, qs, qe = o.read_queries(qfile)
    want, _ = o.search(ichr, qs, qe, 0)
    for r in range(2):
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), "r%d.npy" % r)), want)


def test_bench_spawns_its_ranks_before_touching_the_gpu(tmp_path, monkeypatch):
    """`python bench.py --gpus 2` without WORLD_SIZE starts 2 child processes with the torchrun environment and
    never imports torch / igd_amd in the parent (checked by running spawn_ranks against a stub child)."""
    import importlib
    import bench
    importlib.reload(bench)
    stub = tmp_path / "stub.py"
    stub.write_text("import os, sys\n"
                    "open(os.path.join(%r, 'r' + os.environ['RANK']), 'w').write(' '.join([os.environ['WORLD_SIZE'], "
                    "os.environ['LOCAL_RANK'], os.environ['MASTER_ADDR'], os.environ['MASTER_PORT']] + sys.argv[1:]))\n"
                    "sys.exit(3 if os.environ['RANK'] == '1' else 0)\n" % str(tmp_path))
    monkeypatch.setattr(bench, "__file__", str(stub))
    mods = set(sys.modules)
    rc = bench.spawn_ranks(2, ["--gpus", "2", "--steps", "4"])
    assert rc == 3                                          # the worst child exit code is the parent's
    assert not ({"torch", "igd_amd._native"} & (set(sys.modules) - mods))
    got = [open(os.path.join(str(tmp_path), "r%d" % r)).read().split() for r in range(2)]
    assert got[0][0] == got[1][0] == "2" and [g[1] for g in got] == ["0", "1"]
    assert got[0][2] == "127.0.0.1" and got[0][3] == got[1][3] and got[0][4:] == ["--gpus", "2", "--steps", "4"]


def test_slab_generator_equals_slices_of_the_sorted_set():
    """config 4's per-rank slabs are slices of ONE sorted set (igd_synth_queries_slab == make_queries[lo:hi])"""
    from igd_amd import synth
    from igd_amd.dist import shard_bounds
    n = 300000
    whole = synth.make_queries(n, seed=7, genome=synth.HG38, sorted_=True)
    for world in (2, 3, 8):
        for r in range(world):
            lo, hi = shard_bounds(n, world, r)
            part = synth.make_queries_slab(n, lo, hi, seed=7, genome=synth.HG38)
            assert all(np.array_equal(a[lo:hi], b) for a, b in zip(whole, part)), (world, r)
