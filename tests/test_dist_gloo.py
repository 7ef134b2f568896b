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
