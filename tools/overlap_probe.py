#!/usr/bin/env python3
"""tools/overlap_probe.py -- GPU box: does a second engine handle on a second stream (k_query_bounds of one batch running
beside the scan kernel of another) raise the throughput of a stream of batches?  Per workload: one handle, K steps; two
handles on two streams, K/2 steps each, interleaved."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igd_amd import Database, synth
import bench
dev = torch.device("cuda", 0)
PATH = "/tmp/igdb/rm1900x26316.igd"
if not os.path.exists(PATH + ".done"):
    os.makedirs(os.path.dirname(PATH), exist_ok=True)
    synth.make_db(PATH, files=1900, per_file=26316, seed=1000, nbp_log=14, genome=synth.HG38)
    open(PATH + ".done", "w").write("ok")
dbs = [Database(PATH), Database(PATH)]
sts = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
P = bench.CONFIG4_PER_GPU
work = [("config 2: 10^6 sorted", synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=True), 200),
        ("config 4 share (dense)", synth.make_queries_slab(P, 0, P, seed=7, genome=synth.HG38), 40),
        ("config 4 slab 0 of 8", synth.make_queries_slab(8 * P, 0, P, seed=7, genome=synth.HG38), 40)]
for name, q, K in work:
    jobs = [bench.Job(dbs[i], dev, sts[i].cuda_stream, *q, 0, 1) for i in range(2)]
    for j in jobs:
        for _ in range(3): j.step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(K): jobs[0].step()
    torch.cuda.synchronize(dev)
    one = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K // 2):
        jobs[0].step(); jobs[1].step()
    torch.cuda.synchronize(dev)
    two = (time.perf_counter() - t0) / K
    for i in range(2): dbs[i].sync(sts[i].cuda_stream)
    h0, h1 = jobs[0].d_hits.sum().item(), jobs[1].d_hits.sum().item()
    print("%-24s | one handle %7.1f us/step | two handles, two streams %7.1f us/step | hits %d %d" % (name, 1e6 * one, 1e6 * two, h0, h1), flush=True)
    del jobs
