#!/usr/bin/env python3
"""Unordered batch whose queries all fall into a handful of tiles (worst case for any per-tile grouping):
times igd_hip_search_dev in device-decides mode and checks the counts against the sorted-promise path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from igd_amd import Database

db = Database("/tmp/igdb/rm1900x26316.igd", device=0)
rng = np.random.default_rng(5)
for span_tiles in (1, 10, 1000):
    Q = 1000000
    qs = (50_000_000 + rng.integers(0, 16384 * span_tiles, Q)).astype(np.int32)
    qe = (qs + rng.integers(100, 2000, Q)).astype(np.int32)
    ichr = np.zeros(Q, np.int32)
    order = np.argsort(qs, kind="stable")
    want, wtot = db.search(ichr[order], qs[order], qe[order], flags=1)
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(a).to(dev) for a in (ichr, qs, qe)]
    hits = torch.zeros(db.nfiles, dtype=torch.int64, device=dev)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    for _ in range(3):
        db.search_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), Q, hits.data_ptr(), None, stream=st.cuda_stream, flags=8)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        db.search_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), Q, hits.data_ptr(), None, stream=st.cuda_stream, flags=8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    ok = np.array_equal(hits.cpu().numpy(), want)
    print("queries inside %4d tiles: %.3f ms per batch, %.3g q/s, counts %s (total %d)" % (span_tiles, dt * 1e3, Q / dt, "ok" if ok else "WRONG", wtot))
db.close()
