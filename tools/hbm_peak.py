#!/usr/bin/env python3
"""Measured HBM rates of the box (quote the measured figure next to the 8 TB/s spec):
device-to-device copy (read + write bytes) and a read-only reduction, 2 GiB buffers, best of 10."""
import json, time, torch
dev = torch.device("cuda", 0)
n = 1 << 29                                   # 2 GiB of float32
a = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0)
b = torch.empty_like(a)
def best(f, reps=10):
    f(); torch.cuda.synchronize()
    t = 1e9
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        t = min(t, s.elapsed_time(e) * 1e-3)
    return t
tc = best(lambda: b.copy_(a))
tr = best(lambda: a.sum())
out = {"copy_GBps_read_plus_write": 2 * a.numel() * 4 / tc / 1e9, "read_only_sum_GBps": a.numel() * 4 / tr / 1e9,
       "buffer_GiB": a.numel() * 4 / 2**30, "spec_GBps": 8000}
print(json.dumps(out, indent=1))
