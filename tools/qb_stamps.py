#!/usr/bin/env python3
"""tools/qb_stamps.py -- read gpurun_out/qb_stamps.bin of a -DIGD_EXP=0x1000000 build (s_memtime stamps of wave 0 of every
k_query_bounds workgroup of the last launch: start, tables staged, queries arrived, keys and words done, later[] compacted,
firstQ filled + qw0 stored, stores drained) and print where a workgroup's time goes (ticks of the shader clock)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/qb_stamps.bin", dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[a[:, 6] > 0]
a = a[a[:, 0] >= np.percentile(a[:, 0], 1)]          # (stale rows of an earlier, larger launch)
t0 = a[:, 0].min()
def q(x): return "min %7d  p10 %7d  p50 %7d  p90 %7d  max %7d" % (x.min(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max())
print("workgroups %d, span %d ticks" % (len(a), a[:, 6].max() - t0))
print("start (after the first)   ", q(a[:, 0] - t0))
names = ["tables staged (barrier)  ", "queries arrived          ", "keys and words           ", "later[] compacted (barr.)", "firstQ fill + qw0 store  ", "edges + stores drained   "]
for k in range(6): print(names[k], q(a[:, k + 1] - a[:, k]))
print("whole workgroup           ", q(a[:, 6] - a[:, 0]))
