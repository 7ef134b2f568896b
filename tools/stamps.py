#!/usr/bin/env python3
"""tools/stamps.py -- read gpurun_out/stamps.bin of a -DIGD_EXP=32 build (5 s_memtime stamps per wave of the last
igd_scan_sorted launch: start, descriptors loaded, first unit's records landed, unit loop done, end) and print where
the waves' time goes (shader clock ticks; 100 MHz realtime not used: ratios are what matter)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps.bin", dtype=np.uint64).reshape(-1, 5).astype(np.int64)
a = a[a[:, 4] > 0]
t0 = a[:, 0].min()
a -= t0
tot = a[:, 4].max()
def q(x): return "min %8d  p50 %8d  p99 %8d  max %8d" % (x.min(), np.percentile(x, 50), np.percentile(x, 99), x.max())
print("waves %d, kernel span %d ticks" % (len(a), tot))
print("start (after first wave)   ", q(a[:, 0]))
print("descriptors done - start   ", q(a[:, 1] - a[:, 0]))
print("first records - desc done  ", q(a[:, 2] - a[:, 1]))
print("unit loop done - first recs", q(a[:, 3] - a[:, 2]))
print("end - loop done            ", q(a[:, 4] - a[:, 3]))
print("end (absolute)             ", q(a[:, 4]))
print("loop done (absolute)       ", q(a[:, 3]))
