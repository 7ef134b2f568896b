#!/usr/bin/env python3
"""tools/tail_stamps.py -- read gpurun_out/tail_stamps.bin of a -DIGD_EXP=0x400000 build (5 s_memtime stamps per wave of the
batch's last launch: start, LDS counters cleared, exact walks done, coverage sums done, counters flushed) and print where the
waves' time goes (ticks of the 100 MHz counter)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tail_stamps.bin", dtype=np.uint64).reshape(-1, 5).astype(np.int64)
a = a[a[:, 4] > 0]
a -= a[:, 0].min()
def q(x): return "min %7d  p10 %7d  p50 %7d  p90 %7d  max %7d" % (x.min(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max())
print("waves %d, span %d ticks" % (len(a), a[:, 4].max()))
print("start (after first wave) ", q(a[:, 0]))
print("counters cleared - start ", q(a[:, 1] - a[:, 0]))
print("walks                    ", q(a[:, 2] - a[:, 1]))
print("coverage sums            ", q(a[:, 3] - a[:, 2]))
print("flush                    ", q(a[:, 4] - a[:, 3]))
print("end (absolute)           ", q(a[:, 4]))
