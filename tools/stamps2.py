#!/usr/bin/env python3
"""tools/stamps2.py -- per-wave durations from gpurun_out/stamps.bin of a -DIGD_EXP=32 build (the XCDs' clocks are not
synchronised: only differences inside a wave mean anything): descriptor phase, unit loop, row flush."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps.bin", dtype=np.uint64).reshape(-1, 5).astype(np.int64)
a = a[a[:, 4] > 0]
def q(x): return "min %7d  p10 %7d  p50 %7d  p90 %7d  p99 %7d  max %7d" % (x.min(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), np.percentile(x, 99), x.max())
print("waves", len(a))
print("descriptors ", q(a[:, 1] - a[:, 0]))
print("unit loop   ", q(a[:, 3] - a[:, 1]))
print("flush       ", q(a[:, 4] - a[:, 3]))
print("whole wave  ", q(a[:, 4] - a[:, 0]))
