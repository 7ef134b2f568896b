#!/usr/bin/env python3
"""tools/sp_stamps.py -- read gpurun_out/qb_stamps.bin of a -DIGD_EXP=0x1000000 build after a batch under IGD_HIP_FLAG_BUCKET
(k_query_bounds is not launched, the stamps are k_split_local's: start, counters cleared, wave 0 counted, all counted, table row +
cursors, tuples issued, stores drained) and print where a workgroup's time goes (ticks of the 100 MHz s_memtime clock)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/qb_stamps.bin", dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[a[:, 6] > 0]
a = a[a[:, 0] >= np.percentile(a[:, 0], 1)]
t0 = a[:, 0].min()
def q(x): return "min %7d  p10 %7d  p50 %7d  p90 %7d  max %7d" % (x.min(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max())
print("workgroups %d, span %d ticks" % (len(a), a[:, 6].max() - t0))
print("start (after the first)   ", q(a[:, 0] - t0))
names = ["counters cleared (barr.) ", "wave 0 counted its pairs ", "all waves (barrier)      ", "prefix, table row (barr.)", "tuples issued            ", "stores drained           "]
if len(sys.argv) > 2 and sys.argv[2] == "fine":      # the LAST kernel that stamped was k_split_fine (IGD_HIP_SPLIT_ONE=1): split_fine_whole's staged path
    names = ["table column, sums (barr)", "segments laid out (barr.)", "tuples fetched + counted ", "prefix over tiles (barr.)", "pairs issued             ", "stores drained           "]
for k in range(6): print(names[k], q(a[:, k + 1] - a[:, k]))
print("whole workgroup           ", q(a[:, 6] - a[:, 0]))
