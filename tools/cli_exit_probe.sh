#!/bin/bash
# tools/cli_exit_probe.sh -- GPU box: wall time of `bin/igd search -q` with and without the HIP runtime's exit handlers (IGD_CLEAN_EXIT)
python tools/prep.py > /dev/null 2>&1
python3 - <<'PY'
import subprocess, time, os
for mode in ("1", "0"):
    ts = []
    for _ in range(7):
        t = time.perf_counter()
        subprocess.run(["bin/igd", "search", "/tmp/igdb/rm1900x26316.igd", "-q", "/tmp/igdb/m_q.bed"], stdout=open("/tmp/o_%s.txt" % mode, "wb"), env=dict(os.environ, IGD_CLEAN_EXIT=mode))
        ts.append(time.perf_counter() - t)
    print("IGD_CLEAN_EXIT=%s: best %.3f s, median %.3f s" % (mode, min(ts), sorted(ts)[3]))
print("same output:", open("/tmp/o_0.txt", "rb").read() == open("/tmp/o_1.txt", "rb").read())
PY
