#!/usr/bin/env python3
"""Make the benchmark database and query BEDs under /tmp/igdb (fresh GPU boxes start empty)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igd_amd import synth
D = "/tmp/igdb"; os.makedirs(D, exist_ok=True)
p = os.path.join(D, "rm1900x26316.igd")
if not os.path.exists(p + ".done"):
    synth.make_db(p); open(p + ".done", "w").write("ok")
for name, srt in (("m_q.bed", True), ("m_qs.bed", False)):
    f = os.path.join(D, name)
    if not os.path.exists(f):
        synth.write_bed(f, synth.HG38, *synth.make_queries(1000000, seed=7, genome=synth.HG38, sorted_=srt))
print("prepared", D)
