"""The driver's contract for bench.py: ONE JSON line on stdout with the agreed keys, a roofline object for
the dominant kernel and a cpu_baseline object -- checked here on a small synthetic workload so that the
contract cannot rot unnoticed (the real workload is BASELINE.json's configs[1])."""
import json
import os
import shutil
import subprocess
import sys

import pytest

from helpers import ROOT, short_tmpdir

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_keys():
    d = short_tmpdir("igb")
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--files", "40", "--per-file", "3000", "--queries", "20000",
                            "--steps", "3", "--warmup", "1", "--dir", d], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert p.returncode == 0, p.stderr.decode()[-800:]
        lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
        assert len(lines) == 1
        j = json.loads(lines[0])
        for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
            assert isinstance(j[k], t), k
        assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["vs_baseline"] is None
        assert j["higher_is_better"] is True and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
        assert abs(j["value"] - 20000 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-6
        r = j["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["achieved"] > 0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
        c = j["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == j["unit"] and c["sample"]
        assert c["totals_match_gpu"] is True
    finally:
        shutil.rmtree(d, ignore_errors=True)
