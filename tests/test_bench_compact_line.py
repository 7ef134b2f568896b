"""bench.py's stdout contract without a GPU: the compact line built from a full record -- the real one of round 6
(profiles/r06/bench_extra.json) and one blown up far beyond it -- stays within the 8 KB the driver's stdout tail holds, keeps the
contract keys, and sheds its optional parts (never the contract keys) when it would not fit."""
import json
import os
import sys

from helpers import ROOT

sys.path.insert(0, ROOT)
import bench                                             # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_extra.json")))


def test_the_recorded_round_6_line_is_compact_and_complete():
    full = _full()
    text = bench.compact_line(full, "bench_extra.json")
    assert len(text) <= bench.LINE_CAP and "\n" not in text
    c = json.loads(text)
    for k in CONTRACT:
        assert k in c, k
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"]          # (exact: value = queries / time)
    r = c["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_launch", "traffic", "kernel_ms"):
        assert k in r, k
    assert abs(r["frac"] - full["roofline"]["frac"]) < 1e-5 and r["peak"] == 8000.0
    cb = c["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["cores"] == 1 and cb["totals_match_gpu"] is True and cb["seconds"] > 0
    assert c["matches_oracle"] is True and c["n_ranks_seen"] == 1 and "dropped_for_size" not in c
    rows = c["extra_configs"]
    assert len(rows) == len(full["extra_configs"]) and all(x["matches_oracle"] is True for x in rows)
    assert [x["workload"] for x in rows] == [e["key"] for e in full["extra_configs"]]


def test_an_oversized_record_sheds_optional_parts_not_contract_keys():
    full = _full()
    full["extra_configs"] = [dict(e, key="%s_%03d_%s" % (e["key"], i, "x" * 60)) for i in range(12) for e in full["extra_configs"]]
    full["devices"] = ["rank %d: cuda:%d %s" % (r, r, "y" * 200) for r in range(8)]
    text = bench.compact_line(full, "bench_extra.json")
    assert len(text) <= bench.LINE_CAP
    c = json.loads(text)
    for k in CONTRACT:
        assert k in c, k
    assert "extra_configs" in c.get("dropped_for_size", [])
    assert c["roofline"]["kernel_ms"] > 0 and c["cpu_baseline"]["value"] > 0 and c["extra_file"] == "bench_extra.json"
