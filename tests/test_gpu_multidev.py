"""Multi-GPU in ONE process (the C host's path): IGD_DEVICES=a,b makes `igd search -q` replicate the
database on the listed devices, search contiguous query slabs on them concurrently (one host thread per device)
and add the per-dataset vectors.  The box has one GPU, so the same device is listed twice -- two engine handles,
two streams, two host threads: everything but the second physical GPU.  The table must be byte-identical to the
single-device run and to the oracle's counts (a sum of non-negative integers does not depend on the partition)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, Oracle, parse_hits_table

pytestmark = pytest.mark.gpu
IGD = os.path.join(ROOT, "bin", "igd")


def _run(args, devices=None):
    env = dict(os.environ)
    env.pop("IGD_DEVICES", None)
    if devices:
        env["IGD_DEVICES"] = devices
    p = subprocess.run([IGD] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    return p.stdout


@pytest.mark.parametrize("family,extra", [("smallrand", []), ("smallrand", ["-v", "400"]), ("gtype0", []), ("quirk", [])])
def test_two_engine_handles_give_the_single_device_table(family, extra):
    db, q = os.path.join(GOLDEN, family, "db.igd"), os.path.join(GOLDEN, family, "q.bed")
    one = _run(["search", db, "-q", q] + extra)
    for devs in ("0,0", "0,0,0", "0"):
        assert _run(["search", db, "-q", q] + extra, devs) == one, devs
    o = Oracle(db)
    ichr, qs, qe = o.read_queries(q)
    v = int(extra[1]) if extra else 0
    want, _ = o.search(ichr, qs, qe, v)
    got, total = parse_hits_table(one.decode(), o.nfiles)
    np.testing.assert_array_equal(got, want)
    o.close()


def test_unusable_device_in_the_list_fails_loudly():
    db, q = os.path.join(GOLDEN, "smallrand", "db.igd"), os.path.join(GOLDEN, "smallrand", "q.bed")
    env = dict(os.environ, IGD_DEVICES="0,99")
    p = subprocess.run([IGD, "search", db, "-q", q], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode != 0 and b"Total" not in p.stdout and b"out of range" in p.stderr
