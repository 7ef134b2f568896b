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
    for devs in ("0,0", "0,0,0", "0", "0,0,0,0,0,0,0,0"):     # (eight handles: the shape of the 8-GPU node's one-process job)
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


def test_native_rccl_allreduce_in_the_c_host_with_one_rank():
    """The C host's own RCCL call-site (igd_hip_group_search: ncclCommInitAll over IGD_DEVICES, ncclAllReduce(d_hits, nFiles,
    ncclInt64, ncclSum) on the engine's stream).  The box has one GPU and RCCL wants one rank per GPU, so the group has ONE
    rank here (IGD_MULTI_REDUCE=rccl makes a one-device list take the group path and refuses a silent host add): the table
    equals the plain run's, `[igd timing]` names the reducer, and librccl was mapped by bin/igd -- which a plain one-device
    run never does."""
    db, q = os.path.join(GOLDEN, "smallrand", "db.igd"), os.path.join(GOLDEN, "smallrand", "q.bed")
    one = _run(["search", db, "-q", q])
    env = dict(os.environ, IGD_DEVICES="0", IGD_MULTI_REDUCE="rccl", IGD_TIMING="1", LD_DEBUG="files")
    p = subprocess.run([IGD, "search", db, "-q", q], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert p.stdout == one
    err = p.stderr.decode()
    assert "hits[] summed by: rccl" in err, err[-600:]
    assert "librccl" in err                                   # LD_DEBUG=files: the library was mapped (dlopen at group creation)
    plain = subprocess.run([IGD, "search", db, "-q", q], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, LD_DEBUG="files"), timeout=600)
    assert plain.returncode == 0 and "librccl" not in plain.stderr.decode()
    # a device listed twice cannot be an RCCL group (one rank per GPU): host add, and the note says why
    two = subprocess.run([IGD, "search", db, "-q", q], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=dict(os.environ, IGD_DEVICES="0,0", IGD_TIMING="1"), timeout=600)
    assert two.returncode == 0 and two.stdout == one and b"summed by: host -- a device is listed twice" in two.stderr
    # ... and insisting on RCCL then fails loudly instead of falling back
    bad = subprocess.run([IGD, "search", db, "-q", q], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=dict(os.environ, IGD_DEVICES="0,0", IGD_MULTI_REDUCE="rccl"), timeout=600)
    assert bad.returncode != 0 and b"Total" not in bad.stdout
