"""AddressSanitizer + UBSan over the HOST C code (CPU build only; GPU sanitizers are not available):
  - the oracle CLI on every golden command line and on `create`;
  - the product's host code (igd_main.c, igd_cli_abi.c, igd_core.c, igd_hostpath.c, igd_create.c) up to the point
    where it needs the GPU: on this GPU-less host `create` parses all its input (threads, dictionaries, the
    sequential re-parse for mixed columns, long-line cutting) and then fails loudly; `search` reads the
    header and the index and answers the small golden query files on the host (igd_hostpath.c: -q, -v, -f, 1 and
    5 threads) with the reference's bytes.  Any sanitizer report fails the test.
"""
import glob
import json
import os
import random
import shutil
import subprocess

import pytest

from helpers import GOLDEN, ROOT, short_tmpdir
from test_oracle_create import write_beds

SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")


@pytest.fixture(scope="module")
def sanbin():
    d = short_tmpdir("igsan")
    orc = os.path.join(d, "igd_oracle_san")
    subprocess.check_call(["gcc", "-std=gnu99", *SAN, "-o", orc, "oracle/igd_oracle.c", "oracle/igd_oracle_create.c",
                           "oracle/igd_oracle_main.c", "-lz"], cwd=ROOT)
    igd = os.path.join(d, "igd_san")
    src = ["igd_amd/csrc/igd_main.c", "igd_amd/csrc/igd_cli_abi.c", "igd_amd/csrc/igd_core.c", "igd_amd/csrc/igd_hostpath.c", "igd_amd/csrc/igd_create.c",
           "igd_amd/csrc/igd_hip_lazy.c"]                    # the engine library is dlopen'ed through the rpath at the first engine call
    subprocess.check_call(["gcc", "-std=gnu99", *SAN, "-Iinclude", "-Iigd_amd/csrc", "-o", igd, *src,
                           "-lz", "-lpthread", "-ldl", "-Wl,-rpath," + os.path.join(ROOT, "igd_amd/lib")], cwd=ROOT)
    ing = os.path.join(d, "ingest_san")
    subprocess.check_call(["gcc", "-std=gnu99", *SAN, "-Iinclude", "-Iigd_amd/csrc", "-o", ing, "tests/c/ingest_san_main.c",
                           "igd_amd/csrc/igd_core.c", "igd_amd/csrc/igd_hostpath.c", "-Ligd_amd/lib", "-ligd_hip", "-lz", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "igd_amd/lib")], cwd=ROOT)
    yield {"orc": orc, "igd": igd, "ingest": ing, "dir": d}
    shutil.rmtree(d, ignore_errors=True)


def gpu_present():
    from igd_amd import _native
    return _native.hip().igd_hip_device_count() > 0


def run_clean(cmd, **kw):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=ENV, timeout=600, **kw)
    err = p.stderr.decode(errors="replace")
    assert "AddressSanitizer" not in err and "runtime error" not in err and p.returncode not in (97, 98), err[-3000:]
    return p


def test_oracle_clean_on_all_golden_command_lines(sanbin):
    n = 0
    for man in sorted(glob.glob(os.path.join(GOLDEN, "*", "manifest.json"))):
        case = os.path.dirname(man)
        d = short_tmpdir("igs")
        try:
            dst = os.path.join(d, "c")
            shutil.copytree(case, dst)
            for run in json.load(open(man))["runs"]:
                if "-o" in run["args"]:
                    continue
                run_clean([sanbin["orc"]] + run["args"], cwd=dst)
                n += 1
        finally:
            shutil.rmtree(d, ignore_errors=True)
    assert n >= 30


@pytest.mark.parametrize("mode", ["glob", "gtype0", "mixed", "list", "bed4"])
def test_create_host_code_clean(sanbin, mode):
    rng = random.Random(99)
    d = short_tmpdir("igs")
    try:
        write_beds(rng, d + "/in", 13, 300, 1 << 12, 3 if mode == "gtype0" else 5, gz_some=True)
        if mode == "mixed":
            open(d + "/in/f003.bed", "a").write("chr1\t5\t9\nchrZ\t1\t2\tq\n\n\nchr1\t" + "9" * 30 + "\t12\tx\t" + "7" * 25 + "\n" + "chr2\t1\t5\t" + "w" * 3000 + "\t5\n")
        arg, extra = d + "/in/", []
        if mode == "gtype0":
            extra = ["-s", "0"]
        if mode == "list":
            arg = d + "/list.txt"
            open(arg, "w").write("".join(p + "\n" for p in sorted(glob.glob(d + "/in/*"))) + "/nonexistent\n")
            extra = ["-f"]
        if mode == "bed4":
            arg = d + "/all.bed"
            open(arg, "w").write("".join("chr%d\t%d\t%d\tD%d\t%d\n" % (i % 3, 10 * i, 10 * i + 7 + i % 50, i % 11, i) for i in range(5000)) + "short\tline\n")
            extra = ["-s", "2"]
        exes = [sanbin["orc"]] if gpu_present() else [sanbin["orc"], sanbin["igd"]]   # host ASan + the HIP runtime: CPU hosts only
        for exe in exes:
            shutil.rmtree(d + "/o", ignore_errors=True)
            p = run_clean([exe, "create", arg, d + "/o", "db", "-b", "12"] + extra)
            if exe == sanbin["igd"]:
                assert b"no CPU path" in p.stderr and p.returncode != 0
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_search_host_code_clean_until_the_gpu_is_needed(sanbin):
    if gpu_present():
        pytest.skip("host ASan build is exercised on GPU-less hosts only")
    n = 0
    for fam in ("parse", "edge", "quirk", "gtype0", "smallrand"):
        case = os.path.join(GOLDEN, fam)
        d = short_tmpdir("igs")
        try:
            dst = os.path.join(d, "c")
            shutil.copytree(case, dst)
            for run in json.load(open(os.path.join(case, "manifest.json")))["runs"]:
                if "-o" in run["args"] or "-s" in run["args"] or "-m" in run["args"]:
                    continue
                for threads in ("1", "5"):
                    p = subprocess.run([sanbin["igd"]] + run["args"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=dst,
                                       env=dict(ENV, IGD_HOST_THREADS=threads))
                    err = p.stderr.decode(errors="replace")
                    assert "AddressSanitizer" not in err and "runtime error" not in err and p.returncode == 0, err[-3000:]
                    assert p.stdout.decode() == open(os.path.join(dst, run["stdout"])).read(), (fam, run["args"])
                    n += 1
        finally:
            shutil.rmtree(d, ignore_errors=True)
    assert n >= 40


def test_query_ingest_clean_threaded_and_sequential(sanbin):
    """igdc_read_queries on the parse fixtures (odd lines, CRLF, gz) and on a large text that takes the
    threaded path; threaded and sequential runs must agree."""
    d = short_tmpdir("igs")
    try:
        case = os.path.join(GOLDEN, "parse")
        beds = sorted(glob.glob(case + "/*.bed*"))
        assert beds
        rng = random.Random(4)
        big = d + "/big.bed"
        with open(big, "w") as fh:
            for i in range(300000):
                c = rng.choice(["chr1", "chr2", "chrX", "chr9", "1", "chr"])
                s = rng.randrange(0, 10 ** 7)
                fh.write("%s\t%d\t%d%s\n" % (c, s, s + rng.randrange(-5, 5000), rng.choice(["", "\tname\t5", "\r"])))
            fh.write("chr1\t5")                                   # no newline at EOF, short line
        for bed in beds + [big]:
            for rc in ("1", "0"):
                outs = []
                for env in ({}, {"IGD_PARSE_SEQUENTIAL": "1"}, {"IGD_PARSE_THREADS": "7"}):
                    p = subprocess.run([sanbin["ingest"], case + "/db.igd", bed, rc], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                       env=dict(ENV, **env), timeout=600)
                    err = p.stderr.decode(errors="replace")
                    assert "AddressSanitizer" not in err and "runtime error" not in err and p.returncode == 0, err[-3000:]
                    outs.append(p.stdout.split()[:2])
                assert outs[0] == outs[1] == outs[2], (bed, rc, outs)
    finally:
        shutil.rmtree(d, ignore_errors=True)
