"""Shared test plumbing: paths, on-demand builds, the ctypes face of the CPU oracle,
an independent numpy `.igd` writer, and runners for the oracle / reference binaries.

The oracle (oracle/) is the CHECKER: it is imported here (tests/ is allowed to) and never by
the product package.  The reference binary oracle/_ref/igd exists only where
oracle/Makefile could build it from /root/reference (the build container) or where the
prebuilt file travelled to (the GPU box); tests that need it skip when it is absent.
"""
import ctypes as C
import gzip
import os
import shutil
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
ORACLE_BIN = os.path.join(ORACLE_DIR, "_build", "igd_oracle")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "igd")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def build_oracle():
    if not (os.path.exists(ORACLE_LIB) and os.path.exists(ORACLE_BIN)) or any(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(ORACLE_LIB)
        for f in ("igd_oracle.c", "igd_oracle_create.c", "igd_oracle_main.c", "igd_oracle.h")
    ):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return ORACLE_LIB


def have_ref():
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def short_tmpdir(prefix="igt"):
    """The reference binary overflows 64/128-byte path buffers:
    keep every path it sees short."""
    base = "/tmp"
    return tempfile.mkdtemp(prefix=prefix, dir=base)


# --------------------------------------------------------------------------------------
# oracle via ctypes
class OrcHit(C.Structure):
    _fields_ = [("idx", C.c_int32), ("start", C.c_int32), ("end", C.c_int32)]


class OrcStats(C.Structure):
    _fields_ = [("queries", C.c_int64), ("pairs", C.c_int64), ("S", C.c_int64),
                ("H", C.c_int64), ("B", C.c_int64)]


_orc = None


def orc():
    global _orc
    if _orc is None:
        lib = C.CDLL(build_oracle())
        i32p = C.POINTER(C.c_int32)
        i64p = C.POINTER(C.c_int64)
        lib.orc_open.restype = C.c_void_p
        lib.orc_open.argtypes = [C.c_char_p]
        lib.orc_close.argtypes = [C.c_void_p]
        lib.orc_preload.argtypes = [C.c_void_p]
        for name in ("orc_nfiles", "orc_nctg", "orc_nbp", "orc_gtype"):
            getattr(lib, name).restype = C.c_int32
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.orc_ntile.restype = C.c_int32
        lib.orc_ntile.argtypes = [C.c_void_p, C.c_int32]
        lib.orc_ncnt.restype = C.c_int32
        lib.orc_ncnt.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.orc_ctg_name.restype = C.c_char_p
        lib.orc_ctg_name.argtypes = [C.c_void_p, C.c_int32]
        lib.orc_file_name.restype = C.c_char_p
        lib.orc_file_name.argtypes = [C.c_void_p, C.c_int32]
        lib.orc_file_nr.restype = C.c_int32
        lib.orc_file_nr.argtypes = [C.c_void_p, C.c_int32]
        lib.orc_get_id.restype = C.c_int32
        lib.orc_get_id.argtypes = [C.c_void_p, C.c_char_p]
        lib.orc_get_stats.restype = C.POINTER(OrcStats)
        lib.orc_get_stats.argtypes = [C.c_void_p]
        lib.orc_reset_stats.argtypes = [C.c_void_p]
        lib.orc_parse_bed.restype = C.c_void_p
        lib.orc_parse_bed.argtypes = [C.c_char_p, i32p, i32p]
        lib.orc_read_queries.restype = C.c_int64
        lib.orc_read_queries.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(i32p), C.POINTER(i32p), C.POINTER(i32p)]
        lib.orc_get_overlaps.restype = C.c_int32
        lib.orc_get_overlaps.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, i64p]
        lib.orc_get_overlaps_v.restype = C.c_int32
        lib.orc_get_overlaps_v.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, i64p]
        lib.orc_getOverlaps.restype = C.c_int64
        lib.orc_getOverlaps.argtypes = [C.c_void_p, C.c_char_p, i64p]
        lib.orc_getOverlaps_v.restype = C.c_int64
        lib.orc_getOverlaps_v.argtypes = [C.c_void_p, C.c_char_p, i64p, C.c_int32]
        lib.orc_search_batch.restype = C.c_int64
        lib.orc_search_batch.argtypes = [C.c_void_p, i32p, i32p, i32p, C.c_int64, C.c_int32, i64p]
        lib.orc_enumerate_batch.restype = C.c_int64
        lib.orc_enumerate_batch.argtypes = [C.c_void_p, i32p, i32p, i32p, C.c_int64, i64p,
                                            C.POINTER(OrcHit), C.c_int64]
        lib.orc_getMap.restype = C.c_int64
        lib.orc_getMap.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_void_p]
        lib.free = C.CDLL(None).free
        lib.free.argtypes = [C.c_void_p]
        _orc = lib
    return _orc


def _p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class Oracle:
    """Thin object over liboracle.so for one .igd."""

    def __init__(self, igd_path, preload=True):
        self.lib = orc()
        self.h = self.lib.orc_open(igd_path.encode())
        if not self.h:
            raise RuntimeError("oracle cannot open %s" % igd_path)
        if preload:
            self.lib.orc_preload(self.h)
        self.nfiles = self.lib.orc_nfiles(self.h)
        self.nctg = self.lib.orc_nctg(self.h)
        self.nbp = self.lib.orc_nbp(self.h)
        self.gtype = self.lib.orc_gtype(self.h)

    def close(self):
        if self.h:
            self.lib.orc_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ctg_names(self):
        return [self.lib.orc_ctg_name(self.h, i).decode() for i in range(self.nctg)]

    def get_id(self, name):
        return self.lib.orc_get_id(self.h, name.encode())

    def read_queries(self, qfile):
        pc, ps, pe = (C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)())
        n = self.lib.orc_read_queries(self.h, qfile.encode(), C.byref(pc), C.byref(ps), C.byref(pe))
        if n < 0:
            raise IOError(qfile)
        out = []
        for p in (pc, ps, pe):
            a = np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy() if n else np.zeros(0, np.int32)
            out.append(a.astype(np.int32))
            self.lib.free(C.cast(p, C.c_void_p))
        return out

    def search(self, ichr, qs, qe, v=0):
        ichr, qs, qe = (np.ascontiguousarray(a, dtype=np.int32) for a in (ichr, qs, qe))
        hits = np.zeros(max(self.nfiles, 1), np.int64)
        self.lib.orc_reset_stats(self.h)
        total = self.lib.orc_search_batch(self.h, _p32(ichr), _p32(qs), _p32(qe), len(qs), v, _p64(hits))
        return hits[: self.nfiles], total

    def stats(self):
        s = self.lib.orc_get_stats(self.h).contents
        return dict(queries=s.queries, pairs=s.pairs, S=s.S, H=s.H, B=s.B)

    def enumerate(self, ichr, qs, qe):
        ichr, qs, qe = (np.ascontiguousarray(a, dtype=np.int32) for a in (ichr, qs, qe))
        n = len(qs)
        qoff = np.zeros(n + 1, np.int64)
        total = self.lib.orc_enumerate_batch(self.h, _p32(ichr), _p32(qs), _p32(qe), n, _p64(qoff), None, 0)
        buf = (OrcHit * max(total, 1))()
        self.lib.orc_enumerate_batch(self.h, _p32(ichr), _p32(qs), _p32(qe), n, _p64(qoff), buf, total)
        arr = np.frombuffer(buf, dtype=np.int32).reshape(-1, 3)[:total].copy()
        return qoff, arr  # columns idx,start,end

    def hitmap(self, v=0):
        m = np.zeros((self.nfiles, self.nfiles), np.uint32)
        tot = self.lib.orc_getMap(self.h, 1 if v > 0 else 0, v, m.ctypes.data, None)
        return m, tot

    def file_search(self, qfile, v=0):
        hits = np.zeros(max(self.nfiles, 1), np.int64)
        if v > 0 and self.gtype != 0:
            ret = self.lib.orc_getOverlaps_v(self.h, qfile.encode(), _p64(hits), v)
        else:
            ret = self.lib.orc_getOverlaps(self.h, qfile.encode(), _p64(hits))
        return hits[: self.nfiles], ret


# --------------------------------------------------------------------------------------
# independent numpy writer of the on-disk format.  NOT the product
# writer: tests use it so that loader/engine bugs cannot hide behind a matching writer bug.
def write_igd_numpy(path_igd, files, nbp=16384, gtype=1, contig_order=None, file_names=None):
    """files: list (one per source file) of lists of (chrom, start, end, value).
    Mirrors igd_add/igd_save: drop start>=end, copy the record to every tile
    start//nbp..(end-1)//nbp, stable-sort each tile by start, contigs in first-seen order."""
    ctg_index = {}
    ctgs = []
    if contig_order:
        for c in contig_order:
            ctg_index[c] = len(ctgs)
            ctgs.append(c)
    tiles = []  # per contig: dict tile -> list of (start,seq,idx,end,value)
    ntile = []
    nr = []
    avg = []
    seq = 0
    for idx, recs in enumerate(files):
        nr.append(len(recs))
        avg.append(sum(e - s for (_, s, e, _) in recs) / max(len(recs), 1))
        for (c, s, e, v) in recs:
            if s >= e:
                continue
            if c not in ctg_index:
                ctg_index[c] = len(ctgs)
                ctgs.append(c)
            k = ctg_index[c]
            while len(tiles) <= k:
                tiles.append({})
                ntile.append(0)
            n1, n2 = s // nbp, (e - 1) // nbp
            ntile[k] = max(ntile[k], n2 + 1)
            for j in range(n1, n2 + 1):
                tiles[k].setdefault(j, []).append((s, seq, idx, e, v))
                seq += 1
    while len(tiles) < len(ctgs):
        tiles.append({})
        ntile.append(1)
    with open(path_igd, "wb") as f:
        f.write(np.array([nbp, gtype, len(ctgs)], np.int32).tobytes())
        f.write(np.array(ntile, np.int32).tobytes())
        for k in range(len(ctgs)):
            cnt = np.zeros(ntile[k], np.int32)
            for j, lst in tiles[k].items():
                cnt[j] = len(lst)
            f.write(cnt.tobytes())
        for c in ctgs:
            b = c.encode()[:39]
            f.write(b + b"\0" + b"\xAA" * (39 - len(b)))  # garbage after NUL, as the reference leaves it
        for k in range(len(ctgs)):
            for j in sorted(tiles[k]):
                lst = sorted(tiles[k][j], key=lambda r: (r[0], r[1]))
                if gtype == 1:
                    a = np.array([(r[2], r[0], r[3], r[4]) for r in lst], np.int32)
                else:
                    a = np.array([(r[2], r[0], r[3]) for r in lst], np.int32)
                f.write(a.tobytes())
    tsv = os.path.splitext(path_igd)[0] + "_index.tsv"
    with open(tsv, "w") as f:
        f.write("Index\tFile\tNumber of regions\tAvg size\n")
        for i in range(len(files)):
            name = file_names[i] if file_names else "f%04d.bed" % i
            f.write("%d\t%s\t%d\t%f\n" % (i, name, nr[i], avg[i]))
    return ctgs


def write_bed(path, rows, gz=False):
    """rows: iterable of tuples; joined by tabs."""
    op = gzip.open if gz else open
    with op(path, "wt") as f:
        for r in rows:
            f.write("\t".join(str(x) for x in r) + "\n")


# --------------------------------------------------------------------------------------
# runners
def run_ref(args, cwd=None, timeout=600):
    """Run the reference binary; returns stdout (text)."""
    p = subprocess.run([REF_BIN] + list(args), cwd=cwd, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError("reference failed rc=%d: %s" % (p.returncode, p.stderr.decode()[-400:]))
    return p.stdout.decode()


def ref_create(bed_glob, out_dir, name, b=14, s0=False):
    """`igd create "<glob>" <out_dir>/ <name> -b <b> [-s 0]` (needs >= 10 files unless -s 0)."""
    if not out_dir.endswith("/"):
        out_dir += "/"
    os.makedirs(os.path.join(out_dir, "data0"), exist_ok=True)
    args = ["create", bed_glob, out_dir, name, "-b", str(b)]
    if s0:
        args += ["-s", "0"]
    run_ref(args)
    shutil.rmtree(os.path.join(out_dir, "data0"), ignore_errors=True)
    return os.path.join(out_dir, name + ".igd")


def run_oracle_cli(args, timeout=600):
    build_oracle()
    p = subprocess.run([ORACLE_BIN] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError("oracle cli failed: %s" % p.stderr.decode()[-400:])
    return p.stdout.decode()


def parse_hits_table(text, nfiles):
    """`igd search -q` stdout -> (hits vector, total or None)."""
    hits = np.zeros(nfiles, np.int64)
    total = None
    for line in text.splitlines():
        if line.startswith("Total:"):
            total = int(line.split(":")[1])
            continue
        parts = line.split("\t")
        if len(parts) == 4 and parts[0].strip().isdigit():
            hits[int(parts[0])] = int(parts[2])
    return hits, total


def build_r_call_harness(outdir):
    """igdr_abi.c WITH its `.Call` entry points (-DIGDR_HAVE_R) against the mock of R's C API (tests/mock_r: not R), linked
    with the harness tests/c/r_call_main.c and the product's engine library.  Returns the executable's path."""
    exe = os.path.join(outdir, "r_call_main")
    src = os.path.join(ROOT, "igd_amd", "csrc")
    lib = os.path.join(ROOT, "igd_amd", "lib")
    cmd = ["gcc", "-O1", "-g", "-std=gnu99", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror=implicit-function-declaration",
           "-Werror=incompatible-pointer-types", "-DIGDR_HAVE_R",
           "-I" + os.path.join(ROOT, "tests", "mock_r"), "-I" + os.path.join(ROOT, "include"), "-I" + src, "-I" + os.path.join(ROOT, "tools"),
           "-o", exe, os.path.join(ROOT, "tests", "c", "r_call_main.c"), os.path.join(ROOT, "tests", "mock_r", "mock_r.c"),
           os.path.join(src, "igdr_abi.c"), os.path.join(src, "igd_core.c"), os.path.join(src, "igd_hostpath.c"), os.path.join(src, "igd_create.c"),
           "-L" + lib, "-ligd_hip", "-lz", "-lpthread", "-Wl,-rpath," + lib]
    subprocess.check_call(cmd)
    return exe
