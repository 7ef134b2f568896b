"""ctypes bindings of the native libraries (built in-tree under igd_amd/lib by `make`).

Nothing here computes overlaps: every search entry point ends in libigd_hip.so (hand-written
HIP for gfx950).  If a library is missing this module raises -- there is no Python or CPU
fallback."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# IGD_AMD_LIBDIR: kernel-variant experiments (tools/ab.sh, tools/valu_ab.sh) load another build of the libraries.  Which
# build is mapped is never a guess: igd_hip_build_flags() names it, and Database() refuses one that gives wrong counts.
LIBDIR = os.environ.get("IGD_AMD_LIBDIR") or os.path.join(HERE, "lib")

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)


class NativeMissing(RuntimeError):
    pass


def build(verbose=False):
    """Compile every native target for gfx950 (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", ROOT, "all"], stdout=out)


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process cannot both own the GPU, so when torch is installed but not imported
    yet, map ITS runtime first; libigd_hip.so's NEEDED libamdhip64.so.7 then binds to it by
    SONAME and a later `import torch` finds its own file already mapped."""
    import sys
    if "torch" in sys.modules or os.environ.get("IGD_AMD_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def _load(name):
    path = os.path.join(LIBDIR, name)
    if not os.path.exists(path):
        raise NativeMissing(
            "%s not built: run `make` (or `python -c 'import __graft_entry__ as g; g.build()'`) "
            "in %s -- igd_amd has no fallback path" % (path, ROOT))
    return C.CDLL(path)


class HipDesc(C.Structure):
    _fields_ = [("nbp", C.c_int32), ("gType", C.c_int32), ("nCtg", C.c_int32), ("nFiles", C.c_int32),
                ("nTile", i32p), ("nCnt", i32p), ("records", C.c_void_p), ("nRecords", C.c_int64),
                ("fd", C.c_int), ("fd_offset", C.c_int64)]


class HipCreateDesc(C.Structure):
    _fields_ = [("nbp", C.c_int32), ("gType", C.c_int32), ("nCtg", C.c_int32), ("n", C.c_int64),
                ("ctg", C.c_void_p), ("start", C.c_void_p), ("end", C.c_void_p), ("value", C.c_void_p),
                ("file", C.c_void_p), ("ctgName", C.c_void_p), ("out_fd", C.c_int)]


class HipCreated(C.Structure):
    _fields_ = [("nTile", C.c_void_p), ("nCnt", C.c_void_p), ("nTiles", C.c_int64), ("nRecords", C.c_int64),
                ("records", C.c_void_p)]


class HipHit(C.Structure):
    _fields_ = [("q", C.c_int32), ("idx", C.c_int32), ("start", C.c_int32), ("end", C.c_int32)]


class HipStats(C.Structure):
    _fields_ = [("queries", C.c_int64), ("pairs", C.c_int64), ("S", C.c_int64), ("B", C.c_int64),
                ("H", C.c_int64)]


class CoreDb(C.Structure):
    """struct igdc_db of igd_amd/csrc/igd_core.h"""
    _fields_ = [("nbp", C.c_int32), ("gType", C.c_int32), ("nCtg", C.c_int32), ("nFiles", C.c_int32),
                ("nTile", i32p), ("nCntFlat", i32p), ("nCnt", C.POINTER(i32p)),
                ("tBase", i64p), ("reserved_", C.c_void_p),
                ("cName", C.POINTER(C.c_char_p)), ("fileName", C.POINTER(C.c_char_p)),
                ("fileNr", i32p), ("fileMd", C.POINTER(C.c_double)),
                ("nTileTotal", C.c_int64), ("nRecords", C.c_int64), ("dataOff", C.c_int64),
                ("dict", i32p), ("dictCap", C.c_int32), ("dev", C.c_void_p),
                ("ndev", C.c_int32), ("devs", C.c_void_p * 16), ("grp", C.c_void_p)]


class CoreQueries(C.Structure):
    _fields_ = [("n", C.c_int64), ("cap", C.c_int64), ("ichr", i32p), ("qs", i32p), ("qe", i32p),
                ("unsorted", C.c_int32), ("max_len", C.c_int32)]


IGD_HIP_RULE_NEST = 0
IGD_HIP_RULE_FLAT = 1
class HipTraffic(C.Structure):
    _fields_ = [("units", C.c_int64), ("records", C.c_int64), ("record_bytes", C.c_int64), ("unit_bytes", C.c_int64),
                ("query_bytes", C.c_int64), ("slab_bytes", C.c_int64), ("total", C.c_int64)]


ENUM_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_void_p)
ENUM_SINK8 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int)

IGD_HIP_NO_VALUE_FILTER = -(2 ** 31)
IGD_HIP_FLAG_SORTED = 1
IGD_HIP_FLAG_BUCKET = 2
IGD_HIP_FLAG_EXACT = 4
IGD_HIP_FLAG_ZERO_FIRST = 8
IGD_HIP_FLAG_SHORT = 16        # with SORTED: no query is as long as a tile (verified on the device; dense batches take the DIRECT step)
IGD_HIP_ERR_UNSORTED = -4

_hip = None
_cli = None
_py = None
_synth = None
_r = None


def hip():
    """libigd_hip.so -- include/igd_hip.h"""
    global _hip
    if _hip is None:
        _share_hip_runtime_with_torch()
        L = _load("libigd_hip.so")
        L.igd_hip_device_count.restype = C.c_int
        L.igd_hip_last_error.restype = C.c_char_p
        L.igd_hip_open.argtypes = [C.POINTER(HipDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.igd_hip_close.argtypes = [C.c_void_p]
        L.igd_hip_nfiles.argtypes = [C.c_void_p]
        L.igd_hip_nfiles.restype = C.c_int32
        L.igd_hip_resident_bytes.argtypes = [C.c_void_p]
        L.igd_hip_resident_bytes.restype = C.c_int64
        L.igd_hip_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_int32, C.c_int, C.c_void_p, i64p]
        L.igd_hip_search_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int32, C.c_int, C.c_int, C.c_void_p, i64p]
        L.igd_hip_search_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_int32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.igd_hip_search_runs_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                              C.c_int32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.igd_hip_max_batch.restype = C.c_int64
        L.igd_hip_build_flags.restype = C.c_uint
        L.igd_hip_build_wrong_counts.restype = C.c_uint
        L.igd_hip_sync.argtypes = [C.c_void_p, C.c_void_p]
        L.igd_hip_sync_spin.argtypes = [C.c_void_p, C.c_void_p]
        L.igd_hip_enumerate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.POINTER(C.POINTER(HipHit)), i64p]
        L.igd_hip_free.argtypes = [C.c_void_p]
        L.igd_hip_enumerate_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                               C.c_void_p, ENUM_SINK, C.c_void_p, i64p]
        L.igd_hip_enumerate_stream8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                C.c_void_p, ENUM_SINK8, C.c_void_p, i64p]
        L.igd_hip_hit8_idx_bits.argtypes = [C.c_void_p]
        L.igd_hip_batch_traffic.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_int32, C.c_int, C.c_int, C.POINTER(HipTraffic)]
        L.igd_hip_measure_rates.argtypes = [C.c_int, C.POINTER(C.c_double)]
        L.igd_hip_seqpare_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
        L.igd_hip_batch_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_int32, C.c_int, C.POINTER(HipStats)]
        L.igd_hip_hitmap.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, i64p]
        L.igd_hip_profile_begin.argtypes = [C.c_void_p, C.c_int]
        L.igd_hip_profile_end.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double)]
        L.igd_hip_profile_sampling.argtypes = [C.c_void_p, C.c_int]
        L.igd_hip_scan_kernel_name.restype = C.c_char_p
        L.igd_hip_last_scan_kernel.restype = C.c_char_p
        L.igd_hip_last_scan_kernel.argtypes = [C.c_void_p]
        L.igd_hip_seqpare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
        L.igd_hip_create.argtypes = [C.POINTER(HipCreateDesc), C.c_int, C.POINTER(HipCreated)]
        L.igd_hip_created_free.argtypes = [C.POINTER(HipCreated)]
        _hip = L
    return _hip


def create_arrays(nbp, gtype, nctg, ctg, start, end, value, file, device=0):
    """igd_hip_create on numpy int32 arrays -> dict(nTile, nCnt, records[nRecords, 4|3])."""
    import numpy as np
    L = hip()
    arrs = [np.ascontiguousarray(a, dtype=np.int32) if a is not None else None for a in (ctg, start, end, value, file)]
    p = [a.ctypes.data_as(C.c_void_p) if a is not None else None for a in arrs]
    d = HipCreateDesc(int(nbp), int(gtype), int(nctg), len(arrs[0]), p[0], p[1], p[2], p[3], p[4], None, -1)
    out = HipCreated()
    rc = L.igd_hip_create(C.byref(d), int(device), C.byref(out))
    if rc != 0:
        raise RuntimeError("igd_hip_create failed (%d): %s" % (rc, L.igd_hip_last_error().decode()))
    try:
        w = 3 if gtype == 0 else 4
        ntile = np.ctypeslib.as_array(C.cast(out.nTile, C.POINTER(C.c_int32)), (max(int(nctg), 1),))[:nctg].copy()
        ncnt = np.ctypeslib.as_array(C.cast(out.nCnt, C.POINTER(C.c_int32)), (max(int(out.nTiles), 1),))[:out.nTiles].copy()
        if out.nRecords > 0:
            recs = np.ctypeslib.as_array(C.cast(out.records, C.POINTER(C.c_int32)), (int(out.nRecords), w)).copy()
        else:
            recs = np.zeros((0, w), np.int32)
    finally:
        L.igd_hip_created_free(C.byref(out))
    return {"nTile": ntile, "nCnt": ncnt, "records": recs}


def _bind_core(L):
    L.igdc_open.restype = C.POINTER(CoreDb)
    L.igdc_open.argtypes = [C.c_char_p]
    L.igdc_index_path.restype = C.c_void_p
    L.igdc_index_path.argtypes = [C.c_char_p]
    L.igdc_load_index.argtypes = [C.POINTER(CoreDb), C.c_char_p]
    L.igdc_close.argtypes = [C.POINTER(CoreDb)]
    L.igdc_get_id.argtypes = [C.POINTER(CoreDb), C.c_char_p]
    L.igdc_get_id.restype = C.c_int32
    L.igdc_attach_path.argtypes = [C.POINTER(CoreDb), C.c_char_p, C.c_int]
    L.igdc_parse_bed.restype = C.c_void_p
    L.igdc_parse_bed.argtypes = [C.c_char_p, i32p, i32p, C.c_int]
    L.igdc_read_queries.argtypes = [C.POINTER(CoreDb), C.c_char_p, C.c_int, C.POINTER(CoreQueries)]
    L.igdc_queries_free.argtypes = [C.POINTER(CoreQueries)]
    return L


def cli():
    """libigd.so -- include/igd_search.h + include/igd_base.h (+ the igdc_* core)"""
    global _cli
    if _cli is None:
        hip()
        L = _bind_core(_load("libigd.so"))
        L.igd_search.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.igd_engine_status.restype = C.c_int
        L.get_igdinfo.restype = C.c_void_p
        L.get_igdinfo.argtypes = [C.c_char_p]
        L.get_id.argtypes = [C.c_char_p]
        L.get_id.restype = C.c_int32
        L.parse_bed.restype = C.c_void_p
        L.parse_bed.argtypes = [C.c_char_p, i32p, i32p]
        L.bSearch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.bSearch.restype = C.c_int32
        _cli = L
    return _cli


def pyabi():
    """libigd_py.so -- include/igd_py_abi.h"""
    global _py
    if _py is None:
        hip()
        L = _load("libigd_py.so")
        L.iGD_init.restype = C.c_void_p
        L.get_nFiles.argtypes = [C.c_void_p]
        L.get_nFiles.restype = C.c_int32
        L.open_iGD.argtypes = [C.c_void_p, C.c_char_p]
        L.close_iGD.argtypes = [C.c_void_p]
        L.create_iGD.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.get_overlaps.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, i64p]
        L.get_overlaps.restype = None
        L.getOverlaps.argtypes = [C.c_void_p, C.c_char_p, i64p]
        L.getOverlaps.restype = C.c_int64
        L.igd_engine_status.restype = C.c_int
        L.igd_engine_clear.restype = None
        _py = L
    return _py


def rabi():
    """libigdr.so -- include/igdr_abi.h (plain-C / .C entry points)"""
    global _r
    if _r is None:
        hip()
        L = _load("libigdr.so")
        L.open_iGD.restype = C.c_void_p
        L.open_iGD.argtypes = [C.c_char_p]
        L.close_iGD.argtypes = [C.c_void_p]
        L.get_overlaps32.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, i32p]
        L.igdr_search_n32.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), i32p, i32p, i32p]
        L.search_1.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), i32p, i32p, i64p]
        L.getOverlaps.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), i64p]
        L.igd_engine_status.restype = C.c_int
        L.igd_engine_clear.restype = None
        _r = L
    return _r


def synth():
    """libigd_synth.so -- tools/igd_synth.c"""
    global _synth
    if _synth is None:
        hip()
        L = _load("libigd_synth.so")
        L.igd_synth_db.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int,
                                   C.c_int, C.c_int32, C.c_int32, C.c_int, C.c_int]
        L.igd_synth_db_beds.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_uint64, C.c_int, C.c_int,
                                        C.c_int32, C.c_int32]
        L.igd_synth_queries.argtypes = [C.c_int64, C.c_uint64, C.c_int, C.c_int32, C.c_int32, C.c_int,
                                        C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.igd_synth_queries.restype = C.c_int64
        L.igd_synth_queries_slab.argtypes = [C.c_int64, C.c_uint64, C.c_int, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
        L.igd_synth_queries_slab.restype = C.c_int64
        L.igd_synth_write_bed.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.igd_synth_contig_name.restype = C.c_char_p
        L.igd_synth_contig_name.argtypes = [C.c_int, C.c_int]
        L.igd_synth_ncontigs.argtypes = [C.c_int]
        _synth = L
    return _synth


def free(ptr):
    C.CDLL(None).free(C.c_void_p(ptr))
