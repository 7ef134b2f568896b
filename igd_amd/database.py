"""Python face of one .igd resident on one MI355X.

Mirrors what the reference's front-ends do around the hot path -- load header + index
(get_igdinfo / get_fileinfo), map contig names (get_id), read query files (parse_bed loop) --
and hands every search to the HIP engine (include/igd_hip.h).  numpy arrays for host
batches; raw device pointers (e.g. torch tensors' data_ptr()) for resident batches."""
import ctypes as C
import os

import numpy as np

from . import _native as N


class IgdError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        raise IgdError("%s failed (code %d): %s" % (what, rc, N.hip().igd_hip_last_error().decode()))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Database:
    def __init__(self, igd_path, device=0):
        self._L = N.cli()
        self._H = N.hip()
        self.build_flags = int(self._H.igd_hip_build_flags())
        if self._H.igd_hip_build_wrong_counts() and os.environ.get("IGD_HIP_ALLOW_EXP_BUILD") != "1":
            raise IgdError("%s is a measurement build (IGD_EXP=0x%x) whose counts are WRONG on purpose; "
                           "IGD_HIP_ALLOW_EXP_BUILD=1 loads it anyway" % (os.path.join(N.LIBDIR, "libigd_hip.so"), self.build_flags))
        self.path = igd_path
        self._core = self._L.igdc_open(igd_path.encode())
        if not self._core:
            raise IgdError("cannot read .igd header of %s" % igd_path)
        tsv = self._L.igdc_index_path(igd_path.encode())
        rc = self._L.igdc_load_index(self._core, C.cast(tsv, C.c_char_p))
        N.free(tsv)
        if rc != 0:
            self._L.igdc_close(self._core)
            self._core = None
            raise IgdError("cannot read the _index.tsv next to %s" % igd_path)
        c = self._core.contents
        self.nbp, self.gtype, self.nctg, self.nfiles = c.nbp, c.gType, c.nCtg, c.nFiles
        self.nrecords, self.ntiles = c.nRecords, c.nTileTotal
        self.contig_names = [c.cName[i].decode() for i in range(self.nctg)]
        self.file_names = [c.fileName[i].decode() for i in range(self.nfiles)]
        self.file_nr = [c.fileNr[i] for i in range(self.nfiles)]
        self.ntile = [c.nTile[i] for i in range(self.nctg)]
        rc = self._L.igdc_attach_path(self._core, igd_path.encode(), int(device))
        if rc != 0:
            err = self._H.igd_hip_last_error().decode()
            self._L.igdc_close(self._core)
            self._core = None
            raise IgdError("cannot put %s on GPU %d (code %d): %s -- there is no CPU search path"
                           % (igd_path, device, rc, err))
        self.dev = C.c_void_p(self._core.contents.dev)
        self.device = int(device)

    # ---- lifetime -----------------------------------------------------------------
    def close(self):
        if getattr(self, "_core", None):
            self._L.igdc_close(self._core)
            self._core = None
            self.dev = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def resident_bytes(self):
        return self._H.igd_hip_resident_bytes(self.dev)

    # ---- host helpers (reference: get_id, parse_bed loop) ---------------------------
    def contig_id(self, name):
        return self._L.igdc_get_id(self._core, name.encode())

    def read_queries(self, qfile, require_chr=True):
        q = N.CoreQueries()
        if self._L.igdc_read_queries(self._core, qfile.encode(), 1 if require_chr else 0, C.byref(q)) != 0:
            raise IOError("cannot open query file %s" % qfile)
        n = q.n
        if n:
            out = tuple(np.ctypeslib.as_array(p, shape=(n,)).copy() for p in (q.ichr, q.qs, q.qe))
        else:
            out = tuple(np.zeros(0, np.int32) for _ in range(3))
        self._L.igdc_queries_free(C.byref(q))
        return out

    @staticmethod
    def cli_dispatch(gtype, v):
        """(rule, engine v) that `igd search -q ... -v V` selects (src/igd_search.c:1023-1030)."""
        if gtype != 0 and v > 0:
            return N.IGD_HIP_RULE_FLAT, int(v)
        return N.IGD_HIP_RULE_NEST, N.IGD_HIP_NO_VALUE_FILTER

    # ---- searches ---------------------------------------------------------------------
    def search(self, ichr, qs, qe, v=0, rule=None, value_filter=None, hits=None, flags=0):
        """Host batch.  Default: the CLI dispatch for `-v v`.  Returns (hits int64[nfiles], total).
        flags: 0 (device checks the order and picks merge-join or bucketing), IGD_HIP_FLAG_SORTED [| IGD_HIP_FLAG_SHORT] (a
        promise, verified; a broken one is repaired by a second pass) or IGD_HIP_FLAG_BUCKET (the caller knows the batch is
        unordered: no order check, ~13 us per 10^6 queries less than 0)."""
        ichr, qs, qe = _i32(ichr), _i32(qs), _i32(qe)
        if rule is None:
            rule, vf = self.cli_dispatch(self.gtype, v)
        else:
            vf = N.IGD_HIP_NO_VALUE_FILTER if value_filter is None else int(value_filter)
        if hits is None:
            hits = np.zeros(max(self.nfiles, 1), np.int64)
        total = C.c_int64(0)
        _chk(self._H.igd_hip_search_ex(self.dev, ichr.ctypes.data, qs.ctypes.data, qe.ctypes.data, len(qs),
                                       vf, rule, int(flags), hits.ctypes.data, C.byref(total)), "igd_hip_search")
        return hits[: self.nfiles], total.value

    def search_dev(self, d_ichr, d_qs, d_qe, nq, d_hits, d_total=None, v=0, rule=None,
                   value_filter=None, stream=None, flags=0):
        """Resident batch: arguments are device pointers (ints).  Asynchronous.
        flags: 0, IGD_HIP_FLAG_SORTED (verified promise; sync() raises if broken) [| IGD_HIP_FLAG_SHORT: no query as long as a tile,
        verified too -- a dense batch then takes the DIRECT step] or IGD_HIP_FLAG_BUCKET (known to be unordered: no order check)."""
        if rule is None:
            rule, vf = self.cli_dispatch(self.gtype, v)
        else:
            vf = N.IGD_HIP_NO_VALUE_FILTER if value_filter is None else int(value_filter)
        _chk(self._H.igd_hip_search_dev(self.dev, d_ichr, d_qs, d_qe, int(nq), vf, rule, int(flags), d_hits,
                                        d_total, stream), "igd_hip_search_dev")

    def search_runs_dev(self, d_run_start, d_qs, d_qe, nq, d_hits, d_total=None, v=0, rule=None,
                        value_filter=None, stream=None, flags=0):
        """Resident position-sorted batch given as contig runs (device int32[nctg + 1]: queries [run[c], run[c + 1]) lie on
        contig c) instead of one contig number per query.  Asynchronous; implies the order promise."""
        if rule is None:
            rule, vf = self.cli_dispatch(self.gtype, v)
        else:
            vf = N.IGD_HIP_NO_VALUE_FILTER if value_filter is None else int(value_filter)
        _chk(self._H.igd_hip_search_runs_dev(self.dev, d_run_start, d_qs, d_qe, int(nq), vf, rule, int(flags), d_hits,
                                             d_total, stream), "igd_hip_search_runs_dev")

    @staticmethod
    def contig_runs(ichr, nctg):
        """run_start[nctg + 1] of a batch whose contig numbers are non-decreasing (host, numpy)."""
        ichr = np.ascontiguousarray(ichr, dtype=np.int32)
        if len(ichr) and (np.any(np.diff(ichr) < 0) or ichr[0] < 0 or ichr[-1] >= nctg):
            raise IgdError("contig_runs: the batch is not grouped by ascending contig number")
        return np.searchsorted(ichr, np.arange(nctg + 1, dtype=np.int32), side="left").astype(np.int32)

    def sync(self, stream=None, spin=False):
        """Wait for the stream (spin: polling instead of sleeping on the completion signal) and surface asynchronous errors."""
        _chk((self._H.igd_hip_sync_spin if spin else self._H.igd_hip_sync)(self.dev, stream), "igd_hip_sync")

    def enumerate(self, ichr, qs, qe):
        """`-f`: returns (qoff int64[nq+1], records int32[n,4] = q,idx,start,end) in reference order."""
        ichr, qs, qe = _i32(ichr), _i32(qs), _i32(qe)
        nq = len(qs)
        qoff = np.zeros(nq + 1, np.int64)
        out = C.POINTER(N.HipHit)()
        total = C.c_int64(0)
        _chk(self._H.igd_hip_enumerate(self.dev, ichr.ctypes.data, qs.ctypes.data, qe.ctypes.data, nq,
                                       qoff.ctypes.data, C.byref(out), C.byref(total)), "igd_hip_enumerate")
        n = total.value
        if n:
            rec = np.ctypeslib.as_array(C.cast(out, N.i32p), shape=(n * 4,)).reshape(n, 4).copy()
            self._H.igd_hip_free(out)
        else:
            rec = np.zeros((0, 4), np.int32)
        return qoff, rec

    def enumerate_stream(self, ichr, qs, qe, on_chunk=None):
        """`-f`, streamed (igd_hip_enumerate_stream): on_chunk(q0, q1, qoff, rec) is called per chunk with
        rec = int32[n,4] VIEW of the pinned chunk buffer (valid during the call only).  Returns (qoff, total)."""
        ichr, qs, qe = _i32(ichr), _i32(qs), _i32(qe)
        nq = len(qs)
        qoff = np.zeros(nq + 1, np.int64)
        total = C.c_int64(0)

        def sink(ctx, q0, q1, qoff_p, hits_p):
            if on_chunk is not None:
                n = int(qoff[q1] - qoff[q0])
                rec = (np.ctypeslib.as_array(C.cast(hits_p, N.i32p), shape=(n * 4,)).reshape(n, 4)
                       if n else np.zeros((0, 4), np.int32))
                on_chunk(int(q0), int(q1), qoff, rec)
            return 0

        cb = N.ENUM_SINK(sink)
        _chk(self._H.igd_hip_enumerate_stream(self.dev, ichr.ctypes.data, qs.ctypes.data, qe.ctypes.data, nq,
                                              qoff.ctypes.data, cb, None, C.byref(total)), "igd_hip_enumerate_stream")
        return qoff, total.value

    def hit8_idx_bits(self):
        """How the packed `-f` record (igd_hip_hit8: 8 bytes per overlap) splits its second word for this database -- the low
        `bits` hold idx, the rest end - start -- or -1 when a record does not fit (igd_hip_hit8_idx_bits)."""
        return int(self._H.igd_hip_hit8_idx_bits(self.dev))

    def enumerate_stream8(self, ichr, qs, qe, on_chunk=None):
        """`-f`, streamed in 8 bytes per overlap (igd_hip_enumerate_stream8): on_chunk(q0, q1, qoff, rec, bits) is called per chunk
        with rec = uint32[n,2] VIEW of the pinned chunk buffer: rec[:,0] = start, rec[:,1] = (end - start) << bits | idx.
        Returns (qoff, total)."""
        ichr, qs, qe = _i32(ichr), _i32(qs), _i32(qe)
        nq = len(qs)
        qoff = np.zeros(nq + 1, np.int64)
        total = C.c_int64(0)

        def sink(ctx, q0, q1, qoff_p, hits_p, bits):
            if on_chunk is not None:
                n = int(qoff[q1] - qoff[q0])
                rec = (np.ctypeslib.as_array(C.cast(hits_p, C.POINTER(C.c_uint32)), shape=(n * 2,)).reshape(n, 2)
                       if n else np.zeros((0, 2), np.uint32))
                on_chunk(int(q0), int(q1), qoff, rec, int(bits))
            return 0

        cb = N.ENUM_SINK8(sink)
        _chk(self._H.igd_hip_enumerate_stream8(self.dev, ichr.ctypes.data, qs.ctypes.data, qe.ctypes.data, nq,
                                               qoff.ctypes.data, cb, None, C.byref(total)), "igd_hip_enumerate_stream8")
        return qoff, total.value

    @staticmethod
    def expand_hit8(rec, bits):
        """(start, end, idx) int32 arrays of packed records (igd_hip_hit8_expand)."""
        start = rec[:, 0].astype(np.uint32)
        hi = rec[:, 1].astype(np.uint32)
        idx = (hi & np.uint32((1 << bits) - 1)) if bits else np.zeros(len(hi), np.uint32)
        end = (start + (hi >> np.uint32(bits))).astype(np.uint32)
        return start.view(np.int32), end.view(np.int32), idx.astype(np.int32)

    def hitmap(self, v=0):
        """`-m`: (uint32[nfiles,nfiles], pairs); v>0 keeps records with value > v (getMap_v)."""
        m = np.zeros((self.nfiles, self.nfiles), np.uint32)
        total = C.c_int64(0)
        _chk(self._H.igd_hip_hitmap(self.dev, 1 if v > 0 else 0, int(v), m.ctypes.data, C.byref(total)), "igd_hip_hitmap")
        return m, total.value

    def seqpare(self, ichr, qs, qe, qgroup, ngroups, n_queries_total=None, nr=None):
        """`-s` (Seqpare, seqOverlaps src/igd_search.c:354-451).  Queries in the reference's order: contigs
        of the query file in first-seen order (qgroup = 0,1,.. non-decreasing), inside a contig by start
        (stable).  Returns the per-dataset sums of matched similarities; with n_queries_total (all accepted
        query lines, known contig or not) and nr (regions per dataset) the similarity S = sum/(Nq+nr-sum)."""
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in (ichr, qs, qe, qgroup)]
        sums = np.zeros(max(self.nfiles, 1), np.float64)
        _chk(self._H.igd_hip_seqpare(self.dev, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, len(a[0]),
                                     a[3].ctypes.data, int(ngroups), sums.ctypes.data), "igd_hip_seqpare")
        sums = sums[:self.nfiles]
        if n_queries_total is None or nr is None:
            return sums
        return sums / (float(n_queries_total) + np.asarray(nr, np.float64) - sums)

    def batch_stats(self, d_ichr, d_qs, d_qe, nq, v=0):
        rule, vf = self.cli_dispatch(self.gtype, v)
        st = N.HipStats()
        _chk(self._H.igd_hip_batch_stats(self.dev, d_ichr, d_qs, d_qe, int(nq), vf, rule, C.byref(st)),
             "igd_hip_batch_stats")
        return dict(queries=st.queries, pairs=st.pairs, S=st.S, B=st.B, H=st.H)

    def batch_traffic(self, d_ichr, d_qs, d_qe, nq, v=0, flags=0):
        """Compulsory HBM bytes of the scan kernel for this batch (igd_hip_batch_traffic)."""
        rule, vf = self.cli_dispatch(self.gtype, v)
        t = N.HipTraffic()
        _chk(self._H.igd_hip_batch_traffic(self.dev, d_ichr, d_qs, d_qe, int(nq), vf, rule, int(flags), C.byref(t)),
             "igd_hip_batch_traffic")
        return {k: getattr(t, k) for k, _ in N.HipTraffic._fields_}

    def algorithmic_bytes(self, stats, nq, mode="hits"):
        """bytes one launch has to touch, by the reference's own work terms."""
        b = 4 * stats["S"] + 4 * stats["H"] + 4 * stats["B"] + 16 * stats["pairs"] + 12 * nq + 8 * self.nfiles
        if mode == "v":
            b += 4 * stats["S"]
        elif mode == "f":
            b += 4 * stats["H"] + 16 * stats["H"] + 8 * nq
        return b

    def profile_begin(self, max_launches, every=1):
        """Arm HIP-event timing of the scan kernel for up to max_launches launches, every `every`-th one."""
        _chk(self._H.igd_hip_profile_sampling(self.dev, int(every)), "igd_hip_profile_sampling")
        _chk(self._H.igd_hip_profile_begin(self.dev, int(max_launches)), "igd_hip_profile_begin")

    def profile_end(self):
        n, a, b = C.c_int(0), C.c_double(0), C.c_double(0)
        _chk(self._H.igd_hip_profile_end(self.dev, C.byref(n), C.byref(a), C.byref(b)), "igd_hip_profile_end")
        return dict(launches=n.value, scan_ms=a.value, pipeline_ms=b.value)

    def last_scan_kernel(self):
        """Name of the scan kernel the last batch ran on ("igd_scan_sorted" / "igd_scan_tiles"); waits for it."""
        return (self._H.igd_hip_last_scan_kernel(self.dev) or b"").decode()


def measure_rates(device=0):
    """GB/s of this box, measured now: HBM float4 copy (read+write), HBM float4 read, pinned D2H, pinned H2D."""
    r = (C.c_double * 4)()
    _chk(N.hip().igd_hip_measure_rates(int(device), r), "igd_hip_measure_rates")
    return {"hbm_copy_GBps": r[0], "hbm_read_GBps": r[1], "d2h_GBps": r[2], "h2d_GBps": r[3]}
