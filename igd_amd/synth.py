"""Deterministic synthetic databases / query sets (tools/igd_synth.c) for tests and bench.py."""
import numpy as np

from . import _native as N

HG38, SMALL = 0, 1


def make_db(path, files=1900, per_file=26316, seed=1000, nbp_log=14, genome=HG38, len_mode=None,
            len_a=50, len_b=30000, clustered=False, gtype=1):
    if len_mode is None:
        len_mode = 1 if genome == SMALL else 0
    rc = N.synth().igd_synth_db(path.encode(), files, per_file, seed, nbp_log, genome, len_mode,
                                len_a, len_b, 1 if clustered else 0, gtype)
    if rc != 0:
        raise RuntimeError("igd_synth_db failed for %s" % path)


def make_db_beds(out_dir, files, per_file, seed=1000, genome=SMALL, len_mode=None, len_a=50, len_b=30000):
    if len_mode is None:
        len_mode = 1 if genome == SMALL else 0
    if N.synth().igd_synth_db_beds(out_dir.encode(), files, per_file, seed, genome, len_mode, len_a, len_b) != 0:
        raise RuntimeError("igd_synth_db_beds failed")


def make_queries(n, seed=7, genome=HG38, min_len=100, max_len=1999, sorted_=True, unknown_every=0,
                 extra_span=0):
    ichr = np.empty(n, np.int32)
    qs = np.empty(n, np.int32)
    qe = np.empty(n, np.int32)
    N.synth().igd_synth_queries(n, seed, genome, min_len, max_len, 1 if sorted_ else 0, unknown_every,
                                extra_span, ichr.ctypes.data, qs.ctypes.data, qe.ctypes.data)
    return ichr, qs, qe


def make_queries_slab(n_total, lo, hi, seed=7, genome=HG38, min_len=100, max_len=1999):
    """Queries [lo, hi) of make_queries(n_total, seed, sorted_=True) without building the rest (config 4:
    rank r of N takes slab r of ONE position-sorted set of N x 1.25e7 queries)."""
    m = int(hi) - int(lo)
    ichr = np.empty(m, np.int32)
    qs = np.empty(m, np.int32)
    qe = np.empty(m, np.int32)
    got = N.synth().igd_synth_queries_slab(int(n_total), seed, genome, min_len, max_len, int(lo), int(hi),
                                           ichr.ctypes.data, qs.ctypes.data, qe.ctypes.data)
    if got != m:
        raise RuntimeError("igd_synth_queries_slab failed")
    return ichr, qs, qe


def write_bed(path, genome, ichr, qs, qe):
    ichr, qs, qe = (np.ascontiguousarray(a, np.int32) for a in (ichr, qs, qe))
    if N.synth().igd_synth_write_bed(path.encode(), genome, len(qs), ichr.ctypes.data, qs.ctypes.data,
                                     qe.ctypes.data) != 0:
        raise IOError(path)


def contig_names(genome):
    L = N.synth()
    return [L.igd_synth_contig_name(genome, i).decode() for i in range(L.igd_synth_ncontigs(genome))]
