"""`igd_py` -- the class of the reference's Cython wrapper (/root/reference/src_py/igd_py.pyx:21-44)
with the same methods and argument meaning, bound with ctypes to libigd_py.so
(include/igd_py_abi.h).  INTEGRATION.md shows the unchanged .pyx compiling against the same
library; this ctypes twin exists so the parity tests can run without a Cython build step."""
import numpy as np

from . import _native as N


class IgdEngineError(RuntimeError):
    """The GPU engine behind libigd_py.so could not be used (the library itself only reports it)."""


class igd_py:
    def __init__(self):
        self._L = N.pyabi()
        self._h = self._L.iGD_init()

    def _check(self, what):
        rc = self._L.igd_engine_status()
        if rc:
            self._L.igd_engine_clear()
            raise IgdEngineError("%s: GPU engine failure (code %d): %s -- there is no CPU search path"
                                 % (what, rc, N.hip().igd_hip_last_error().decode()))

    def __del__(self):
        try:
            if self._h:
                self._L.close_iGD(self._h)
                self._h = None
        except Exception:
            pass

    def get_nFiles(self):
        return self._L.get_nFiles(self._h)

    def create(self, iPath, oPath, igdName, bin_size):
        import ctypes as C
        # exactly what the reference's .pyx passes: immutable bytes objects (the C side copies them)
        self._L.create_iGD(self._h, str.encode(iPath), str.encode(oPath), str.encode(igdName), int(bin_size))
        self._check("create")

    def open(self, igdFile):
        self._L.open_iGD(self._h, str.encode(igdFile))
        self._check("open")

    def search_1(self, chrm, qs, qe, hits):
        assert hits.dtype == np.int64 and hits.flags["C_CONTIGUOUS"] and hits.ndim == 1
        self._L.get_overlaps(self._h, str.encode(chrm), int(qs), int(qe),
                             hits.ctypes.data_as(N.i64p))
        self._check("search_1")

    def search_n(self, qFile, hits):
        assert hits.dtype == np.int64 and hits.flags["C_CONTIGUOUS"] and hits.ndim == 1
        n = self._L.getOverlaps(self._h, str.encode(qFile), hits.ctypes.data_as(N.i64p))
        self._check("search_n")
        return n
