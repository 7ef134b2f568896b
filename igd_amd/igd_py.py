"""`igd_py` -- the class of the reference's Cython wrapper (/root/reference/src_py/igd_py.pyx:21-44)
with the same methods and argument meaning, bound with ctypes to libigd_py.so
(include/igd_py_abi.h).  INTEGRATION.md shows the unchanged .pyx compiling against the same
library; this ctypes twin exists so the parity tests can run without a Cython build step."""
import numpy as np

from . import _native as N


class igd_py:
    def __init__(self):
        self._L = N.pyabi()
        self._h = self._L.iGD_init()

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
        # the C side appends "/" and "*" in place (src_py/igd_create.c:22-31): give it room
        i = C.create_string_buffer(str.encode(iPath), len(iPath) + 8)
        o = C.create_string_buffer(str.encode(oPath), len(oPath) + 8)
        self._L.create_iGD(self._h, i, o, str.encode(igdName), int(bin_size))

    def open(self, igdFile):
        self._L.open_iGD(self._h, str.encode(igdFile))

    def search_1(self, chrm, qs, qe, hits):
        assert hits.dtype == np.int64 and hits.flags["C_CONTIGUOUS"] and hits.ndim == 1
        self._L.get_overlaps(self._h, str.encode(chrm), int(qs), int(qe),
                             hits.ctypes.data_as(N.i64p))

    def search_n(self, qFile, hits):
        assert hits.dtype == np.int64 and hits.flags["C_CONTIGUOUS"] and hits.ndim == 1
        return self._L.getOverlaps(self._h, str.encode(qFile), hits.ctypes.data_as(N.i64p))
