"""The reference's UNCHANGED Cython wrapper (src_py/igd_py.pyx) compiles and links against
libigd_py.so with only setup.py changed, as INTEGRATION.md says.  Needs /root/reference (build
container, CPU only); the .pyx is read from there at test time and everything built from it lives and
dies in a directory under /tmp: nothing of the reference's wrapper -- source, generated C or compiled
module -- enters the repository or travels to the GPU box.  What the wrapper's batch call returns on the
GPU is pinned through the ctypes class against the same tests/golden/pywrap.json (tests/test_gpu_golden.py)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

from helpers import ROOT, short_tmpdir

PYX = "/root/reference/src_py/igd_py.pyx"


@pytest.mark.skipif(not os.path.exists(PYX), reason="reference tree not present")
def test_unchanged_pyx_builds_against_libigd_py():
    pytest.importorskip("Cython")
    import numpy
    d = short_tmpdir("igx")
    try:
        shutil.copy(PYX, os.path.join(d, "igd_py.pyx"))
        lib = os.path.join(ROOT, "igd_amd", "lib")
        open(os.path.join(d, "setup.py"), "w").write(
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "ext = Extension('igd_py', sources=['igd_py.pyx'], include_dirs=[%r, %r],\n"
            "                library_dirs=[%r], libraries=['igd_py', 'igd_hip'], runtime_library_dirs=[%r])\n"
            "setup(ext_modules=cythonize([ext], language_level=3))\n"
            % (numpy.get_include(), os.path.join(ROOT, "include", "pyabi"), lib, lib))
        subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=d,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        code = ("import sys; sys.path.insert(0, %r); import igd_py; g = igd_py.igd_py(); "
                "print('nFiles', g.get_nFiles()); del g" % d)
        out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, check=True).stdout.decode()
        assert "nFiles 0" in out          # a never-opened handle: 0 files, and dropping it is safe
        # open / get_nFiles / search_1 (answered on the host from the interval's own tiles: no GPU needed) return
        # what the same wrapper returned on the reference's own C code (tests/golden/pywrap.json)
        pins = json.load(open(os.path.join(ROOT, "tests", "golden", "pywrap.json")))
        code = ("import sys, json; sys.path.insert(0, %r); import numpy as np; import igd_py as iGD\n"
                "igd = iGD.igd_py(); igd.open(%r); n = igd.get_nFiles(); out = {'nFiles': n, 'search_1': {}}\n"
                "for key in %r:\n"
                "    c, rng = key.split(':'); s, e = rng.split('-'); v = np.zeros(n, dtype='int64')\n"
                "    igd.search_1(c, int(s), int(e), v); out['search_1'][key] = v.tolist()\n"
                "print(json.dumps(out))\n"
                % (d, os.path.join(ROOT, "tests", "golden", "smallrand", "db.igd"), list(pins["search_1"].keys())))
        p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-800:]
        got = json.loads(p.stdout.decode().strip().splitlines()[-1])
        assert got["nFiles"] == pins["nFiles"] and got["search_1"] == pins["search_1"]
    finally:
        shutil.rmtree(d, ignore_errors=True)
