"""The reference's UNCHANGED Cython wrapper (src_py/igd_py.pyx) compiles and links against
libigd_py.so with only setup.py changed, as INTEGRATION.md says.  Needs /root/reference (build
container); the .pyx is read from there at test time, never copied into the repo."""
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
    finally:
        shutil.rmtree(d, ignore_errors=True)
