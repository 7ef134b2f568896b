#!/usr/bin/env python3
"""oracle/build_ref_pyx.py -- TEST INFRASTRUCTURE (see oracle/Makefile).

Builds the reference's UNCHANGED Cython wrapper (/root/reference/src_py/igd_py.pyx, read where it lies)
against this repository's libigd_py.so, exactly as INTEGRATION.md section 2 tells a maintainer to: only
setup.py differs (no C sources, include/pyabi, -ligd_py -ligd_hip).  Everything intermediate (the copy
Cython works on, the generated C) stays in a temporary directory that is removed; only the compiled
extension module lands in oracle/_ref/pyx/ (git-ignored, travels to the GPU box like oracle/_ref/igd),
where tests/test_gpu_golden.py imports it and replays the reference's src_py/igd_test.py calls."""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PYX = os.environ.get("IGD_REF_PYX", "/root/reference/src_py/igd_py.pyx")


def main():
    if not os.path.exists(PYX):
        print("reference pyx %s absent: keeping prebuilt oracle/_ref/pyx (if any)" % PYX)
        return 0
    try:
        import Cython  # noqa: F401
        import numpy
    except Exception as e:
        print("Cython/numpy not importable (%s): skipping the reference pyx build" % e)
        return 0
    lib = os.path.join(ROOT, "igd_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libigd_py.so")):
        print("igd_amd/lib/libigd_py.so not built yet: skipping the reference pyx build")
        return 0
    d = tempfile.mkdtemp(prefix="igdpyx", dir="/tmp")
    try:
        shutil.copy(PYX, os.path.join(d, "igd_py.pyx"))
        open(os.path.join(d, "setup.py"), "w").write(
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "ext = Extension('igd_py', sources=['igd_py.pyx'], include_dirs=[%r, %r],\n"
            "                library_dirs=[%r], libraries=['igd_py', 'igd_hip'],\n"
            "                runtime_library_dirs=['$ORIGIN/../../../igd_amd/lib'])\n"
            "setup(ext_modules=cythonize([ext], language_level=3))\n"
            % (numpy.get_include(), os.path.join(ROOT, "include", "pyabi"), lib))
        subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=d,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = os.path.join(ROOT, "oracle", "_ref", "pyx")
        os.makedirs(out, exist_ok=True)
        so = glob.glob(os.path.join(d, "igd_py*.so"))
        if not so:
            print("reference pyx: no extension module produced")
            return 1
        for f in glob.glob(os.path.join(out, "igd_py*.so")):
            os.unlink(f)
        shutil.copy(so[0], out)
        print("built oracle/_ref/pyx/%s: the reference's Cython wrapper on libigd_py.so" % os.path.basename(so[0]))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
