import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the reference binary oracle/_ref/igd")


def pytest_collection_modifyitems(config, items):
    from helpers import have_ref
    skip_ref = pytest.mark.skip(reason="oracle/_ref/igd (reference binary) not present")
    for item in items:
        if "ref" in item.keywords and not have_ref():
            item.add_marker(skip_ref)


def pytest_sessionstart(session):
    """Build the native libraries / oracle when the tree was copied without them."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    need = [os.path.join(root, "igd_amd", "lib", n) for n in
            ("libigd_hip.so", "libigd.so", "libigd_py.so", "libigdr.so", "libigd_synth.so")]
    need += [os.path.join(root, "bin", "igd"), os.path.join(root, "bin", "igd_synth"),
             os.path.join(root, "oracle", "_build", "liboracle.so"), os.path.join(root, "oracle", "_build", "igd_oracle")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-s", "-C", root, "all"], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")], stdout=subprocess.DEVNULL)
