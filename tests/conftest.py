import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the reference binary oracle/_ref/igd")
    config.addinivalue_line("markers", "hostpath: a gpu test that keeps the default IGD_HOST_MAX_QUERIES (small files on the host)")


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


@pytest.fixture(autouse=True)
def _gpu_tests_send_every_query_file_to_the_engine(request, monkeypatch):
    """Query files of at most IGD_HOST_MAX_QUERIES lines (default: 25 000 per usable host thread -- igdc_host_limit, igdc_host_limit_enum) are counted on the host by the flavours' file
    entry points (igd_hostpath.c: the reference's cheap start for small jobs).  The `-m gpu` tests are the parity tests of
    the HIP path and their fixtures are small, so they run with the limit at 0: every file goes to the engine.  A GPU test
    that wants the product's default behaviour carries the marker `hostpath`."""
    if request.node.get_closest_marker("gpu") and not request.node.get_closest_marker("hostpath"):
        monkeypatch.setenv("IGD_HOST_MAX_QUERIES", "0")
    yield
