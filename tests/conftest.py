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
