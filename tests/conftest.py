import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
PKG = os.path.join(REPO, 'practical-collab-perception_amd')
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture
def lib_option():
    """set (name, value) / restore (name, None) an option of the library's table (pcp_set_option); everything touched is restored afterwards"""
    from pcp_amd import lib
    touched = {}

    def setter(name, value):
        prev = lib.set_option(name, value)
        touched.setdefault(name, prev)
    yield setter
    for name, prev in touched.items():
        lib.set_option(name, prev)
