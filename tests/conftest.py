import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored: on a fresh checkout build them once (hipcc cross-compiles without a GPU, gcc
    builds the oracle) -- the same thing __graft_entry__.build() does.  Never a fallback: a failed build fails the tests."""
    # both builders compare mtimes and do nothing when the binaries are newer than their sources, so calling them every
    # session costs nothing and an edited .hip / .c can never be tested against a stale binary
    from handmvnet_amd.build import build as build_engine
    build_engine(verbose=False)
    from oracle.oracle import build as build_oracle
    build_oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
