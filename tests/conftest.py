import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_pkg()


_CORPORA = {}


def corpus(**kw):
    """Session cache of synthetic corpora keyed by their parameters."""
    import synth
    key = tuple(sorted(kw.items()))
    if key not in _CORPORA:
        _CORPORA[key] = synth.make_corpus(**kw)
    return _CORPORA[key]


@pytest.fixture(scope="session")
def gpu(pkg):
    """A device handle factory; fails loudly (no skip, no fallback) when the HIP path is unusable."""
    made = []

    def make():
        g = pkg.GpuIndex(0)
        made.append(g)
        return g

    yield make
    for g in made:
        g.close()
