import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The checker (oracle/) is built here when a compiler is around and it is stale; the bindings themselves never build."""
    import shutil
    import oracle_binding as ob
    src = os.path.join(ROOT, "oracle", "pcsf_oracle.c")
    stale = not os.path.exists(ob.LIB) or os.path.getmtime(ob.LIB) < os.path.getmtime(src)
    if stale and shutil.which("make") and shutil.which("gcc"):
        ob.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _oracle_model(name):
    import oracle_binding as ob
    return ob.OracleModel(os.path.join(GOLDEN, "models", name))


@pytest.fixture(scope="session")
def oracle_models():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = _oracle_model(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def gpu_contexts():
    """ScoringContext per model name on cuda:0 (through the C-ABI); model arrays come from the files."""
    import phylocsfpp_amd as pa
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = pa.load_model(os.path.join(GOLDEN, "models", name)).context(device=0)  # the package's own reader
        return cache[name]
    yield get
    for c in cache.values():
        c.close()
