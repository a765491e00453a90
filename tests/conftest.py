import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built():
    """Build librt_amd.so (hipcc cross-compiles gfx950 without a GPU) and the oracle once."""
    import __graft_entry__ as ge
    ge.build()
    return ge


@pytest.fixture(scope="session")
def renderer(built):
    """One device context for all GPU tests (single process, single GPU)."""
    from ray_tracer_amd import engine
    r = engine.Renderer(0)
    # soak runs: RT_TEST_TUNE="lanes_min_kslots=1,object_tree_min=2" puts every test's small scenes through code that only big
    # ones reach by default (the dispatch in parts, the object hierarchy); RT_RANDOM_SEEDS=N widens the seeded random-scene tests
    for kv in filter(None, os.environ.get("RT_TEST_TUNE", "").split(",")):
        k, v = kv.split("=")
        r.set_tuning(k, int(v))
    yield r
    r.close()
