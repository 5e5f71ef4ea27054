import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/liborc.so), built on demand."""
    from oracle import orc as _orc
    _orc.build()
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def hrt_lib():
    """libhip_raytrace.so (built by __graft_entry__.build()); tests fail loudly if it is missing."""
    from ilgpu_raytracing_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return engine.lib()


@pytest.fixture(scope="session")
def hooks_lib(hrt_lib):
    """libhip_raytrace_test.so: the same sources with the test hooks of include/hrt_test_hooks.h compiled in."""
    from ilgpu_raytracing_amd import engine
    return engine.hooks()


@pytest.fixture(scope="session")
def hooks_renderer(hooks_lib):
    """An RTRenderer on the hooks build (math probes, treelet limits); product behaviour is tested on `renderer`."""
    from ilgpu_raytracing_amd import engine
    r = engine.RTRenderer([0], library=hooks_lib)
    yield r
    r.close()


@pytest.fixture(scope="session")
def renderer(hrt_lib):
    """One RTRenderer (one hrt_ctx on device 0) shared by the GPU tests."""
    from ilgpu_raytracing_amd import engine
    r = engine.RTRenderer([0])
    yield r
    r.close()
