import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def sky_small():
    import pathtracing_amd as pa
    return pa.procedural_sky(128, 64)


@pytest.fixture(scope="session", params=["wavefront", "sched", "mega_old"])
def tracer(request):
    """One libptrace context on cuda:0 per backend: the wavefront stages (default) and the
    single persistent kernel must both match the oracle."""
    import pathtracing_amd as pa
    t = pa.PathTracer(0)
    t.set_option("backend", {"sched": 0, "wavefront": 1, "mega_old": 2}[request.param])
    yield t
    t.close()
