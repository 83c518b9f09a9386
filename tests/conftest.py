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


@pytest.fixture(scope="session", params=["sched-bvh4", "sched-bvh2", "sched-bvh8", "pool-bvh4", "pool-bvh2"])
def tracer(request):
    """One libptrace context on cuda:0 per implementation variant: the scheduled persistent kernel (default) over the
    four-wide, the binary and the eight-wide BVH, and the kernel with a ray pool per wave.  All must match the oracle.
    (The wavefront stages, three times slower and bit-identical in every test of two rounds, keep a golden-vector test, a
    configuration test and the regression tests that loop over the backends.)"""
    import pathtracing_amd as pa
    t = pa.PathTracer(0)
    backend, width = request.param.split("-bvh")
    t.set_option("backend", {"sched": 0, "wavefront": 1, "pool": 2}[backend])
    t.set_option("bvh_width", int(width))
    t.variant = request.param
    yield t
    t.close()
