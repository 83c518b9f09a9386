"""The randomised soaks of tools/soak_*.py as short, fixed-seed GPU tests: every frame / dispatch / scene they draw is
compared bit for bit with the oracle (soak_build: with the host-built tree's image and the exhaustive ray query).  Longer runs with other seeds: python tools/soak_<name>.py [seconds] [seed]."""
import importlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _soak(name):
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    return importlib.import_module("soak_" + name)


@pytest.mark.parametrize("name,seconds,seed,at_least", [
    ("parity", 8.0, 11, 100),        # random scene / size / renderMode sequences
    ("params", 6.0, 12, 2000),       # dispatch parameters: 1x1 images, tiles, zero samples, accumulation, odd cameras
    ("geometry", 6.0, 13, 1000),     # degenerate, coincident, NaN / infinite primitives; build and refit
    ("session", 6.0, 14, 5000),      # one context through mode switches, resizes, scene changes, refits
    ("master", 6.0, 15, 5000),       # the C++ RayTracingMaster mirror, camera free to wander
    ("build", 8.0, 16, 15),          # host builder against both device builders on scenes of 150 .. 90 000 triangles (host-built image as the reference)
])
def test_soak(name, seconds, seed, at_least, monkeypatch):
    monkeypatch.setenv("SOAK_SCENES", "fixture,xr")
    n = _soak(name).run(seconds, seed)
    assert n >= at_least, f"only {n} items checked in {seconds} s"
