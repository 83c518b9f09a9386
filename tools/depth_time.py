"""Experiment: time of a 1080p _Depth dispatch on the bench scene."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pa
scene = pa.Scene("hall260k"); sky = pa.procedural_sky(256, 128)
W, H = 1920, 1080
t = pa.PathTracer(0); t.load_scene(scene, sky); t.resize(W, H)
p = pa.make_params(scene, W, H, depth=1)
for _ in range(3): t.render(p)
t.sync(); t.stats()
t0 = time.perf_counter()
for _ in range(20): t.render(p)
t.sync(); dt = (time.perf_counter() - t0) / 20
st = t.stats()
print("depth dispatch ms", round(dt * 1e3, 3), "Grays/s", round(st["rays"] / 20 / dt / 1e9, 2), "nodes/ray", round(st["nodes_visited"] / st["rays"], 2))
