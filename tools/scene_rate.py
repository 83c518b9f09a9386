"""Headline-settings rate (1920x1080, 8 bounces, 16 spp, one dispatch at a time) on a named scene;
a second argument "ggx" gives every mesh object a GGX copper surface (the extended kernel)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pa
name = sys.argv[1] if len(sys.argv) > 1 else "blob70k"
scene = pa.Scene(name); sky = pa.procedural_sky(2048, 1024)
W, H = 1920, 1080
t = pa.PathTracer(0); t.load_scene(scene, sky); t.resize(W, H)
if len(sys.argv) > 2 and sys.argv[2] == "ggx":
    t.set_mesh_materials(pa.materials([((0.0, 0.0, 0.0), 0.8, (0.95, 0.64, 0.54), pa.PT_LOBE_GGX)] * len(scene.objects)))
t.render(pa.make_params(scene, W, H, seed=0.1, spp=16, max_bounces=8)); t.sync(); t.stats()
t0 = time.perf_counter()
for k in range(5):
    t.render(pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=16, max_bounces=8))
t.sync(); dt = (time.perf_counter() - t0) / 5
st = t.stats()
print(name, "ms/frame", round(dt * 1e3, 3), "rays/frame", st["rays"] // 5, "Mrays/s", round(st["rays"] / 5 / dt / 1e6, 1), "nodes/ray", round(st["nodes_visited"] / st["rays"], 2), flush=True)
