import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pathtracing_amd as pa
scene = pa.Scene(sys.argv[1] if len(sys.argv) > 1 else "city10m"); sky = pa.procedural_sky(64, 32)
with pa.PathTracer(0) as tr:
    tr.set_option("instancing", 0); tr.set_option("tree_build", int(os.environ.get("BUILD", "1")))
    tr.load_scene(scene, sky); tr.resize(64, 36)
    p = pa.make_params(scene, 64, 36, seed=0.5, spp=1, max_bounces=2)
    tr.render(p); tr.sync()
    print("build ms", tr.stats()["bvh_build_ms"], flush=True)
