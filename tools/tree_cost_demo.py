import os, sys
sys.path.insert(0, '/root/repo')
import pathtracing_amd as pa
for name in sys.argv[1:]:
    scene = pa.Scene(name); sky = pa.procedural_sky(64, 32)
    for build in (0, 1):
        with pa.PathTracer(0) as tr:
            tr.set_option("instancing", 0); tr.set_option("tree_build", build)
            tr.load_scene(scene, sky); tr.resize(64, 36)
            p = pa.make_params(scene, 64, 36, seed=0.5, spp=1, max_bounces=2)
            tr.render(p); tr.sync()
            print(name, "tree_build", build, flush=True)
            tr.update_transforms(scene.objects.copy()); tr.sync()
