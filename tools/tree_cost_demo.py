"""Surface-area cost of the four-wide tree as the host builds it (tree_build 0), as the device builds it top-down (1) and by
clustering (3): the sum of node areas over the root's area, printed by the library under PTRACE_BUILD_TIMING after a refit.
usage: PTRACE_BUILD_TIMING=1 python tools/tree_cost_demo.py scene ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pa
for name in sys.argv[1:]:
    scene = pa.Scene(name); sky = pa.procedural_sky(64, 32)
    for build in (0, 1, 3):
        with pa.PathTracer(0) as tr:
            tr.set_option("instancing", 0); tr.set_option("tree_build", build)
            tr.load_scene(scene, sky); tr.resize(64, 36)
            p = pa.make_params(scene, 64, 36, seed=0.5, spp=1, max_bounces=2)
            tr.render(p); tr.sync()
            print(name, "tree_build", build, flush=True)
            tr.update_transforms(scene.objects.copy()); tr.sync()
