import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pathtracing_amd as pa
W, H = 1920, 1080
scene = pa.Scene("hall260k"); sky = pa.procedural_sky(2048, 1024)
out = {}
for backend in (0, 1):
    tr = pa.PathTracer(0); tr.set_option("backend", backend); tr.load_scene(scene, sky); tr.resize(W, H)
    frames = []
    for mode in ("sync", "async"):
        for k in range(4):
            p = pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=1, max_bounces=8, accumulate=pa.PT_ACCUM_SUM, sample_index=k)
            tr.render(p)
            if mode == "sync":
                frames.append(tr.read_result().copy())
        if mode == "async":
            frames.append(tr.read_converged().copy())
        else:
            frames.append(tr.read_converged().copy())
    out[backend] = frames
    tr.close()
for i, (a, b) in enumerate(zip(out[0], out[1])):
    neq = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
    print("item", i, "differing pixels", int(neq.sum()), "means", float(a[..., :3].mean()), float(b[..., :3].mean()))
    if neq.any():
        ys, xs = np.nonzero(neq); print("  first", ys[:5], xs[:5], a[ys[0], xs[0]], b[ys[0], xs[0]])
