"""Randomised session soak (run on the GPU box): ONE context lives through a long random sequence of what a host does -
frames in every renderMode with the camera lists kept as RayTracingMaster.cs:492-526 keeps them, mode and stripe-count
switches in mid-cycle, resizes, scene changes, transform updates (refits), depth dispatches - while a model of the three
targets and of the shader's camera arrays is advanced with the oracle.  After every frame Result, Result1 and Result2
must equal the model's, bit for bit (any NaN equal to any NaN).
usage: python tools/soak_session.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as pa  # noqa: E402
from oracle import pt_oracle as po  # noqa: E402



def diagnose(t, p, old, before, gpu, model, name, objs, W, H, frames):
    """After a mismatch: keep the inputs, then render the frame again from the same targets - as it was, on the two-child
    tree and on the other kernels - to tell a race from a wrong hit and a kernel from the tree."""
    import ctypes
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "soak_session_failure.npz"), params=np.frombuffer(ctypes.string_at(ctypes.byref(p), ctypes.sizeof(p)), np.uint8),
             cams=np.frombuffer(b"".join(ctypes.string_at(ctypes.byref(c), ctypes.sizeof(c)) for c in old), np.uint8), n_cams=len(old),
             objs=objs, scene=name, W=W, H=H, frame=frames, before=np.stack(before), gpu=np.stack(gpu), model=np.stack(model))

    def again(label):
        t.write_target(pa.PT_TARGET_RESULT, before[0]); t.write_converged(before[1]); t.write_target(pa.PT_TARGET_DENOISED, before[2])
        t.set_camera_history(old)
        t.render(p)
        out = (t.read_result(), t.read_converged(), t.read_target(pa.PT_TARGET_DENOISED))
        eq_first = all((a.view(np.uint32) == b.view(np.uint32)).all() for a, b in zip(out, gpu))
        eq_model = all(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all() for a, b in zip(out, model))
        print("  again,", label, ": equals the first GPU frame", eq_first, ", equals the model", eq_model, flush=True)

    again("unchanged")
    for key, val, label in (("bvh_width", 2, "two-child tree"), ("bvh_width", 8, "eight-child tree"), ("bvh_width", 4, "four-child tree again"), ("backend", 1, "wavefront stages"),
                            ("backend", 2, "ray pool"), ("backend", 0, "scheduled kernel again")):
        try:
            t.set_option(key, val)
            again(label)
        except Exception as e:       # an option a build does not have must not hide the mismatch itself
            print("  ", label, "not run:", e, flush=True)


def run(budget=60.0, seed=1):
    """One soak of `budget` seconds from `seed`; raises AssertionError on the first mismatch, returns the count of checked items."""
    rng = np.random.default_rng(seed)
    sky = pa.procedural_sky(64, 32)
    scenes = {n: pa.Scene(n) for n in ("fixture", "xr")}
    MODES = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12]


    def same(a, b):
        return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


    t = pa.PathTracer(0)
    name = "fixture"; scene = scenes[name]; objs = np.array(scene.objects).copy()
    orc = po.Oracle(scene.spheres, objs, scene.vertices, scene.indices, sky=sky, accel=True)
    t.load_scene(scene, sky)
    W, H = 40, 24
    t.resize(W, H)
    res = np.zeros((H, W, 4), np.float32); res1 = np.zeros_like(res); res2 = np.zeros_like(res)
    cams = [pa.PtCameraHistory() for _ in range(100)]      # the shader's four uniform arrays
    old = []                                               # the host's lists
    mode, divisions, counter = 2, 3, 0
    eye = np.array(scene.eye, np.float64)
    t0 = time.time(); frames = 0; lit = 0; events = {"resize": 0, "scene": 0, "refit": 0, "mode": 0}
    while time.time() - t0 < budget:
        r = rng.random()
        if r < 0.03:
            W = int(rng.integers(3, 80)); H = int(rng.integers(3, 50))
            t.resize(W, H)                                  # InitRenderTexture: new, zero-filled targets
            res = np.zeros((H, W, 4), np.float32); res1 = np.zeros_like(res); res2 = np.zeros_like(res)
            events["resize"] += 1
        elif r < 0.06:
            name = "xr" if name == "fixture" else "fixture"; scene = scenes[name]; objs = np.array(scene.objects).copy()
            t.load_scene(scene, sky)
            orc = po.Oracle(scene.spheres, objs, scene.vertices, scene.indices, sky=sky, accel=True)
            eye = np.array(scene.eye, np.float64)
            events["scene"] += 1
        elif r < 0.16 and len(objs):
            objs["m"][:, 12:15] += rng.uniform(-0.2, 0.2, (len(objs), 3)).astype(np.float32)
            t.update_transforms(objs)
            orc = po.Oracle(scene.spheres, objs, scene.vertices, scene.indices, sky=sky, accel=True)
            events["refit"] += 1
        if rng.random() < 0.2:
            mode = int(rng.choice(MODES)); events["mode"] += 1
        if rng.random() < 0.05:
            divisions = int(rng.integers(1, 6))
        eye = eye + rng.uniform(-0.15, 0.15, 3)
        depth = int(rng.random() < 0.05)
        p = pa.make_params(scene, W, H, seed=float(rng.random()), spp=1 if mode >= 10 else int(rng.integers(1, 3)), max_bounces=int(rng.integers(1, 5)),
                           eye=tuple(eye), target=tuple(scene.target), render_mode=mode, accumulation_frames=int(rng.integers(1, 9)),
                           divisions=divisions, counter=counter % divisions, left_eye=int(rng.integers(0, 2)), depth=depth)
        counter += 1                                                    # :737, in every mode
        # SetShaderParameters: upload the lists as they stand, then clear / extend them (:492-526)
        t.set_camera_history(old)
        cams[:len(old)] = old
        old_uploaded = list(old)
        if mode < 5 or mode > 8 or p.counter == 0:
            old = []
        if mode < 11 and mode != 1:
            old = old + [pa.camera_history_entry(tuple(eye), tuple(scene.target), scene.fov, W / H)]
        before = (res.copy(), res1.copy(), res2.copy())
        t.render(p)
        orc.render_history(p, cams, W, H, res, res1, threads=4, result2=res2)
        g = (t.read_result(), t.read_converged(), t.read_target(pa.PT_TARGET_DENOISED))
        for which, a, b in (("Result", g[0], res), ("Result1", g[1], res1), ("Result2", g[2], res2)):
            if not same(a, b):
                bad = np.argwhere(~((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(-1))
                print("MISMATCH at frame", frames, which, "scene", name, f"{W}x{H}", "mode", mode, "divisions", divisions, "counter", p.counter,
                      "depth", depth, "spp", p.samples_per_pixel, "pixels", len(bad), "first", bad[:3].tolist(), events, flush=True)
                for y, x in bad[:3]:
                    print("  ", (int(x), int(y)), "gpu", a[y, x], "model", b[y, x], flush=True)
                diagnose(t, p, old_uploaded, before, g, (res, res1, res2), name, objs, W, H, frames)
                raise AssertionError("soak mismatch (details printed above)")
        frames += 1
        lit += int((res[..., :3] > 0).mean() > 0.5)
        if frames % 500 == 0:
            print(frames, "frames identical,", f"{time.time() - t0:.0f} s", events, flush=True)
    print(f"session soak ok: {frames} frames bit-identical ({lit} of them with more than half of the image lit); {events}", flush=True)
    t.close()
    return frames


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
