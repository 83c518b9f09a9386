"""Randomised soak of the C++ RayTracingMaster mirror (run on the GPU box): random renderMode / stripe-count / size / sample
switches and camera moves through libptmaster's Render(), the frames it dispatches re-rendered by the oracle from the
parameters it uploads and a Python restatement of its camera-list handling (RayTracingMaster.cs:492-526, 735-738).
usage: python tools/soak_master.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as pa  # noqa: E402
from oracle import pt_oracle as po  # noqa: E402
from pathtracing_amd import _ffi  # noqa: E402



def run(budget=60.0, seed=1):
    """One soak of `budget` seconds from `seed`; raises AssertionError on the first mismatch, returns the count of checked items."""
    rng = np.random.default_rng(seed)
    lib = _ffi.ptmaster(); pt = _ffi.ptrace()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    s = pa.Scene("fixture")
    assert lib.ptm_master_load_scene(h, s.handle, 64, 32) == 0, lib.ptm_master_error(h)
    orc = po.Oracle.from_scene(s, pa.procedural_sky(64, 32), accel=True)
    ctx = lib.ptm_master_context(h)


    def same(a, b):
        return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


    W, H, spp, maxb, mode, div = 40, 24, 1, 4, 2, 3
    lib.ptm_master_configure(h, W, H, spp, maxb, 0, pa.PT_ACCUM_NONE, 6)
    lib.ptm_master_set_divisions(h, div)
    res = np.zeros((H, W, 4), np.float32); res1 = np.zeros_like(res); res2 = np.zeros_like(res)
    cams = [pa.PtCameraHistory() for _ in range(100)]; old = []
    eye = np.array(s.eye, np.float64)
    t0 = time.time(); frames = 0; trail = []; recent = []
    while time.time() - t0 < budget:
        if rng.random() < 0.04:
            W2 = int(rng.integers(4, 70)); H2 = int(rng.integers(4, 45))
            if (W2, H2) != (W, H):                       # InitRenderTexture re-creates the targets only when the size changes (:545)
                W, H = W2, H2
                res = np.zeros((H, W, 4), np.float32); res1 = np.zeros_like(res); res2 = np.zeros_like(res)
        if rng.random() < 0.1:
            spp = int(rng.integers(1, 3)); maxb = int(rng.integers(1, 5))
        if rng.random() < 0.2:
            mode = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12]))
        if rng.random() < 0.05:
            div = int(rng.integers(1, 6)); lib.ptm_master_set_divisions(h, div)
        lib.ptm_master_configure(h, W, H, spp, maxb, int(rng.random() < 0.05), pa.PT_ACCUM_NONE, int(rng.integers(1, 9)))
        lib.ptm_master_set_render_mode(h, mode)
        eye = eye + rng.uniform(-0.15, 0.15, 3)
        lib.ptm_master_set_camera(h, (C.c_float * 3)(*eye), (C.c_float * 3)(*s.target), C.c_float(s.fov), C.c_float(0.0))
        p = pa.PtFrameParams()
        lib.ptm_master_peek_params(h, C.byref(p))
        recent = (recent + [bytes(p)])[-3:]
        trail = (trail + [(frames, W, H, mode, div, p.counter, p.depth, p.samples_per_pixel, p.max_bounces, p.accumulate, float(res.max()), float(res1.max()), float(res2.max()))])[-4:]
        assert p.render_mode == mode and p.divisions == div and p.samples_per_pixel == (spp if mode < 10 else 1)
        assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
        cams[:len(old)] = old
        orc.render_history(p, cams, W, H, res, res1, threads=4, result2=res2)
        if mode < 5 or mode > 8 or p.counter == 0:
            old = []
        if mode < 11 and mode != 1:
            e = pa.camera_history_entry(tuple(eye), tuple(s.target), s.fov, W / H)      # the mirror builds it from the same eye and target
            old = old + [e]
        got = [np.empty((H, W, 4), np.float32) for _ in range(3)]
        assert pt.pt_read_result(ctx, got[0].ctypes.data) == 0 and pt.pt_read_converged(ctx, got[1].ctypes.data) == 0
        assert pt.pt_read_target(ctx, pa.PT_TARGET_DENOISED, got[2].ctypes.data) == 0
        for which, a, b in (("Result", got[0], res), ("Result1", got[1], res1), ("Result2", got[2], res2)):
            if not same(a, b):
                bad = np.argwhere(~((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(-1))
                print("MISMATCH at frame", frames, which, f"{W}x{H}", "mode", mode, "div", div, "counter", p.counter, "pixels", len(bad), flush=True)
                print("  last frames (index, W, H, mode, div, counter, depth, spp, bounces, accumulate, model Result/Result1/Result2 max BEFORE the frame):", trail, flush=True)
                os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
                np.savez(os.path.join(ROOT, 'gpurun_out', 'soak_master_failure.npz'), params=np.frombuffer(b''.join(recent), np.uint8), gpu=a, model=b,
                         which=which, gpu_all=np.stack(got), model_all=np.stack([res, res1, res2]), trail=np.array(trail, np.float64))
                y, x = bad[0]; print("  first", (int(x), int(y)), "gpu", a[y, x], "model", b[y, x], flush=True)
                raise AssertionError("soak mismatch (details printed above)")
        frames += 1
        if frames % 1000 == 0:
            print(frames, "frames identical,", f"{time.time() - t0:.0f} s", flush=True)
    lib.ptm_master_destroy(h)
    print(f"master soak ok: {frames} frames bit-identical", flush=True)
    return frames


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
