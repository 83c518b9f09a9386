#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (run here, committed).

The reference ships no golden data (it has no tests at all), so these vectors pin
the oracle against regressions and give the GPU tests a fixed target; they are
NOT outputs of the reference itself.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as pa  # noqa: E402
from oracle import pt_oracle as po  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
os.makedirs(out, exist_ok=True)

# (i) rand(): first 16 draws at seed 0.5 for four pixels of a 1080p frame
pix = [(0, 0), (1, 0), (959, 539), (1919, 1079)]
rand = np.stack([po.rand_sequence(0.5, x, y, 16) for x, y in pix])

# (ii) camera rays at the corners and centre, fov 90, near .1, far 100 (TestScene.unity:851-853)
c2w, ip = pa.camera_matrices((0, 4.6, 0), (0, 4.6, 1), 90.0, 1920 / 1080, z_near=0.1, z_far=100.0)
cam_px = [(0, 0), (1919, 0), (0, 1079), (1919, 1079), (960, 540)]
cam = np.stack([po.camera_ray(c2w, ip, x, y, 1920, 1080) for x, y in cam_px])

# (v) fixture scene, 64x36: hit ids, depth, 1-spp (4 and 8 bounces) and 16-spp images
s = pa.Scene("fixture")
sky = pa.procedural_sky(128, 64)
o = po.Oracle.from_scene(s, sky)
W, H = 64, 36
res = {}
for name, kw in {"spp1_b4": dict(spp=1, max_bounces=4), "spp1_b8": dict(spp=1, max_bounces=8),
                 "spp16_b8": dict(spp=16, max_bounces=8), "depth": dict(depth=1)}.items():
    img, ids, st = o.render(pa.make_params(s, W, H, seed=0.5, **kw), W, H, threads=4)
    res[name] = img
    res[name + "_rays"] = np.array([st["rays"], st["trace_calls"], st["paths"]], np.int64)
    res["ids"] = ids
np.savez_compressed(os.path.join(out, "fixture_64x36.npz"), rand_pixels=np.array(pix), rand=rand, cam_c2w=c2w, cam_ip=ip,
                    cam_pixels=np.array(cam_px), cam_rays=cam, **res)
print("wrote", os.path.join(out, "fixture_64x36.npz"), {k: v.shape for k, v in res.items()})

# (viii) renderMode 2: three frames of the fixture scene with a moving camera, 2 spp, 4 bounces, _AccumulationFrames 5
sys.path.insert(0, os.path.join(ROOT, "tests"))
from reproj_util import frame_params, oracle_sequence  # noqa: E402

seq = oracle_sequence(o, s, W, H, 3, spp=2, max_bounces=4, frames_n=5)
p2, _ = frame_params(s, W, H, 2, 2, 4, 5)
p2.render_mode = 1
plain2 = o.render(p2, W, H, threads=4)[0]
rp = {f"result{k}": r for k, (r, h, ids, st) in enumerate(seq)}
rp["plain_alpha2"] = plain2[..., 3]
rp["taken2"] = np.int64((np.abs(seq[2][0][..., 3] - plain2[..., 3]) > 0).sum())
np.savez_compressed(os.path.join(out, "reproj_64x36.npz"), **rp)
print("wrote reproj_64x36.npz; pixels that took the history in frame 2:", int(rp["taken2"]), "of", W * H)

# (ix) renderMode 0: the same three frames through reprojection -> Result2, denoiser and TAA -> Result
seq0 = oracle_sequence(o, s, W, H, 3, spp=2, max_bounces=4, frames_n=5, mode=0)
dn = {}
for k, fr in enumerate(seq0):
    dn[f"result{k}"] = fr[0]; dn[f"result2_{k}"] = fr[4]
np.savez_compressed(os.path.join(out, "denoise_64x36.npz"), **dn)
print("wrote denoise_64x36.npz; mean of the displayed frame 2:", float(seq0[2][0][..., :3].mean()))

# (x) renderMode 5 and 6: six frames, 4 stripes per cycle, moving camera; the displayed frame and the history mosaic
from reproj_util import oracle_stripe_sequence  # noqa: E402

st = {}
for mode in (5, 6):
    seqs = oracle_stripe_sequence(o, s, W, H, 6, 4, mode, spp=1, max_bounces=4)
    st[f"mode{mode}_result5"] = seqs[5][0]; st[f"mode{mode}_result1_5"] = seqs[5][1]
np.savez_compressed(os.path.join(out, "stripes_64x36.npz"), **st)
print("wrote stripes_64x36.npz")

# (xi) the remaining renderModes on the same scene: 3 and 10 (three frames, moving camera; last frame), 11 and 12 (two
# rendered frames, three paused; last frame), 7 and 8 (ten frames, 4 stripes per cycle; last frame's Result and Result2).
# renderMode 4 needs a tall image for its hard-coded offsets to land on screen and is covered by the parity tests only.
from reproj_util import oracle_pause_sequence  # noqa: E402

om = {}
for mode, spp in ((3, 2), (10, 1)):
    om[f"mode{mode}_result2"] = oracle_sequence(o, s, W, H, 3, spp=spp, max_bounces=4, frames_n=5, mode=mode)[2][0]
for mode in (11, 12):
    om[f"mode{mode}_result4"] = oracle_pause_sequence(o, s, W, H, 2, 3, mode)[4][0]
for mode in (7, 8):
    seqs = oracle_stripe_sequence(o, s, W, H, 10, 4, mode, spp=1, max_bounces=4)
    om[f"mode{mode}_result9"] = seqs[9][0]; om[f"mode{mode}_result2_9"] = seqs[9][4]
np.savez_compressed(os.path.join(out, "othermodes_64x36.npz"), **om)
print("wrote othermodes_64x36.npz", {k: float(v[..., :3].mean()) for k, v in om.items()})
