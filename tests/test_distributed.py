"""N > 1 path on CPU: world_size-2 gloo processes, each rendering its share of the samples with
the CPU oracle standing in for the GPU, reduced exactly as bench.py reduces on RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pathtracing_amd import distributed as D  # noqa: E402


def test_shard_helpers():
    assert D.shard_spp(1024, 8) == [128] * 8
    assert D.shard_spp(10, 4) == [3, 3, 2, 2] and sum(D.shard_spp(7, 3)) == 7
    assert D.shard_spp(1, 4) == [1, 0, 0, 0]
    rows = [D.tile_rows(1080, r, 8) for r in range(8)]
    assert rows[0][0] == 0 and sum(h for _, h in rows) == 1080
    assert all(rows[i][0] + rows[i][1] == rows[i + 1][0] for i in range(7))
    assert {D.sample_index(s, r, 4) for s in range(3) for r in range(4)} == set(range(12))
    assert [D.eye_for_rank(r, 2) for r in range(2)] == [-1, 1]


W, H, STEPS, SPP = 32, 18, 2, 2


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pathtracing_amd as pa
    from oracle import pt_oracle as po
    s = pa.Scene("fixture"); sky = pa.procedural_sky(32, 16)
    orc = po.Oracle.from_scene(s, sky, accel=True)
    acc = D.ShardedAccumulator(dst=0)
    for step in range(STEPS):
        k = D.sample_index(step, rank, world)
        img, _, _ = orc.render(pa.make_params(s, W, H, seed=pa.frame_seed(k), spp=SPP, max_bounces=4), W, H)
        acc.add(torch.from_numpy(img), SPP)
    # tile split: every rank renders its rows of one frame, gathered on rank 0
    y0, h = D.tile_rows(H, rank, world)
    p = pa.make_params(s, W, H, seed=0.5, spp=1, max_bounces=4, tile=(0, y0, W, h))
    part, _, _ = orc.render(p, W, H)
    rows = torch.from_numpy(part[y0:y0 + h].copy())
    gathered = [torch.zeros((D.tile_rows(H, r, world)[1], W, 4)) for r in range(world)] if rank == 0 else None
    dist.gather(rows, gathered, dst=0)
    if rank == 0:
        np.savez(out_path, image=acc.image().numpy(), weight=acc.weight, tiled=torch.cat(gathered).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_spp_sharded_reduce_matches_single_process(tmp_path, world):
    out = str(tmp_path / "r.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, out), nprocs=world, join=True, start_method="spawn")
    got = np.load(out)
    import pathtracing_amd as pa
    from oracle import pt_oracle as po
    s = pa.Scene("fixture"); sky = pa.procedural_sky(32, 16)
    orc = po.Oracle.from_scene(s, sky, accel=True)
    total = np.zeros((H, W, 4), np.float64)
    for k in range(STEPS * world):
        img, _, _ = orc.render(pa.make_params(s, W, H, seed=pa.frame_seed(k), spp=SPP, max_bounces=4), W, H)
        total += img.astype(np.float64) * SPP
    want = total / (STEPS * world * SPP)
    assert got["weight"] == STEPS * world * SPP
    rmse = np.sqrt(np.mean((got["image"] - want) ** 2))
    assert rmse < 1e-6 * max(1.0, np.abs(want).max())            # fp32 sum order differs across ranks: RMSE, not bits
    full, _, _ = orc.render(pa.make_params(s, W, H, seed=0.5, spp=1, max_bounces=4), W, H)
    assert (got["tiled"].view(np.uint32) == full.view(np.uint32)).all()   # disjoint rows: bit-exact
