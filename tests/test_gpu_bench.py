"""bench.py's N > 1 branches, run as the driver runs them (python -m torch.distributed.run, one process per rank) with two
ranks on the one GPU of the test box: RCCL refuses two ranks on one device, so the collectives go through gloo and host
memory (PTRACE_BENCH_BACKEND / PTRACE_BENCH_DEVICE); everything else is the code an 8-GPU node runs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, nproc=2):
    env = dict(os.environ, PTRACE_BENCH_DEVICE="0", PTRACE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", str(nproc), "--width", "480", "--height", "270", "--scene", "blob70k",
           "--no-cpu-baseline", "--no-stereo", "--no-reproj"] + extra
    if nproc == 1:
        cmd = [sys.executable, "bench.py", "--width", "480", "--height", "270", "--scene", "blob70k", "--no-cpu-baseline", "--no-stereo", "--no-reproj"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_share_the_samples_and_match_one_rank():
    d = _bench(["--steps", "3", "--warmup", "1", "--spp", "4"])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and [r["rank"] for r in d["ranks"]] == [0, 1]
    assert all(r["device"] == 0 and r["mrays_per_s"] > 0 for r in d["ranks"])
    assert d["config"]["frames_in_image"] == 6 and d["scaling"] == "weak"
    assert d["rmse_vs_one_rank"] is not None and d["rmse_vs_one_rank"] < 1e-6      # fp32 sums in another order, nothing more
    assert d["value"] > 0 and d["value_single_dispatch"] > 0 and d["roofline"]["frac"] > 0


def test_two_ranks_render_a_fixed_sample_count():
    d = _bench(["--total-spp", "24", "--spp", "4", "--warmup", "1"])
    assert d["scaling"] == "strong" and d["config"]["frames_in_image"] == 6 and d["time_to_image_ms"] > 0
    assert d["rmse_vs_one_rank"] < 1e-6


def test_two_ranks_take_one_eye_each():
    d = _bench(["--split", "eye", "--steps", "3", "--warmup", "1", "--width", "258", "--height", "276"])
    assert d["n_gpus"] == 2 and d["higher_is_better"] is False and d["config"]["stereo_pairs"] == 1
    assert len(d["ranks"]) == 2 and d["value"] > 0
    one = _bench(["--split", "eye", "--steps", "3", "--warmup", "1", "--width", "258", "--height", "276"], nproc=1)
    assert one["n_gpus"] == 1 and len(one["eye_image_means_rank0"]) == 2


def test_two_ranks_split_the_rows_of_a_frame():
    d = _bench(["--split", "rows", "--steps", "2", "--warmup", "1", "--spp", "2"])
    assert d["n_gpus"] == 2 and d["identical_to_one_gpu_image"] is True and len(d["ranks"]) == 2
