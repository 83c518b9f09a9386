"""bench.py's N > 1 branches, run as the driver runs them (python -m torch.distributed.run, one process per rank) with two
ranks on the one GPU of the test box: RCCL refuses two ranks on one device, so the collectives go through gloo and host
memory (PTRACE_BENCH_BACKEND / PTRACE_BENCH_DEVICE); everything else is the code an 8-GPU node runs."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A port nobody listens on: bound, read, closed - not derived from the pid, which two launches of one test share."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _bench(extra, nproc=2):
    env = dict(os.environ, PTRACE_BENCH_DEVICE="0", PTRACE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    port = _free_port()
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


def _bench_rccl_one_rank(extra):
    """bench.py with a process group of ONE rank on the RCCL backend (PTRACE_BENCH_FORCE_DIST=1): init_process_group("nccl",
    device_id=...), and every collective of the N-rank job on device tensors - the identity here, but the very calls."""
    env = dict(os.environ, PTRACE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    for k in ("PTRACE_BENCH_DEVICE", "PTRACE_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--width", "480", "--height", "270", "--scene", "blob70k", "--no-cpu-baseline", "--no-stereo",
           "--no-reproj"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [["--steps", "3", "--warmup", "1", "--spp", "4"], ["--total-spp", "32", "--spp", "4", "--warmup", "1"],
                                   ["--split", "rows", "--steps", "2", "--warmup", "1", "--spp", "2"],
                                   ["--split", "eye", "--steps", "2", "--warmup", "1", "--width", "258", "--height", "276"]],
                         ids=["spp", "total-spp", "rows", "eye"])
def test_the_rccl_backend_carries_every_collective_of_the_job(extra):
    d = _bench_rccl_one_rank(extra)
    assert d["collective_backend"] == "nccl" and d["rccl_world_size"] == 1 and d["world_size"] == 1 and d["n_gpus"] == 1
    assert len(d["ranks"]) == 1 and d["ranks"][0]["device"] == 0 and d["ranks"][0]["device_name"] and d["ranks"][0]["mrays_per_s"] > 0
    assert d["value"] > 0
    if "--total-spp" in extra:
        assert d["config"]["frames_in_image"] == 8 and d["time_to_image_ms"] > 0
    if "rows" in extra:
        assert d["identical_to_one_gpu_image"] is True


def test_a_request_for_two_gpus_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` without a launcher's environment starts its own two ranks (here both on device 0, collectives
    through host memory) and relays their one line; it never prints a one-rank line for a two-GPU request."""
    env = dict(os.environ, PTRACE_BENCH_DEVICE="0", PTRACE_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--width", "480", "--height", "270", "--scene", "blob70k", "--no-cpu-baseline", "--no-stereo",
           "--no-reproj", "--steps", "2", "--warmup", "1", "--spp", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and len(d["ranks"]) == 2 and d["collective_backend"] == "gloo" and d["rccl_world_size"] is None
    # and a launcher that started another number of ranks than asked for is an error, not a relabelled line
    env1 = dict(env, WORLD_SIZE="1", RANK="0")
    bad = subprocess.run(cmd, cwd=ROOT, env=env1, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
