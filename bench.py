#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A step = one pass of the hot path over one frame of the workload: a 1920x1080
image of the ~260 k-triangle hall scene, 8 bounces, --spp samples per pixel per
GPU.  With N GPUs every rank renders its own --spp samples of the same image
(different sample indices / seeds; the scene is replicated), adds them into its
own accumulation buffer inside the kernel's final write, and the job ends with
one sum-reduce of those buffers to rank 0 over RCCL, inside the timed region:
weak scaling - "many spp sharded across the GPUs with an RCCL reduce".

Prints ONE JSON line (rank 0): metric Mrays/s = Trace() calls executed by the
kernels (counted on the device) per second of wall time, summed over ranks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NODE_BYTES = 56      # one four-wide BVH node as fetched: origin, exponents, 4 links, 4 x 6 plane bytes (csrc/bvh.hpp)
PRIM_BYTES = 48      # one primitive record: v0, e1, e2 (+ id, kind)
PIXEL_BYTES = 48     # per pixel: RGBA32F Result write + accumulation buffer read and write
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="hall260k")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=16, help="samples per pixel per GPU per step")
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--in-flight", type=int, default=3,
                    help="contexts per GPU that render alternate steps on their own streams, so that one dispatch's "
                         "ramp-up and tail are filled by the next (1 = one dispatch at a time)")
    ap.add_argument("--split", choices=["spp", "eye", "rows"], default="spp",
                    help="how N GPUs share the work: samples (default; one sum-reduce per job), stereo eyes (config 3: no "
                         "collective), or row blocks of one frame (a gather)")
    ap.add_argument("--total-spp", type=int, default=0,
                    help="with --split spp: render this many samples per pixel in total, in frames of --spp, sharded over the "
                         "ranks (time to image, strong scaling); 0 = every rank renders --steps frames (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-stereo", action="store_true", help="skip the 1080p stereo frame time (2 eyes, 1 spp, 4 bounces)")
    ap.add_argument("--no-reproj", action="store_true", help="skip the renderMode-2 (temporal reprojection) frame time")
    return ap.parse_args()


def cpu_baseline(scene, sky, args):
    """The CPU oracle (a port of the reference shader) on a bounded sample of the same workload."""
    from oracle import pt_oracle as po
    import pathtracing_amd as pa
    threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
    W, H = args.width, args.height
    tile = (0, 0, W, H)
    spp = 4
    orc = po.Oracle.from_scene(scene, sky, accel=True)
    p = pa.make_params(scene, W, H, seed=pa.frame_seed(0), spp=spp, max_bounces=args.bounces, tile=tile)
    t0 = time.perf_counter()
    _, _, st = orc.render(p, W, H, threads=threads)
    dt = time.perf_counter() - t0
    out = {"value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "nproc": os.cpu_count(), "kind": "port",
           "sample": f"the whole {W}x{H} frame at {spp} spp (of {args.spp}), {args.bounces} bounces, "
                     f"oracle with its own median-split BVH, {st['rays']} rays in {dt:.1f} s"}
    # the reference's own algorithm has no acceleration structure: time the exhaustive loop on a much smaller sample
    tile2 = (W // 2 - 96, H // 2 - 54, 192, 108)
    orc2 = po.Oracle.from_scene(scene, sky, accel=False)
    p2 = pa.make_params(scene, W, H, seed=pa.frame_seed(0), spp=1, max_bounces=args.bounces, tile=tile2)
    t0 = time.perf_counter()
    _, _, st2 = orc2.render(p2, W, H, threads=threads)
    dt2 = time.perf_counter() - t0
    # one thread, smaller sample
    tile1 = (W // 4, H // 4, W // 2, H // 2)
    p1 = pa.make_params(scene, W, H, seed=pa.frame_seed(0), spp=1, max_bounces=args.bounces, tile=tile1)
    t0 = time.perf_counter()
    _, _, st1 = orc.render(p1, W, H, threads=1)
    dt1 = time.perf_counter() - t0
    out["single_thread"] = {"value": st1["rays"] / dt1 / 1e6, "unit": "Mrays/s", "cores": 1,
                            "sample": f"central {tile1[2]}x{tile1[3]} tile, 1 spp, {st1['rays']} rays in {dt1:.1f} s"}
    out["exhaustive_loop"] = {"value": st2["rays"] / dt2 / 1e6, "unit": "Mrays/s",
                              "sample": f"{tile2[2]}x{tile2[3]} tile, 1 spp, every triangle tested per ray as the "
                                        f"reference does, {st2['rays']} rays in {dt2:.1f} s"}
    return out


def cpu_baseline_c1(args):
    """Configuration 1 on the host cores: the sphere field of RayTracingMaster.cs:32-34 on the ground plane under the sky,
    1920x1080, 8 bounces, 1 spp, by the oracle's loop over every sphere (the shader's own algorithm)."""
    from oracle import pt_oracle as po
    import pathtracing_amd as pa
    scene = pa.Scene("spheres"); sky = pa.procedural_sky(2048, 1024)
    threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
    W, H = 1920, 1080
    orc = po.Oracle.from_scene(scene, sky, accel=False)
    out = {}
    for name, n, reps in (("all", threads, 8), ("one", 1, 1)):
        t0 = time.perf_counter()
        for k in range(reps):
            _, _, st = orc.render(pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=1, max_bounces=8), W, H, threads=n)
        dt = (time.perf_counter() - t0) / reps
        out[name] = {"value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "ms_per_frame": dt * 1e3, "cores": n}
    return {"value": out["all"]["value"], "unit": "Mrays/s", "ms_per_frame": out["all"]["ms_per_frame"], "cores": threads,
            "nproc": os.cpu_count(), "kind": "port", "single_thread": out["one"],
            "sample": f"whole frames: spheres scene ({len(scene.spheres)} spheres + ground plane + sky), {W}x{H}, 8 bounces, 1 spp, "
                      "exhaustive loop over the spheres as in the shader; mean of 8 frames"}


# Arithmetic per unit of work, counted from the kernel source (one fused multiply-add = 2 flops; conversions, selects and
# integer work not counted): a four-child node visit = 24 plane FMAs + 6 FMAs/multiplies of set-up + 4 x (5 min/max +
# 1 multiply + 2 compares); a primitive test = Moller-Trumbore (two cross products, four dot products, one divide);
# Shade() + ray set-up per ray ~ 150.
FLOPS_PER_NODE, FLOPS_PER_PRIM, FLOPS_PER_RAY = 48 + 9 + 32, 45, 150
VALU_PEAK_TFLOPS = 157.3


def valu_view(solo, launches, kernel_ms):
    flops = (solo["nodes_visited"] * FLOPS_PER_NODE + solo["prims_tested"] * FLOPS_PER_PRIM + solo["rays"] * FLOPS_PER_RAY) / launches
    achieved = flops / (kernel_ms * 1e-3) / 1e12
    return {"achieved": achieved, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / VALU_PEAK_TFLOPS,
            "flops_per_ray": flops * launches / max(1, solo["rays"]),
            "note": "counted useful flops; the VALU itself is busy about half of the time at 45 % of lanes (profiles/)"}


def measured_traffic(args):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes, when they were
    taken on this workload, and the file they come from; (None, None) otherwise (bench.py cannot run the profiler on
    itself: this figure is NOT measured by the run that prints it)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))      # the most recent one
    if not found or (args.scene, args.width, args.height, args.spp, args.bounces) != ("hall260k", 1920, 1080, 16, 8):
        return None, None
    t = json.load(open(found[-1]))
    src = "committed file profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; kernel %s, %s), not this run" % (
        os.path.basename(found[-1]), t.get("kernel", "?"), t.get("workload", "?"))
    return (2.0 * t["fetch_size_kb_per_launch"] + t["write_size_kb_per_launch"]) * 1024.0, src


def limiter_view(kernel):
    """What the committed rocprofv3 PMC passes say limits the dominant kernel (profiles/r*_limiter.json, written from those
    passes by tools/limiter_json.py): lanes active per vector instruction, how busy the vector ALUs and the texture
    addressers are, the share of wave time spent waiting.  bench.py cannot profile itself; None without the file."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_limiter.json")))
    if not found:
        return None
    t = json.load(open(found[-1]))
    view = dict(t.get(kernel, t.get("path_sched_kernel<4>")))
    view["source"] = "committed file profiles/%s (%s), kernel %s; not this run" % (
        os.path.basename(found[-1]), t.get("source", "rocprofv3 --pmc passes"), kernel if kernel in t else "path_sched_kernel<4>")
    return view


def gather_ceiling():
    """The measured ceiling for what the traversal does - every lane chasing its own chain of 64-byte records, each fetched
    as four 16-byte loads, table resident in L2 / Infinity Cache (tools/ubench/gather_rate, profiles/r*_gather_rate.json)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gather_rate.json")))
    if not found:
        return None
    c = json.load(open(found[-1]))
    c["source"] = "committed file profiles/%s (tools/ubench/gather_rate on an MI355X), not this run" % os.path.basename(found[-1])
    return c


def free_port():
    """A TCP port nobody listens on right now: bound, read and closed again."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """`bench.py --gpus N` with N > 1 and no launcher's environment: start the N ranks here, one process per GPU, exactly as
    the documented command line does (python -m torch.distributed.run ... bench.py --gpus N ...), relay rank 0's JSON line
    and leave with the job's exit code.  Runs before anything in this process touches the GPU: the ranks are children,
    never a replacement of a process that has initialised HIP.  A request for N GPUs therefore yields an N-rank line or a
    non-zero exit, never a one-rank line."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE; starting the ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    job = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in job.stdout.splitlines() if ln.startswith("{")]
    for ln in job.stdout.splitlines():
        print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr, flush=True)
    if job.returncode == 0 and len(lines) == 1 and json.loads(lines[0]).get("n_gpus") != args.gpus:
        raise SystemExit("bench.py: the job reported %s GPUs for --gpus %d" % (json.loads(lines[0]).get("n_gpus"), args.gpus))
    raise SystemExit(job.returncode if job.returncode else (0 if len(lines) == 1 else 1))


def setup_dist(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "PTRACE_BENCH_DEVICE" in os.environ:        # rehearsals of the N > 1 path on a box with fewer GPUs than ranks
        local_rank = int(os.environ["PTRACE_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); refusing to report "
                         f"one as the other")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dist = None
    # PTRACE_BENCH_FORCE_DIST=1: a process group of one rank too, so that a single GPU can rehearse the RCCL calls themselves
    if world > 1 or (os.environ.get("PTRACE_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist
        # "nccl" is RCCL.  PTRACE_BENCH_BACKEND=gloo (with PTRACE_BENCH_DEVICE) rehearses the N > 1 code on a box with one GPU,
        # where RCCL refuses two ranks on one device; the collectives then go through host memory.
        backend = os.environ.get("PTRACE_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world and dist.get_rank() == rank
    return rank, local_rank, world, dist


def make_collective(dist):
    def collective(fn, tensor, **kw):
        """One collective on a CUDA tensor: directly on RCCL, through a host copy on the rehearsal backend."""
        if dist.get_backend() == "nccl":
            fn(tensor, **kw)
        else:
            host = tensor.cpu()
            fn(host, **kw)
            tensor.copy_(host)
    return collective


def rank_table(torch, dist, collective, world, rank, local_rank, rays, dt):
    """Device ordinal, device name and Mrays/s of every rank, so that the line shows which N GPUs took part."""
    name = torch.cuda.get_device_name(local_rank)
    mine = torch.tensor([float(local_rank), rays / dt / 1e6], dtype=torch.float64, device="cuda")
    if dist is None:
        return [{"rank": 0, "device": local_rank, "device_name": name, "mrays_per_s": float(mine[1])}]
    allr = torch.zeros((world, 2), dtype=torch.float64, device="cuda")
    allr[rank] = mine
    collective(dist.all_reduce, allr, op=dist.ReduceOp.SUM)
    names = torch.zeros((world, 64), dtype=torch.int32, device="cuda")       # the names, as character codes, summed into place
    code = list(name.encode("ascii", "replace"))[:64]
    names[rank, :len(code)] = torch.tensor(code, dtype=torch.int32, device="cuda")
    collective(dist.all_reduce, names, op=dist.ReduceOp.SUM)
    return [{"rank": r, "device": int(allr[r, 0]), "device_name": bytes(int(c) for c in names[r].tolist() if c).decode("ascii", "replace"),
             "mrays_per_s": float(allr[r, 1])} for r in range(world)]


def group_view(dist):
    """What carried the collectives: {"world_size", "backend", "rccl_world_size"} - the last one is the size of the RCCL
    communicator when the backend is RCCL ("nccl" in torch), None on the host-memory rehearsal backend or without a group."""
    if dist is None:
        return {"world_size": 1, "collective_backend": None, "rccl_world_size": None}
    backend = dist.get_backend()
    return {"world_size": dist.get_world_size(), "collective_backend": backend,
            "rccl_world_size": dist.get_world_size() if backend == "nccl" else None}


def relative_rmse(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(1e-30, np.sqrt(np.mean(b ** 2))))


def main():
    args = parse()
    self_launch(args)               # N > 1 without a launcher: the ranks are started here, before any GPU call
    rank, local_rank, world, dist = setup_dist(args)
    import torch
    import pathtracing_amd as pa
    from pathtracing_amd import distributed as D
    collective = make_collective(dist) if dist is not None else None
    if args.split == "eye":
        return eye_split_job(args, pa, torch, dist, collective, rank, local_rank, world)
    if args.split == "rows":
        return row_split_job(args, pa, torch, D, dist, collective, rank, local_rank, world)

    W, H = args.width, args.height
    scene = pa.Scene(args.scene)
    sky = pa.procedural_sky(2048, 1024)
    strong = args.total_spp > 0
    if strong:
        # time to image: --total-spp samples per pixel in frames of --spp samples (one _Seed per frame, the shader's RNG
        # counter running on across a frame's samples), frame k rendered by rank k mod N
        if args.total_spp % args.spp:
            raise SystemExit("--total-spp must be a multiple of --spp (a frame is one dispatch of --spp samples)")
        n_frames = args.total_spp // args.spp
        my_frames = [k for k in range(n_frames) if k % world == rank]
        args.steps = max(1, len(my_frames))
    # Every context is one "ComputeShader instance" with its own targets and stream.  A dispatch's persistent grid
    # fills the GPU, but its ramp-up and above all its tail (lanes finishing their last pixel) leave slots empty;
    # with several dispatches queued on different streams the next one's blocks move in as slots free up.
    n_ctx = max(1, min(args.in_flight, args.steps))
    trs, accums, streams = [], [], []
    for i in range(n_ctx):
        t = pa.PathTracer(local_rank)
        t.load_scene(scene, sky)
        t.resize(W, H)
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")   # this context's sum over its steps ("Result1")
        t.bind_target(pa.PT_TARGET_CONVERGED, acc.data_ptr())
        st_i = torch.cuda.Stream()
        t.set_stream(st_i.cuda_stream)
        t.trace_rays(np.array([[0, 5, 0, 0, -1, 0]], np.float32))   # builds the acceleration structure now, not inside a timed step
        trs.append(t); accums.append(acc); streams.append(st_i)
    tr = trs[0]
    total = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")     # this rank's sum; on rank 0 the job's sum
    job = D.ShardedAccumulator(dst=0)

    def frame_index(step):
        return my_frames[step] if strong else D.sample_index(step, rank, world)

    def run(first, count, timed):
        """`count` steps, dealt round-robin to the contexts: each adds its frames into its own accumulation buffer
        inside the kernel's final write (one _Seed per frame index); then the rank's buffers are added up and ONE
        sum-reduce takes them to rank 0 (ShardedAccumulator: a weighted sum, the weight being the frames rendered)."""
        for j in range(count):
            p = pa.make_params(scene, W, H, seed=pa.frame_seed(frame_index(first + j)), spp=args.spp,
                               max_bounces=args.bounces, accumulate=pa.PT_ACCUM_SUM, sample_index=j // n_ctx)
            trs[j % n_ctx].render(p)
        cur = torch.cuda.current_stream()
        used = min(n_ctx, count)
        for i in range(used):
            cur.wait_stream(streams[i])
        total.zero_()
        for i in range(used):
            total.add_(accums[i])
        acc = job if timed else D.ShardedAccumulator(dst=0)
        acc.add(total, 1.0, weight=float(count), reduce=(lambda t, **kw: collective(dist.reduce, t, **kw)) if dist is not None else None,
                all_reduce=(lambda t, **kw: collective(dist.all_reduce, t, **kw)) if dist is not None else None)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # A context hands out pixel blocks by what they cost in its previous dispatch over the same rectangle; with fewer warm-up
    # steps than contexts some context would meet the timed region without that, so those render one untimed frame first.
    for i in range(min(args.warmup, n_ctx), n_ctx):
        trs[i].render(pa.make_params(scene, W, H, seed=0.123, spp=args.spp, max_bounces=args.bounces))
    warm = 0 if strong else args.warmup
    if strong:
        # warm-up frames are extra frames that are not part of the image
        for j in range(args.warmup):
            trs[j % n_ctx].render(pa.make_params(scene, W, H, seed=0.123, spp=args.spp, max_bounces=args.bounces, accumulate=pa.PT_ACCUM_SUM, sample_index=0))
        if dist is not None:
            collective(dist.reduce, total, dst=0, op=dist.ReduceOp.SUM)
    else:
        run(0, args.warmup, False)             # also warms the collective up
    fence()
    for t in trs:
        t.stats()                   # close the statistics epoch of the warm-up
    n_timed = len(my_frames) if strong else args.steps
    t0 = time.perf_counter()
    run(warm, n_timed, True)
    fence()
    dt = time.perf_counter() - t0
    sts = [t.stats() for t in trs]
    st = dict(sts[0])
    for k in ("rays", "nodes_visited", "prims_tested", "paths"):
        st[k] = sum(x[k] for x in sts)
    # The dominant kernel by itself: the same launches one at a time on one context, timed by HIP events on its stream.
    # (In the region above up to n_ctx launches overlap, so a launch there lasts longer than it keeps the GPU busy.)
    solo_launches = 3
    for j in range(solo_launches + 1):      # the first one lets the context measure its blocks' costs; it is not timed
        if j == 1:
            tr.stats()
        tr.render(pa.make_params(scene, W, H, seed=pa.frame_seed(D.sample_index(args.warmup + j, rank, world)), spp=args.spp,
                                 max_bounces=args.bounces, accumulate=pa.PT_ACCUM_SUM, sample_index=j))
    solo = tr.stats()

    t_all = torch.tensor([dt], dtype=torch.float64, device="cuda")
    counts = torch.tensor([st["rays"], st["nodes_visited"], st["prims_tested"], st["paths"]], dtype=torch.float64, device="cuda")
    ranks = rank_table(torch, dist, collective, world, rank, local_rank, st["rays"], dt)
    if dist is not None:
        collective(dist.all_reduce, t_all, op=dist.ReduceOp.MAX)
        collective(dist.all_reduce, counts, op=dist.ReduceOp.SUM)
    dt = float(t_all.item())
    rays, nodes, prims, paths = [float(x) for x in counts.tolist()]

    if rank == 0:
        img = job.image().cpu().numpy()     # every frame is already a mean over --spp samples
        frames_total = int(round(job.weight))
        mrays = rays / dt / 1e6
        # The same frame indices rendered by this rank alone, added up in index order: what the N-rank image must match
        # (fp32 sums in another order: an RMSE, not bits).
        rmse = None
        if world > 1 and frames_total <= 256:
            ref = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
            tr.bind_target(pa.PT_TARGET_CONVERGED, ref.data_ptr())
            order = range(frames_total) if strong else [D.sample_index(warm + j, r, world) for j in range(n_timed) for r in range(world)]
            for j, k in enumerate(order):
                tr.render(pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=args.spp, max_bounces=args.bounces,
                                         accumulate=pa.PT_ACCUM_SUM, sample_index=j))
            tr.sync()
            rmse = relative_rmse(img, (ref / float(frames_total)).cpu().numpy())
            tr.bind_target(pa.PT_TARGET_CONVERGED, accums[0].data_ptr())
            tr.stats()
        # roofline of the dominant kernel from the solo launches: counted bytes per launch over the launch duration.
        # roofline["pipelined"] is the rate over the timed region, where n_ctx launches overlap.
        kernel = {0: "path_sched_kernel<4>", 1: "wf_trace<4>", 2: "path_pool_kernel<4>"}.get(
            {"sched": 0, "mega": 0, "wavefront": 1, "wave": 1, "pool": 2}.get(os.environ.get("PTRACE_BACKEND", "sched"), 0))
        launches = n_timed
        kernel_ms = solo["kernel_ms"] / solo_launches
        r_rank = solo["rays"]
        alg_bytes = (solo["nodes_visited"] * NODE_BYTES + solo["prims_tested"] * PRIM_BYTES) / solo_launches + W * H * PIXEL_BYTES
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        pipe_bytes = st["nodes_visited"] * NODE_BYTES + st["prims_tested"] * PRIM_BYTES + launches * W * H * PIXEL_BYTES
        pipe_achieved = pipe_bytes / dt / 1e9
        single = r_rank / solo_launches / (kernel_ms * 1e-3) / 1e6
        ceiling = gather_ceiling()
        traffic, traffic_source = measured_traffic(args)
        lane_loads = (solo["nodes_visited"] * 4 + solo["prims_tested"] * 3) / solo_launches      # 16-byte loads per lane
        out = {
            "metric": "Mrays/s (all bounces)", "value": mrays, "unit": "Mrays/s", "n_gpus": world,
            "steps": n_timed, "warmup": args.warmup, "ms_per_step": dt / n_timed * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the same workload with one dispatch at a time on one stream - what a host that issues pt_render per frame gets,
            # and the run the roofline object describes; `value` keeps --in-flight dispatches queued on as many streams
            "value_single_dispatch": single,
            "config": {"workload": f"{args.scene} ({scene.n_triangles} triangles + {len(scene.spheres)} spheres + ground plane, "
                                   f"2048x1024 procedural sky) {W}x{H}, {args.bounces} bounces, {args.spp} spp per GPU per step",
                       "parallelism": (f"{args.total_spp} spp in frames of {args.spp}, frame k on rank k mod {world}" if strong else f"spp-sharded x{world}")
                       + ", scene replicated, in-kernel accumulation per GPU, one RCCL sum-reduce of the accumulation buffers per job"
                       if world > 1 else "single GPU",
                       "rays_per_step": rays / n_timed / world, "paths_per_step": paths / n_timed / world,
                       "frames_in_image": frames_total, "dispatches_in_flight": n_ctx},
            "ranks": ranks, **group_view(dist),
            "rmse_vs_one_rank": rmse,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel, "kernel_ms": kernel_ms, "launches_timed": solo_launches,
                         "pipelined": {"launches_in_flight": n_ctx, "achieved": pipe_achieved, "frac": pipe_achieved / HBM_PEAK_GBS,
                                       "effective_ms_per_launch": dt / launches * 1e3},
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_ray": (solo["nodes_visited"] * NODE_BYTES + solo["prims_tested"] * PRIM_BYTES) / max(1, r_rank),
                         "nodes_per_ray": solo["nodes_visited"] / max(1, r_rank),
                         "prims_per_ray": solo["prims_tested"] / max(1, r_rank),
                         "kernel_mrays_per_s": single,
                         # The nominal HBM fraction above prices cache-resident gathers against DRAM bandwidth.  The ceilings
                         # that apply: the scene (nodes + records ~ 17 MB) sits in L2 / Infinity Cache and every lane
                         # fetches its own 64-byte record as 16-byte loads, for which the L1 sustains the measured rate below.
                         "cache_gather": None if ceiling is None else {
                             "achieved_tb_s": achieved / 1e3,
                             "lane_loads_g_per_s": lane_loads / (kernel_ms * 1e-3) / 1e9,
                             "lane_loads_ceiling_g_per_s": ceiling["lane_loads_g_per_s"],
                             "frac_of_l1_gather_ceiling": lane_loads / (kernel_ms * 1e-3) / 1e9 / ceiling["lane_loads_g_per_s"],
                             "ceiling_tb_s": ceiling["lane_loads_g_per_s"] * 16e-3,
                             "guide_rows_from_l2_tb_s": [16.8, 18.8], "guide_rows_from_infinity_cache_tb_s": 8.6,
                             "note": ceiling["note"], "source": ceiling["source"]},
                         "limiter": limiter_view(kernel),
                         # the other roofline the kernel could sit under: counted arithmetic against the fp32 vector peak
                         "valu": valu_view(solo, solo_launches, kernel_ms)},
            "image_mean_rgb": [float(x) for x in img[..., :3].reshape(-1, 3).mean(axis=0)],
            "image_crc": int(np.bitwise_xor.reduce(np.ascontiguousarray(img).view(np.uint32).reshape(-1))),
            "bvh": {"nodes": st["bvh_nodes"], "prims": st["bvh_prims"], "depth": st["bvh_depth"], "build_ms": st["bvh_build_ms"]},
        }
        if strong:
            out["time_to_image_ms"] = dt * 1e3
        # what a transform change costs: matrices in, primitive records and node boxes redone on the device
        tr.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            tr.update_transforms(scene.objects)
        tr.sync()
        out["bvh"]["refit_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        out["bvh"]["objects"] = int(len(scene.objects))
        if not args.no_stereo and world == 1:
            out["stereo_1080p"] = stereo_frame_time(pa, torch)
        if not args.no_stereo and world == 1:
            out["blob70k_1080p"] = secondary_scene_rate(pa, torch, sky)
            out["ggx_1080p"] = secondary_scene_rate(pa, torch, sky, name=args.scene, ggx=True)
            out["reuse_primary_hit"] = secondary_scene_rate(pa, torch, sky, name=args.scene, reuse=True)
            # what bit-exactness costs: the same workload with Shade() on the hardware's approximate transcendentals
            # (same primary hits and random numbers, image within a relative RMSE of 5e-3: tests/test_gpu_fast_intrinsics.py)
            out["fast_intrinsics"] = secondary_scene_rate(pa, torch, sky, name=args.scene, fast=True)
            out["value_fast_intrinsics"] = out["fast_intrinsics"]["value"]
            # the same workload on the tree the host builder makes (option tree_build 0; the default builds on the device):
            # build_ms against bvh.build_ms above, and the rays per second of its tree
            out["host_tree"] = secondary_scene_rate(pa, torch, sky, name=args.scene, host_tree=True)
            # 41 % of the headline's rays repeat a pixel's camera ray (the shader applies no jitter); this is the rate of
            # the rays that differ, i.e. of incoherent traversal
            out["value_distinct_rays"] = out["reuse_primary_hit"]["value"]
        if not args.no_reproj and world == 1:
            out["reproj_1080p"] = reproj_frame_time(pa, torch, scene, sky)
        if not args.no_stereo and world == 1:
            # the headline workload with the eye advancing 0.02 units per step: every dispatch hands its blocks out by the
            # costs the PREVIOUS camera position measured
            out["value_moving_camera"] = moving_camera_rate(pa, tr, scene, args)
            out["scene_change"] = scene_change_time(pa, torch, scene, sky)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, sky, args)
            out["cpu_baseline_c1"] = cpu_baseline_c1(args)
            out["c1_gpu"] = c1_gpu(pa, torch)
        print(json.dumps(out), flush=True)
    for t in trs:
        t.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def eye_split_job(args, pa, torch, dist, collective, rank, local_rank, world):
    """Configuration 3: a stereo XR frame, 2 x 2064x2208, 4 bounces, 1 spp + running-mean accumulation.  Rank r owns eye
    r mod 2 of stereo pair r div 2 (its own accumulator, no reduce: the eyes go to different displays); with one rank both
    eyes run on its GPU side by side, one context and stream each.  metric: ms per stereo frame (max over ranks)."""
    W, H = (args.width, args.height) if (args.width, args.height) != (1920, 1080) else (2064, 2208)
    scene = pa.Scene("xr"); sky = pa.procedural_sky(2048, 1024)
    from pathtracing_amd import distributed as D
    eyes = [-1, 1] if world == 1 else [D.eye_for_rank(rank, world)]      # -1: left eye, +1: right eye
    sep = 0.022                                                   # TestScene.unity:870
    trs = []
    for e in eyes:
        t = pa.PathTracer(local_rank)
        t.load_scene(scene, sky); t.resize(W, H)
        if len(eyes) == 2:
            t.set_option("blocks_per_cu", 2)                      # two dispatches side by side on one GPU
        t.trace_rays(np.array([[0, 5, 0, 0, -1, 0]], np.float32))
        trs.append(t)

    def frame(k):
        for e, t in zip(eyes, trs):
            off = np.array([0.5 * e * sep, 0, 0], np.float32)
            t.render(pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=1, max_bounces=4, accumulate=pa.PT_ACCUM_MEAN,
                                    sample_index=min(k, 10), eye=scene.eye + off, target=scene.target + off, fov=90.0))

    def fence():
        for t in trs:
            t.sync()
        if dist is not None:
            dist.barrier()

    for k in range(args.warmup):
        frame(k)
    fence()
    for t in trs:
        t.stats()
    t0 = time.perf_counter()
    for k in range(args.steps):
        frame(args.warmup + k)
    fence()
    dt = time.perf_counter() - t0
    rays = sum(t.stats()["rays"] for t in trs)
    means = [float(t.read_converged()[..., :3].mean()) for t in trs]
    t_all = torch.tensor([dt], dtype=torch.float64, device="cuda")
    r_all = torch.tensor([float(rays)], dtype=torch.float64, device="cuda")
    ranks = rank_table(torch, dist, collective, world, rank, local_rank, rays, dt)
    if dist is not None:
        collective(dist.all_reduce, t_all, op=dist.ReduceOp.MAX)
        collective(dist.all_reduce, r_all, op=dist.ReduceOp.SUM)
    if rank == 0:
        pairs = max(1, world // 2)
        print(json.dumps({
            "metric": "ms/frame for a stereo XR frame (2 x %dx%d)" % (W, H), "value": float(t_all.item()) / args.steps * 1e3, "unit": "ms",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t_all.item()) / args.steps * 1e3,
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"xr scene, stereo {W}x{H} per eye, 4 bounces, 1 spp + running-mean accumulation",
                       "parallelism": ("both eyes on one GPU, one context and stream per eye" if world == 1 else
                                       f"eye r mod 2 of stereo pair r div 2 on rank r ({pairs} pair(s)); no collective in the frame"),
                       "stereo_pairs": pairs},
            "mrays_per_s": float(r_all.item()) / float(t_all.item()) / 1e6, "ranks": ranks,
            **group_view(dist), "eye_image_means_rank0": means}), flush=True)
    for t in trs:
        t.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


def row_split_job(args, pa, torch, D, dist, collective, rank, local_rank, world):
    """Latency mode: one frame of the headline workload per step, rank r rendering its block of rows (tile_*), the blocks
    gathered on rank 0 - disjoint pixels, no arithmetic, so the image equals the one-GPU image bit for bit (checked on
    rank 0 against its own full render).  metric: ms per frame."""
    W, H = args.width, args.height
    scene = pa.Scene(args.scene); sky = pa.procedural_sky(2048, 1024)
    t = pa.PathTracer(local_rank)
    t.load_scene(scene, sky); t.resize(W, H)
    t.trace_rays(np.array([[0, 5, 0, 0, -1, 0]], np.float32))
    res = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    t.bind_target(pa.PT_TARGET_RESULT, res.data_ptr())
    t.set_stream(torch.cuda.current_stream().cuda_stream)
    y0, h = D.tile_rows(H, rank, world)
    blocks = [D.tile_rows(H, r, world) for r in range(world)]
    hmax = max(b[1] for b in blocks)
    full = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") if rank == 0 else None

    def frame(k):
        if h > 0:
            t.render(pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=args.spp, max_bounces=args.bounces, tile=(0, y0, W, h)))
        if dist is None:
            return res
        mine = torch.zeros((hmax, W, 4), dtype=torch.float32, device="cuda")     # equal-sized pieces for the gather
        mine[:h] = res[y0:y0 + h]
        pieces = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
        if dist.get_backend() == "nccl":
            dist.gather(mine, pieces, dst=0)
        else:
            host = mine.cpu(); hp = [p.cpu() for p in pieces] if pieces else None
            dist.gather(host, hp, dst=0)
            pieces = [p.cuda() for p in hp] if hp else None
        if rank == 0:
            for (by, bh), p in zip(blocks, pieces):
                full[by:by + bh] = p[:bh]
        return full

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        frame(k)
    fence(); t.stats()
    t0 = time.perf_counter()
    for k in range(args.steps):
        img = frame(args.warmup + k)
    fence()
    dt = time.perf_counter() - t0
    rays = t.stats()["rays"]
    t_all = torch.tensor([dt], dtype=torch.float64, device="cuda"); r_all = torch.tensor([float(rays)], dtype=torch.float64, device="cuda")
    ranks = rank_table(torch, dist, collective, world, rank, local_rank, rays, dt)
    if dist is not None:
        collective(dist.all_reduce, t_all, op=dist.ReduceOp.MAX); collective(dist.all_reduce, r_all, op=dist.ReduceOp.SUM)
    if rank == 0:
        got = img.cpu().numpy().copy()
        t.render(pa.make_params(scene, W, H, seed=pa.frame_seed(args.warmup + args.steps - 1), spp=args.spp, max_bounces=args.bounces))
        t.sync()
        same = bool((res.cpu().numpy().view(np.uint32) == got.view(np.uint32)).all())
        print(json.dumps({
            "metric": "ms/frame, rows split over the GPUs", "value": float(t_all.item()) / args.steps * 1e3, "unit": "ms", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t_all.item()) / args.steps * 1e3, "higher_is_better": False,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} {W}x{H}, {args.bounces} bounces, {args.spp} spp", "parallelism": f"contiguous row blocks x{world}, gathered on rank 0"},
            "mrays_per_s": float(r_all.item()) / float(t_all.item()) / 1e6, "ranks": ranks, **group_view(dist),
            "identical_to_one_gpu_image": same}), flush=True)
    t.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


def c1_gpu(pa, torch, frames=20):
    """Configuration 1 on the GPU, the frame cpu_baseline_c1 times on the host cores: the sphere field on the ground plane
    under the sky, 1920x1080, 8 bounces, 1 spp, one dispatch at a time (tests/test_gpu_configs.py compares every pixel of
    this frame with the oracle's)."""
    scene = pa.Scene("spheres"); sky = pa.procedural_sky(2048, 1024)
    W, H = 1920, 1080
    t = pa.PathTracer(torch.cuda.current_device())
    t.load_scene(scene, sky); t.resize(W, H)
    for k in range(3):
        t.render(pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=1, max_bounces=8))
    t.sync(); t.stats()
    t0 = time.perf_counter()
    for k in range(frames):
        t.render(pa.make_params(scene, W, H, seed=pa.frame_seed(3 + k), spp=1, max_bounces=8))
    t.sync()
    dt = (time.perf_counter() - t0) / frames
    st = t.stats()
    t.close()
    return {"value": st["rays"] / frames / dt / 1e6, "unit": "Mrays/s", "ms_per_frame": dt * 1e3, "frames": frames,
            "config": f"spheres scene ({len(scene.spheres)} spheres + ground plane + sky), {W}x{H}, 8 bounces, 1 spp, 1 GPU, one dispatch at a time"}


def scene_change_time(pa, torch, scene, sky, repeats=4):
    """What a scene change costs on the headline scene, from pt_set_meshes to the first finished frame: the call itself
    (checks and copies the buffers), then a small 1-spp frame that has to wait for the tree (built on the device by
    default).  The reference's counterpart is the buffer re-upload of RebuildMeshObjectBuffers (RayTracingMaster.cs:371-413)."""
    W, H = 320, 180
    t = pa.PathTracer(torch.cuda.current_device())
    t.load_scene(scene, sky); t.resize(W, H)
    p = pa.make_params(scene, W, H, seed=0.5, spp=1, max_bounces=4)
    t.render(p); t.sync()
    best = None
    for k in range(repeats):
        t.sync(); t0 = time.perf_counter()
        t.set_meshes(scene.objects, scene.vertices, scene.indices)
        t1 = time.perf_counter()
        t.render(p); t.sync()
        t2 = time.perf_counter()
        st = t.stats()
        cur = {"set_meshes_ms": (t1 - t0) * 1e3, "first_frame_ms": (t2 - t1) * 1e3, "build_ms": st["bvh_build_ms"], "frame_kernel_ms": st["kernel_ms"]}
        if best is None or cur["set_meshes_ms"] + cur["first_frame_ms"] < best["set_meshes_ms"] + best["first_frame_ms"]:
            best = cur
    t.close()
    best["config"] = f"{scene.name}: pt_set_meshes with the same {scene.n_triangles} triangles, then a {W}x{H} frame of 1 spp; best of {repeats}"
    return best


def moving_camera_rate(pa, tr, scene, args, steps=6):
    """Mrays/s of the headline workload, one dispatch at a time on the context `tr`, with the eye moving 0.02 units per
    step (sideways and forwards): the block order of every dispatch comes from the previous position's costs."""
    W, H = args.width, args.height
    eye0 = np.array(scene.eye, np.float32)
    def eye(k):
        return eye0 + np.array([0.02 * k * 0.6, 0.0, 0.02 * k * 0.8], np.float32)
    tr.render(pa.make_params(scene, W, H, seed=pa.frame_seed(0), spp=args.spp, max_bounces=args.bounces, eye=eye(0)))
    tr.sync(); tr.stats()
    t0 = time.perf_counter()
    for k in range(steps):
        tr.render(pa.make_params(scene, W, H, seed=pa.frame_seed(1 + k), spp=args.spp, max_bounces=args.bounces, eye=eye(1 + k)))
    tr.sync()
    dt = time.perf_counter() - t0
    st = tr.stats()
    return {"value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "config": "the headline workload, one dispatch at a time, the eye advancing 0.02 units per step"}


def stereo_frame_time(pa, torch, frames=30):
    """ms per stereo frame: two 1920x1080 eyes, 1 spp, 4 bounces (the shader's own count), running-mean
    accumulation, one GPU, on the repository's default scene."""
    scene = pa.Scene("xr")
    sky = pa.procedural_sky(2048, 1024)
    eyes = []
    for _ in range(2):
        t = pa.PathTracer(torch.cuda.current_device())
        t.load_scene(scene, sky); t.resize(1920, 1080)
        # each eye's persistent grid takes half of the resident-block slots, so the two dispatches run side by side
        # and each fills the other's ramp-up and tail (measured: 0.75 ms with full grids one after the other)
        t.set_option("blocks_per_cu", int(os.environ.get("STEREO_BLOCKS", "2")))
        eyes.append(t)
    sep = 0.022
    def frame(k):
        for e, t in enumerate(eyes):
            off = np.array([(-0.5 + e) * sep, 0, 0], np.float32)
            p = pa.make_params(scene, 1920, 1080, seed=pa.frame_seed(k), spp=1, max_bounces=4,
                               accumulate=pa.PT_ACCUM_MEAN, sample_index=min(k, 10),
                               eye=scene.eye + off, target=scene.target + off, fov=90.0)
            t.render(p)
    for k in range(3):
        frame(k)
    for t in eyes:
        t.sync()
    t0 = time.perf_counter()
    for k in range(frames):
        frame(3 + k)
    for t in eyes:
        t.sync()
    ms = (time.perf_counter() - t0) / frames * 1e3
    for t in eyes:
        t.close()
    return {"ms_per_frame": ms, "frames": frames, "config": "xr scene, 2 x 1920x1080, 1 spp, 4 bounces, mean accumulation, 1 GPU, one context and stream per eye, "
                      "blocks_per_cu 2 each (the two dispatches overlap)"}


def secondary_scene_rate(pa, torch, sky, frames=5, name="blob70k", ggx=False, reuse=False, fast=False, host_tree=False):
    """Mrays/s on another scene at the headline settings (1920x1080, 8 bounces, 16 spp, one GPU): the ~70 k-triangle
    mesh, or (ggx) the bench scene with every mesh object given a GGX metal surface instead of the shader's
    hard-coded Phong one (PtMaterial, an extension)."""
    scene = pa.Scene(name)
    t = pa.PathTracer(torch.cuda.current_device())
    if host_tree:   # option tree_build 0: the tree made by the host builder instead of on the device (the default from 16 k primitives on)
        t.set_option("tree_build", 0)
    t.load_scene(scene, sky); t.resize(1920, 1080)
    if fast:
        t.set_option("fast_intrinsics", 1)
    if reuse:   # option: a pixel's camera ray (the same for all its samples: the shader has no jitter) is traced once
        t.set_option("reuse_primary_hit", 1)
    if ggx:
        t.set_mesh_materials(pa.materials([((0.0, 0.0, 0.0), 0.8, (0.95, 0.64, 0.54), pa.PT_LOBE_GGX)] * len(scene.objects)))
    t.render(pa.make_params(scene, 1920, 1080, seed=pa.frame_seed(0), spp=16, max_bounces=8)); build_ms = t.stats()["bvh_build_ms"]
    t0 = time.perf_counter()
    for k in range(frames):
        t.render(pa.make_params(scene, 1920, 1080, seed=pa.frame_seed(1 + k), spp=16, max_bounces=8))
    t.sync()
    dt = time.perf_counter() - t0
    st = t.stats()
    t.close()
    return {"value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "ms_per_frame": dt / frames * 1e3, "frames": frames,
            "build_ms": build_ms, "nodes_per_ray": st["nodes_visited"] / max(1, st["rays"]), "bvh_nodes": st["bvh_nodes"],
            "config": f"{name} ({scene.n_triangles} triangles + spheres + plane{', GGX copper on every mesh' if ggx else ''}), "
                      "1920x1080, 8 bounces, 16 spp, 1 GPU, one dispatch at a time"
                      + (", option reuse_primary_hit: value counts only the rays actually traced (same image)" if reuse else "")
                      + (", option fast_intrinsics: approximate hardware transcendentals in Shade(), not bit-exact" if fast else "")
                      + (", option tree_build 0: the tree built by the host's SAH builder (same image)" if host_tree else "")}


def reproj_frame_time(pa, torch, scene, sky, frames=30):
    """ms per 1920x1080 frame (1 spp, 4 bounces, camera moving every frame, history of one camera) in renderMode 1
    (plain), 2 (temporal reprojection: Result -> Result1 copy + 5 history texels per pixel, fused into the path
    kernel) and 0 (the reference's start-up mode: reprojection into Result2, then the a-trous filter and the TAA
    resolve as two streaming kernels)."""
    W, H = 1920, 1080
    t = pa.PathTracer(torch.cuda.current_device())
    t.load_scene(scene, sky); t.resize(W, H)
    def cam(k):
        return (scene.eye[0] + 0.01 * k, scene.eye[1], scene.eye[2] + 0.02 * k)
    # frame parameters and camera-history entries are built before the clock starts: the figure is the library's
    # and the GPU's time per frame, not this script's struct packing
    n_total = frames + 16
    hist_entry = [pa.camera_history_entry(cam(k), scene.target, scene.fov, W / H) for k in range(n_total)]

    def run(mode):
        # the pause modes (11 / 12) follow three frames of renderMode 2 and upload no cameras (RayTracingMaster.cs:513-526)
        prm = [pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=1, max_bounces=4, eye=cam(k),
                              render_mode=2 if (mode >= 11 and k < 3) else mode, accumulation_frames=10) for k in range(n_total)]
        def frame(k):
            if mode != 1 and (mode < 11 or k <= 3):
                t.set_camera_history([hist_entry[k - 1]] if k else [])
            t.render(prm[k])
        t.clear_camera_history()
        for k in range(3):
            frame(k)
        t.sync()
        t0 = time.perf_counter()
        for k in range(frames):
            frame(3 + k)
        t.sync()
        return (time.perf_counter() - t0) / frames * 1e3

    def run_stripes(mode, divisions=10):
        prm = [pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=1, max_bounces=4, eye=cam(k), render_mode=mode,
                              divisions=divisions, counter=k % divisions) for k in range(n_total)]
        old = []
        def frame(k):
            nonlocal old
            if old:
                t.set_camera_history(old)
            t.render(prm[k])
            if k % divisions == 0:
                old = []
            old.append(hist_entry[k])
        t.clear_camera_history()
        for k in range(divisions):
            frame(k)
        t.sync()
        t0 = time.perf_counter()
        for k in range(frames):
            frame(divisions + k)
        t.sync()
        return (time.perf_counter() - t0) / frames * 1e3

    plain = run(1)
    ms2 = run(2)
    ms0 = run(0)
    shown = t.read_result()
    ms5 = run_stripes(5)
    ms6 = run_stripes(6)
    others = {f"mode{m}": run(m) for m in (3, 10, 11, 12)}
    others.update({f"mode{m}": run_stripes(m) for m in (7, 8)})
    t.close()
    return {"ms_per_frame": ms2, "plain_ms_per_frame": plain, "denoise_mode_ms_per_frame": ms0,
            "stripes_ms_per_frame": ms5, "stripes_depth_ms_per_frame": ms6, "frames": frames,
            "other_modes_ms_per_frame": others,
            "denoise_mode_image_mean": float(shown[..., :3].mean()),
            "config": "bench scene, 1920x1080, 1 spp, 4 bounces, _AccumulationFrames 10, 1 GPU; renderMode 2 / 1 / 0; "
                      "renderMode 5 / 6 with 10 stripes (one path-traced per frame, the others reprojected); other_modes: 3 (blur), "
                      "10 (paths only where the history has nothing), 11 / 12 (pause: reprojection only), 7 / 8 (10 stripes into "
                      "Result2, every pixel reprojected; every tenth frame renders the whole image)"}


if __name__ == "__main__":
    main()
