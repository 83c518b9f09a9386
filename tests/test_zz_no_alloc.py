"""pt_trace_rays_static - the ray query that allocates nothing - against pt_trace_rays, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import pathtracing_amd as pa

pytestmark = pytest.mark.gpu

MAX_RAYS = 32768      # PT_TRACE_RAYS_STATIC_MAX


def _rays(rng, n, far=False):
    od = np.zeros((n, 6), np.float32)
    od[:, :3] = rng.uniform(-9, 9, (n, 3)); od[:, 1] = np.abs(od[:, 1]) + 0.1
    if far:
        od[:, :3] *= 1e4
    d = rng.normal(0, 1, (n, 3)); d[:, 1] -= 0.3
    od[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    return od


def _free_device_memory():
    hip = C.CDLL("libamdhip64.so")
    free = C.c_size_t(); total = C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _allocation_calls():
    """hipMalloc / hipFree / hipHostMalloc / hipHostFree calls the library has made so far (it routes every one through a counter)."""
    from pathtracing_amd import _ffi
    return _ffi.ptrace().pt_allocation_calls()


@pytest.mark.parametrize("width", [4, 2, 8])
def test_static_ray_query_equals_the_allocating_one(width):
    rng = np.random.default_rng(7 + width)
    t = pa.PathTracer(0)
    try:
        t.set_option("bvh_width", width)
        s = pa.Scene("blob70k")
        t.load_scene(s, pa.procedural_sky(64, 32))
        first, second = _rays(rng, 20000), _rays(rng, 777)            # two calls in a row, the second smaller
        want = [t.trace_rays(od) for od in (first, second)]           # the existing entry point (it also builds the scene)
        assert (want[0][0] > 0).mean() > 0.05 and (want[0][0] == 0).any() and (want[0][0] == -1).any()
        got0 = t.trace_rays_static(first)
        before, calls = _free_device_memory(), _allocation_calls()
        got1 = t.trace_rays_static(second)
        assert _allocation_calls() == calls                           # not one allocation or release call was made
        assert _free_device_memory() == before
        # the control: the allocating entry point makes such calls and leaves free memory as it was, which is why the
        # free-memory figure alone proves nothing
        t.trace_rays(second)
        assert _allocation_calls() >= calls + 6 and _free_device_memory() == before
        for (gi, gt), (wi, wt) in zip((got0, got1), want):
            assert (gi == wi).all() and (gt.view(np.uint32) == wt.view(np.uint32)).all()
        # the exhaustive loop through the same entry point, and the largest call it takes
        bi, bt = t.trace_rays_static(second, brute_force=True)
        assert (bi == want[1][0]).all() and (bt.view(np.uint32) == want[1][1].view(np.uint32)).all()
        big = _rays(rng, MAX_RAYS)
        wi, wt = t.trace_rays(big); gi, gt = t.trace_rays_static(big)
        assert (gi == wi).all() and (gt.view(np.uint32) == wt.view(np.uint32)).all()
        gi, gt = t.trace_rays_static(second)                          # and a small one after the largest
        assert (gi == want[1][0]).all() and (gt.view(np.uint32) == want[1][1].view(np.uint32)).all()
    finally:
        t.close()


def test_static_ray_query_refuses_what_would_need_an_allocation():
    rng = np.random.default_rng(3)
    t = pa.PathTracer(0)
    try:
        s = pa.Scene("fixture")
        t.load_scene(s, None)
        with pytest.raises(pa.PtraceError, match="built"):            # nothing has built the scene yet
            t.trace_rays_static(_rays(rng, 10))
        t.trace_rays(_rays(rng, 10))
        ids, ts = t.trace_rays_static(_rays(rng, 10))
        assert len(ids) == 10
        with pytest.raises(pa.PtraceError, match="at most 32768"):
            t.trace_rays_static(_rays(rng, MAX_RAYS + 1))
        with pytest.raises(pa.PtraceError, match="farther"):          # would need the sphere boxes re-inflated: a rebuild
            t.trace_rays_static(_rays(rng, 10, far=True))
        assert len(t.trace_rays_static(np.zeros((0, 6), np.float32))[0]) == 0
    finally:
        t.close()


def test_overlapping_static_queries_on_one_gpu_are_refused_not_mixed():
    """Every context on a GPU shares the one scratch array: while a call holds it, a second call - here from another thread
    and another context - returns PT_ERR_INVALID ("in use") instead of overwriting the first call's rays; afterwards the
    array is free again.  Both calls' results, when they do run, are right."""
    import threading
    rng = np.random.default_rng(9)
    s = pa.Scene("blob70k")
    ts = [pa.PathTracer(0) for _ in range(2)]
    try:
        for t in ts:
            t.load_scene(s, None)
        rays = [_rays(rng, MAX_RAYS), _rays(rng, MAX_RAYS)]
        want = [t.trace_rays(r) for t, r in zip(ts, rays)]
        refused, wrong, done = [0, 0], [0, 0], [0, 0]

        def worker(k):
            for _ in range(60):
                try:
                    gi, gt = ts[k].trace_rays_static(rays[k])
                except pa.PtraceError as e:
                    assert "in use" in str(e)
                    refused[k] += 1
                    continue
                done[k] += 1
                if not ((gi == want[k][0]).all() and (gt.view(np.uint32) == want[k][1].view(np.uint32)).all()):
                    wrong[k] += 1

        th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert wrong == [0, 0] and min(done) > 0, (wrong, done, refused)
        gi, gt = ts[0].trace_rays_static(rays[0])                     # nobody holds the array any more
        assert (gi == want[0][0]).all()
    finally:
        for t in ts:
            t.close()
