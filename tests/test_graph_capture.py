"""HIP-graph capture of one optimisation step through the C ABI (launch-bound small problems: the authored configs[4]
loop on a 64^2 grid is ~40 launches per step): hf_ray_intersect + hf_adjoint + hf_adam_step (Adam kernel + rebuild of
the acceleration data) issued on a capturing stream, replayed, against the same calls issued eagerly.  The launches
take their scratch from the capture half of the handle's ring and touch no event while capturing (hf_capi.cpp)."""
import ctypes as C
import time

import pytest

pytestmark = pytest.mark.gpu


def _setup(hf, N=64, film=64, spp=4):
    import torch
    from hf_amd import _capi
    from hf_amd.shape import _DIFF_ROWS, _fill, _rows
    dev = torch.device("cuda", 0)
    h0 = hf.workload.sine_heights(N, N, device=dev)
    shape = hf.Heightfield(heightfield=h0.clone(), max_height=0.5)
    rays = hf.workload.ortho_rays(film, film, spp, dev, origin=(0.6, 0.35, 2.0), target=(0.0, 0.0, 0.25), scale=(0.9, 0.9, 1.0))
    R = rays.shape[1]
    st = dict(t=torch.empty(R, device=dev), uv=torch.empty((2, R), device=dev), prim=torch.empty(R, dtype=torch.int32, device=dev),
              si=torch.empty((18, R), device=dev), gsi=torch.zeros((18, R), device=dev), grad=torch.zeros((N, N), device=dev),
              m=torch.zeros((N, N), device=dev), v=torch.zeros((N, N), device=dev))
    st["gsi"][0] = 1.0
    lib = _capi.lib()
    r_s = shape._rays_struct(rays[0:3], rays[3:6], rays[6]); pi_s = shape._pi_struct(st["t"], st["uv"], st["prim"])
    si_s = _fill(_capi.hf_si_t(), _DIFF_ROWS, _rows(st["si"], R)); g_s = _fill(_capi.hf_si_grad_t(), _DIFF_ROWS, _rows(st["gsi"], R))
    flags = int(hf.RayFlags.All)
    hd = shape.heightfield

    def step(stream):
        _capi.check(lib.hf_ray_intersect(shape._h, R, C.byref(r_s), flags, None, C.byref(pi_s), C.byref(si_s), stream))
        st["grad"].zero_()
        _capi.check(lib.hf_adjoint(shape._h, R, C.byref(r_s), C.byref(pi_s), flags, None, C.byref(g_s), st["grad"].data_ptr(),
                                   None, None, stream))
        # (the step number is a host scalar baked into a captured launch: both paths use step 1 every time)
        _capi.check(lib.hf_adam_step(shape._h, hd.data_ptr(), st["grad"].data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(),
                                     0.002, 0.9, 0.999, 1e-8, 1, 0, stream))

    def reset():
        hd.copy_(h0); st["m"].zero_(); st["v"].zero_()
        shape.parameters_changed(["heightfield"])
        torch.cuda.synchronize()
    return torch, dev, shape, st, hd, step, reset, (rays, r_s, pi_s, si_s, g_s)


def test_one_step_captured_and_replayed_equals_eager(hf):
    torch, dev, shape, st, hd, step, reset, keep = _setup(hf)
    K = 12
    reset()
    t0 = time.perf_counter()
    for _ in range(K):
        step(torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    eager_s = (time.perf_counter() - t0) / K
    ref_h, ref_t = hd.clone(), st["t"].clone()

    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        step(s.cuda_stream)                         # warm-up on a side stream, as torch asks before a capture
    torch.cuda.current_stream(dev).wait_stream(s)
    reset()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(torch.cuda.current_stream(dev).cuda_stream)
    reset()                                         # what the capture did to the buffers is undone; replays do the work
    t0 = time.perf_counter()
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    graph_s = (time.perf_counter() - t0) / K
    assert torch.equal(st["t"], ref_t) or float((st["t"] - ref_t).abs()[torch.isfinite(ref_t)].max()) < 1e-4
    assert float((hd - ref_h).abs().max()) < 2e-5, float((hd - ref_h).abs().max())      # float atomics order only
    assert float((ref_h - hf.workload.sine_heights(64, 64, device=dev)).abs().max()) > 1e-3   # the steps did move the heights
    print(f"per step: eager {1e6 * eager_s:.0f} us, graph replay {1e6 * graph_s:.0f} us")
    # eager launches after the capture still work (separate halves of the scratch ring)
    step(torch.cuda.current_stream(dev).cuda_stream); torch.cuda.synchronize()


def test_capture_ring_refuses_the_33rd_captured_launch_and_reset_returns_the_blocks(hf):
    """Every captured trace launch keeps its scratch block for the replays: the 33rd on one handle is refused with
    HF_EINVAL (it would share work counters with the first), hf_capture_reset hands the blocks back (hf.h, HIP graphs)."""
    from hf_amd import _capi
    torch, dev, shape, st, hd, step, reset, keep = _setup(hf)
    rays, r_s, pi_s, si_s, g_s = keep
    lib = _capi.lib()
    R = rays.shape[1]
    _capi.check(lib.hf_capture_reset(shape._h))
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    rcs = []
    with torch.cuda.graph(g, stream=s):
        for _ in range(34):
            rcs.append(lib.hf_ray_intersect_preliminary(shape._h, R, C.byref(r_s), None, C.byref(pi_s), s.cuda_stream))
    assert rcs[:32] == [_capi.HF_OK] * 32
    assert rcs[32] == _capi.HF_EINVAL and rcs[33] == _capi.HF_EINVAL
    assert b"captured trace launches" in lib.hf_last_error_string()
    st["t"].fill_(-1.0)
    g.replay(); torch.cuda.synchronize()
    ref = shape.ray_intersect_preliminary(hf.Ray3f(rays[0:3], rays[3:6], rays[6])).t
    assert torch.equal(st["t"], ref)                       # the 32 captured launches replay correctly
    del g
    _capi.check(lib.hf_capture_reset(shape._h))
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        rc = lib.hf_ray_intersect_preliminary(shape._h, R, C.byref(r_s), None, C.byref(pi_s), s.cuda_stream)
    assert rc == _capi.HF_OK
    st["t"].fill_(-1.0)
    g2.replay(); torch.cuda.synchronize()
    assert torch.equal(st["t"], ref)


def test_scratch_ring_under_48_threads(hf, oracle):
    """more host threads than eager scratch slots (32).  Every thread traces its own small wavefronts on
    its own stream, many times; each result must be the single-threaded one (a launch that lost its work counters to
    another leaves rays untraced)."""
    import threading
    import numpy as np
    torch, dev, shape, st, hd, step, reset, keep = _setup(hf)
    reset()
    nthreads, reps, n = 48, 40, 4096
    rays_all = hf.workload.ortho_rays(64, 64, 48, dev, origin=(0.6, 0.35, 2.0), target=(0.0, 0.0, 0.25), scale=(0.9, 0.9, 1.0))
    ref = shape.ray_intersect_preliminary(hf.Ray3f(rays_all[0:3], rays_all[3:6], rays_all[6]))
    torch.cuda.synchronize()
    errs = []

    def work(k):
        try:
            s = torch.cuda.Stream(dev)
            r = rays_all[:, k * n:(k + 1) * n].contiguous()
            with torch.cuda.stream(s):
                for _ in range(reps):
                    pi = shape.ray_intersect_preliminary(hf.Ray3f(r[0:3], r[3:6], r[6]))
                    hit = shape.ray_test(hf.Ray3f(r[0:3], r[3:6], r[6]))
                s.synchronize()
            if not (torch.equal(pi.t, ref.t[k * n:(k + 1) * n]) and torch.equal(pi.prim_index, ref.prim_index[k * n:(k + 1) * n])
                    and torch.equal(hit, torch.isfinite(ref.t[k * n:(k + 1) * n]))):
                errs.append(k)
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
