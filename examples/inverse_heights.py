#!/usr/bin/env python3
"""Authored stand-in for BASELINE.json configs[4] ("inverse loop: 100 Adam steps recovering heights
from multi-light renders"; the notebook in the reference snapshot contains no such loop -- SURVEY 0).

A minimal direct-lighting differentiable render on top of the heightfield shape:
    primary rays (orthographic)  ->  shape.ray_intersect (HIP traversal + fused SI)
    image_k = box-filtered  albedo/pi * E * max(0, <n, l_k>)  for K directional lights (hf_direct_lighting),  depth = t
    loss = sum_k |image_k - target_k|^2        (multi-light renders only; --depth-weight adds a depth term)
    loss.backward()  ->  HIP adjoint scatters dL/dheight;  hf_amd.Adam.step() = hf_adam_step: Adam update on the
        device + rebuild of the acceleration data (what params.update / scene.parameters_changed do once per
        optimiser step, util.py:185-232, scene.cpp:343-385)
Geometry is attached (prb-style).  --silhouette adds the discontinuity term the way prb_reparam.py:317-366 does for
the camera ray: the primary rays go through hf_amd.reparameterize_ray (identity in primal mode), the intersection is
differentiated w.r.t. the reparameterised direction as well, and every sample is weighted by the determinant
(hf_direct_lighting's per-sample weight row: direct_reparam.py:164-180).

    python examples/inverse_heights.py [--grid 128 --film 256 --steps 100]
Sharded (BASELINE configs[3] / [4]): under torchrun (torch.distributed initialised by this script) rank r renders the
image tiles hf_amd.workload.partition_tiles(film, film, world)[r] of the ONE wavefront -- targets sliced the same
way, the image loss normalised by the pixel count of the whole film so that the ranks' losses add up to the
single-rank loss -- and the gradient texture is summed with one all-reduce per step; the Adam step is replicated.
--virtual-ranks V runs the same partition on one device, shard after shard (what tests/test_gpu_inverse_loop.py
compares with the unsharded loop).
"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hf_amd  # noqa: E402

LIGHTS = torch.tensor([[0.5, 0.2, 0.84], [-0.5, 0.3, 0.81], [0.1, -0.6, 0.79], [0.0, 0.0, 1.0]])


def render_reparameterized(shape, ray, lights, spp, aux=8, kappa=2e4, seed=0, ray_index=None):
    """primary rays through reparameterize_ray; per-sample diffuse shading x determinant and the box film in
    hf_direct_lighting (weight row = the determinant); the gradient reaches the heights through hf_adjoint (shading
    normal), the auxiliary rays (reparameterised direction) and the determinant (weight gradient)"""
    d, det = hf_amd.reparameterize_ray(shape, ray, num_rays=aux, kappa=kappa, exponent=3.0, seed=seed,
                                       ray_index=ray_index)
    ray2 = hf_amd.Ray3f(ray.o, d, ray.maxt)
    si = shape.ray_intersect(ray2, hf_amd.RayFlags.All)
    valid = si.is_valid()
    images = hf_amd.direct_lighting(si, ray2, lights, albedo=1.0, spp=spp, weight=det)
    depth = torch.where(valid, si.t, torch.zeros_like(si.t))
    return images, depth, valid


def render(shape, ray, lights, spp, shadows=False, silhouette=False, aux=8, kappa=2e4, film=None, ray_index=None):
    """film: None = box filter (pixel = mean of its samples, inside hf_direct_lighting); (positions [2, n], width,
    height) = the reference's default Gaussian reconstruction filter (hf_film_splat) on the per-sample values"""
    if silhouette:
        return render_reparameterized(shape, ray, lights, spp, aux, kappa, ray_index=ray_index)
    si = shape.ray_intersect(ray, hf_amd.RayFlags.All)
    valid = si.is_valid()
    if film is not None:
        samples = hf_amd.direct_lighting(si, ray, lights, albedo=1.0, spp=1)          # [K, n]
        images = hf_amd.film_gaussian(samples, film[0], film[1], film[2])
        return images, torch.where(valid, si.t, torch.zeros_like(si.t)), valid
    vis = None
    if shadows:  # detached visibility of each light: one any-hit launch per light (scene.cpp:290-293)
        with torch.no_grad():
            vis = torch.stack([~shape.ray_test(si.spawn_ray(l[:3])) for l in lights]).to(torch.uint8)
    # diffuse direct lighting + box-filter film on the wavefront (hf_direct_lighting): [K, pixels]
    images = hf_amd.direct_lighting(si, ray, lights, albedo=1.0, spp=spp, vis=vis)
    depth = torch.where(valid, si.t, torch.zeros_like(si.t))
    return images, depth, valid


def centred_error(h, target):
    """mean |h - h*| after removing the mean offset: shading under directional lights observes the surface
    gradient, not its absolute height (the photometric-stereo ambiguity)"""
    d = h - target
    return float((d - d.mean()).abs().mean())


def run(grid=128, film=256, spp=1, steps=100, lr=0.02, device="cuda", verbose=True, seed=0, shadows=False,
        depth_weight=0.0, silhouette=False, aux=8, kappa=2e4, gaussian_film=False, virtual_ranks=0, record=None):
    """record (optional list): receives the height texture after every step (trajectory comparisons in the tests)"""
    import torch.distributed as dist
    dev = torch.device(device)
    lights = torch.cat([LIGHTS / LIGHTS.norm(dim=1, keepdim=True), torch.full((len(LIGHTS), 1), math.pi)], 1)  # E = pi
    target_h = hf_amd.workload.sine_heights(grid, grid, device=dev)
    # ---- the shards this process renders: its rank's image tiles under torch.distributed, all V shards one after
    # the other with --virtual-ranks V, else the whole film ----
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if distributed:
        assert not gaussian_film, "the Gaussian film splats across tile borders: not sharded"
        shard_pixels = [hf_amd.workload.partition_tiles(film, film, dist.get_world_size())[dist.get_rank()]]
    elif virtual_ranks and virtual_ranks > 1:
        assert not gaussian_film
        shard_pixels = hf_amd.workload.partition_tiles(film, film, virtual_ranks)
    else:
        shard_pixels = [None]
    npix_total = film * film
    target = hf_amd.Heightfield(heightfield=target_h, max_height=0.5)
    shards = []
    for pixels in shard_pixels:
        # camera looking down at 30 degrees off vertical so that every ray meets the surface
        rays = hf_amd.workload.ortho_rays(film, film, spp, dev, seed=seed, origin=(0.6, 0.35, 2.0),
                                          target=(0.0, 0.0, 0.25), scale=(0.95, 0.95, 1.0), pixels=pixels)
        ray = hf_amd.Ray3f(rays[0:3], rays[3:6], rays[6])
        flm = (hf_amd.workload.film_positions(film, film, spp, dev, seed=seed), film, film) if gaussian_film else None
        with torch.no_grad():
            tgt_img, tgt_depth, tgt_valid = render(target, ray, lights, spp, shadows, film=flm)
        rid = hf_amd.workload.ray_indices(film, film, spp, dev, pixels) if silhouette else None
        shards.append((ray, flm, tgt_img, tgt_depth, tgt_valid, rid))
    shape = hf_amd.Heightfield(heightfield=torch.full_like(target_h, 0.5), max_height=0.5)
    shape.heightfield.requires_grad_(True)
    opt = hf_amd.Adam(shape, lr=lr)                       # hf_adam_step: optimizers.py:263-300 + params.update
    hist = []
    t0 = time.perf_counter()
    n_rays = sum(len(sh[0]) for sh in shards)
    for it in range(steps):
        opt.zero_grad()
        total = 0.0
        for ray, flm, tgt_img, tgt_depth, tgt_valid, rid in shards:    # (backward accumulates into heightfield.grad)
            images, depth, valid = render(shape, ray, lights, spp, shadows, silhouette, aux, kappa, film=flm,
                                          ray_index=rid)
            both = valid & tgt_valid
            # the multi-light renders only (configs[4]); mean over the pixels of the WHOLE film: shard losses add up
            loss = ((images - tgt_img) ** 2).sum() / npix_total
            # (the depth term is normalised per FILM sample -- npix_total * spp -- since round 3, so that the shards'
            # terms add up; rounds 1-2 divided by the number of valid samples: the same --depth-weight weighs less now)
            if depth_weight > 0:                                    # optional extra supervision, off by default
                loss = loss + depth_weight * (((depth - tgt_depth) ** 2) * both).sum() / (npix_total * spp)
            loss.backward()
            total = total + loss.detach()
        if distributed:                                             # image loss and gradient texture: sums over ranks
            dist.all_reduce(total)
        hf_amd.allreduce_gradient(shape.heightfield.grad)
        opt.step()                                            # Adam update + rebuild of the acceleration data (replicated)
        hist.append(float(total))
        if record is not None:
            record.append(shape.heightfield.detach().clone())
        if it == 0:  # the first step pays the one-off costs (code-object load, allocator warm-up)
            torch.cuda.synchronize()
            t_first = time.perf_counter() - t0
        if verbose and (it % 10 == 0 or it == steps - 1):
            err = float((shape.heightfield.detach() - target_h).abs().mean())
            print(f"step {it:4d}  loss {hist[-1]:.6f}  mean |h - h*| {err:.5f}  (offset removed: "
                  f"{centred_error(shape.heightfield.detach(), target_h):.5f})", flush=True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    err = float((shape.heightfield.detach() - target_h).abs().mean())
    if verbose:
        print(f"{steps} Adam steps, {n_rays} rays/step on this process: {wall:.2f} s wall-clock end to end "
              f"(first step {1e3 * t_first:.0f} ms, then {1e3 * (wall - t_first) / max(1, steps - 1):.2f} ms/step)")
    run.last_centred_error = centred_error(shape.heightfield.detach(), target_h)
    run.start_centred_error = centred_error(torch.full_like(target_h, 0.5), target_h)
    return hist, err, wall


def run_captured(grid=128, film=256, spp=1, steps=100, lr=0.02, device="cuda", seed=0, record=None):
    """The image-loss loop of run() (box film, no shadows, attached geometry) with ONE optimisation step captured into a
    HIP graph and replayed `steps` times: trace -> direct lighting -> loss and its image gradient -> lighting adjoint ->
    hf_adjoint -> hf_adam_step_scheduled (update + rebuild).  No autograd, no allocation, no host work per step: the
    step sizes of all steps sit in a device table (hf_adam_lr_t: hf_adam_step's own arithmetic), a device counter picks
    the entry, the losses go to a device array.  Same kernels on the same inputs as run(): the trajectories agree to
    the order of the float atomics (tests/test_gpu_inverse_loop.py).
    Returns (loss history, mean |h - h*|, timing): timing = wall-clock of the replays, HIP-event time of the replays
    (the kernels of a step back to back), and both per step."""
    import ctypes as C
    from hf_amd import _capi
    from hf_amd.shape import _DIFF_ROWS, _fill, _rows
    dev = torch.device(device)
    lib = _capi.lib()
    lights = torch.cat([LIGHTS / LIGHTS.norm(dim=1, keepdim=True), torch.full((len(LIGHTS), 1), math.pi)], 1)
    K = lights.shape[0]
    target_h = hf_amd.workload.sine_heights(grid, grid, device=dev)
    target = hf_amd.Heightfield(heightfield=target_h, max_height=0.5)
    rays = hf_amd.workload.ortho_rays(film, film, spp, dev, seed=seed, origin=(0.6, 0.35, 2.0),
                                      target=(0.0, 0.0, 0.25), scale=(0.95, 0.95, 1.0))
    ray = hf_amd.Ray3f(rays[0:3], rays[3:6], rays[6])
    with torch.no_grad():
        tgt_img, _, _ = render(target, ray, lights, spp)
    tgt_img = tgt_img.contiguous()
    R, npix = rays.shape[1], film * film
    shape = hf_amd.Heightfield(heightfield=torch.full_like(target_h, 0.5), max_height=0.5)
    h = shape.heightfield
    m, v, grad_h = torch.zeros_like(h), torch.zeros_like(h), torch.zeros_like(h)
    t = torch.empty(R, device=dev); uv = torch.empty((2, R), device=dev); prim = torch.empty(R, dtype=torch.int32, device=dev)
    si = torch.empty((18, R), device=dev); gsi = torch.zeros((18, R), device=dev)
    images = torch.empty((K, npix), device=dev); gimg = torch.empty((K, npix), device=dev)
    hist = torch.zeros(steps + 1, device=dev)
    idx = torch.zeros(1, dtype=torch.int64, device=dev)
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    b1, b2, eps = 0.9, 0.999, 1e-8
    lr_tab = torch.tensor([lib.hf_adam_lr_t(lr, b1, b2, k + 1) for k in range(steps + 1)], dtype=torch.float32, device=dev)
    r_s = shape._rays_struct(rays[0:3], rays[3:6], rays[6]); pi_s = shape._pi_struct(t, uv, prim)
    si_s = _fill(_capi.hf_si_t(), _DIFF_ROWS, _rows(si, R)); g_s = _fill(_capi.hf_si_grad_t(), _DIFF_ROWS, _rows(gsi, R))
    rows_si, rows_g = _rows(si, R), _rows(gsi, R)
    shn = (C.c_void_p * 3)(*rows_si[9:12]); gnp = (C.c_void_p * 3)(*rows_g[9:12])
    dd = (C.c_void_p * 3)(r_s.d[0], r_s.d[1], r_s.d[2])
    L = (_capi.hf_dir_light_t * K)()
    for k, (x, y, z, e) in enumerate(lights.tolist()):
        L[k].to_light[0], L[k].to_light[1], L[k].to_light[2], L[k].irradiance = x, y, z, e
    flags = int(hf_amd.RayFlags.All)

    def step(stream):
        grad_h.zero_()
        _capi.check(lib.hf_ray_intersect(shape._h, R, C.byref(r_s), flags, None, C.byref(pi_s), C.byref(si_s), stream))
        _capi.check(lib.hf_direct_lighting(R, spp, C.byref(shn), C.byref(dd), rows_si[0], K, L, 1.0, None, images.data_ptr(), stream))
        diff = images - tgt_img
        hist.scatter_(0, idx, ((diff * diff).sum() / npix).reshape(1))
        torch.mul(diff, 2.0 / npix, out=gimg)
        _capi.check(lib.hf_direct_lighting_adjoint(R, spp, C.byref(shn), C.byref(dd), rows_si[0], K, L, 1.0, None, gimg.data_ptr(),
                                                   C.byref(gnp), stream))
        _capi.check(lib.hf_adjoint(shape._h, R, C.byref(r_s), C.byref(pi_s), flags, None, C.byref(g_s), grad_h.data_ptr(),
                                   None, None, stream))
        _capi.check(lib.hf_adam_step_scheduled(shape._h, h.data_ptr(), grad_h.data_ptr(), m.data_ptr(), v.data_ptr(),
                                               lr_tab.data_ptr(), ctr.data_ptr(), b1, b2, eps, 0, stream))
        idx.add_(1)

    # warm-up outside the capture (code objects load at the first launch), then the state back to step 0
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        step(side.cuda_stream)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    with torch.no_grad():
        h.fill_(0.5)
    m.zero_(); v.zero_(); ctr.zero_(); idx.zero_(); hist.zero_()
    _capi.check(lib.hf_set_heights(shape._h, h.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(torch.cuda.current_stream(dev).cuda_stream)
    # (the capture itself executed nothing)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for it in range(steps):
        graph.replay()
        if record is not None:
            record.append(h.detach().clone())
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1)
    _capi.check(lib.hf_capture_reset(shape._h))
    err = float((h.detach() - target_h).abs().mean())
    run_captured.last_centred_error = centred_error(h.detach(), target_h)
    return hist[:steps].tolist(), err, {"wall_clock_s": wall, "gpu_ms_total": gpu_ms, "wall_ms_per_step": 1e3 * wall / steps,
                                        "gpu_ms_per_step": gpu_ms / steps}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--film", type=int, default=256)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--lr", type=float, default=0.02)
    ap.add_argument("--shadows", action="store_true", help="shadow rays towards every light (one ray_test per light)")
    ap.add_argument("--depth-weight", type=float, default=0.0, help="weight of an extra depth term (0 = images only)")
    ap.add_argument("--silhouette", action="store_true", help="reparameterised primary rays (discontinuity term)")
    ap.add_argument("--aux", type=int, default=8, help="auxiliary rays per primary ray of --silhouette")
    ap.add_argument("--gaussian-film", action="store_true", help="Gaussian reconstruction filter instead of the box film")
    ap.add_argument("--virtual-ranks", type=int, default=0, help="render the tile partition of V ranks on this one device")
    ap.add_argument("--captured", action="store_true", help="one step captured into a HIP graph and replayed (run_captured)")
    a = ap.parse_args()
    if a.captured:
        hist, err, tm = run_captured(a.grid, a.film, a.spp, a.steps, a.lr)
        print(f"loss {hist[0]:.6f} -> {hist[-1]:.6f}, mean |h - h*| {err:.5f}; {a.steps} replayed steps: {tm['wall_clock_s']:.3f} s wall-clock, "
              f"{tm['wall_ms_per_step']:.3f} ms per step ({tm['gpu_ms_per_step']:.3f} ms on the GPU)")
        sys.exit(0)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:   # torchrun: one process per GPU, RCCL (HF_BENCH_BACKEND=gloo: ranks may share a device)
        import torch.distributed as dist
        backend = os.environ.get("HF_BENCH_BACKEND", "nccl")
        local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=world)
    run(a.grid, a.film, a.spp, a.steps, a.lr, shadows=a.shadows, depth_weight=a.depth_weight, silhouette=a.silhouette,
        aux=a.aux, gaussian_film=a.gaussian_film, virtual_ranks=a.virtual_ranks,
        verbose=int(os.environ.get("RANK", "0")) == 0)
    if world > 1:
        dist.destroy_process_group()
