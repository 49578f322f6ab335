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
    a = ap.parse_args()
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
