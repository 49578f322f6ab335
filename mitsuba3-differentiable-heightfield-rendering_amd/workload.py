"""Synthetic workload of SURVEY.md section 8(d): procedural sine heightfield and the
ray wavefront of an oblique orthographic sensor with per-sample jitter.

Restates (for input generation only, outside any timed region):
  wavefront index -> pixel       src/render/integrator.cpp:251-268
  sample position                src/render/integrator.cpp:377-399
  orthographic sample_ray        src/sensors/orthographic.cpp:119-142
  orthographic_projection        include/mitsuba/render/sensor.h:266-301
  look_at                        include/mitsuba/core/transform.h:254-282
  sample_tea_32                  include/mitsuba/core/random.h:76-91 (sampler seeding,
                                 src/render/sampler.cpp:116-130; the PCG32 stream it
                                 seeds lives in Dr.Jit, so the two TEA words are used
                                 directly as the 2-D jitter -- documented stand-in)
"""
import math

import numpy as np
import torch


def sine_heights(width, height=None, device="cpu"):
    """h = 0.5 + 0.25 sin(2 pi fx u) cos(2 pi fy v) + 0.125 sin(2 pi 7 (u+v)),
    fx = fy = 4 at N=64, scaled with N/64 up to 32."""
    height = height or width
    f = float(min(32.0, max(1.0, 4.0 * max(width, height) / 64.0)))
    u = torch.arange(width, dtype=torch.float64, device=device) / (width - 1)
    v = torch.arange(height, dtype=torch.float64, device=device) / (height - 1)
    U, V = u[None, :], v[:, None]
    h = 0.5 + 0.25 * torch.sin(2 * math.pi * f * U) * torch.cos(2 * math.pi * f * V) \
        + 0.125 * torch.sin(2 * math.pi * 7.0 * (U + V))
    return h.to(torch.float32)


def look_at(origin, target, up):
    o, t, u = (np.asarray(a, np.float64) for a in (origin, target, up))
    d = t - o
    d /= np.linalg.norm(d)
    left = np.cross(u, d)
    left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m


def tea32(v0, v1, rounds=4):
    """sample_tea_32 on int64 tensors holding uint32 values"""
    M = 0xFFFFFFFF
    s = 0
    for _ in range(rounds):
        s = (s + 0x9E3779B9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xA341316C) ^ (v1 + s) ^ ((v1 >> 5) + 0xC8013EA4))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xAD90777D) ^ (v0 + s) ^ ((v0 >> 5) + 0x7E95761E))) & M
    return v0, v1


def partition_tiles(film_w, film_h, world, tile=32):
    """Image-tile partition of one wavefront over `world` ranks (BASELINE configs[3]: "pixel-tiled across
    8xMI355X"): the film is cut into tile x tile pixel blocks (MI_BLOCK_SIZE = 32, src/render/integrator.cpp:
    130-140) in row-major block order and block b goes to rank b % world -- interleaved, so every rank sees the
    same mix of empty and expensive image regions.  Returns one int64 tensor of pixel indices (y * film_w + x,
    the wavefront's index>>log2(spp) of integrator.cpp:251-268) per rank, block after block, row-major inside
    a block; the lists are disjoint and cover the film (tests/test_partition.py)."""
    bx, by = (film_w + tile - 1) // tile, (film_h + tile - 1) // tile
    yy, xx = torch.meshgrid(torch.arange(film_h), torch.arange(film_w), indexing="ij")
    block = (yy // tile) * bx + (xx // tile)                       # block id of every pixel
    pix = (yy * film_w + xx).reshape(-1)
    block = block.reshape(-1)
    # stable sort by (block, then row-major inside the block): pixel order within a rank's list
    order = torch.argsort(block * (tile * tile) + ((yy % tile) * tile + (xx % tile)).reshape(-1), stable=True)
    pix, block = pix[order], block[order]
    return [pix[(block % world) == r].contiguous() for r in range(world)]


def ortho_rays(film_w, film_h, spp, device, start=0, count=None, seed=0,
               origin=(1.5, 1.5, 1.5), target=(0.0, 0.0, 0.125), up=(0.0, 0.0, 1.0),
               scale=(1.6, 1.6, 1.0), near=1e-2, far=1e4, chunk=1 << 24, pixels=None):
    """Rays [start, start+count) of the W*H*spp wavefront as a [7, count] float32 tensor
    (ox,oy,oz,dx,dy,dz,maxt).  With `pixels` (int64 tensor of pixel indices, e.g. one rank's list from
    partition_tiles) the result is instead the spp samples of exactly those pixels, pixel after pixel: ray
    k*spp + s is wavefront index pixels[k]*spp + s -- the same rays, bit for bit, as in the full wavefront."""
    total = film_w * film_h * spp
    if pixels is not None:
        pixels = pixels.to(device=device, dtype=torch.int64)
        start, count = 0, int(pixels.numel()) * spp
    count = total - start if count is None else count
    to_world = look_at(origin, target, up) @ np.diag([scale[0], scale[1], scale[2], 1.0])
    tw = torch.tensor(to_world, dtype=torch.float64, device=device)
    d = tw[:3, 2] / torch.linalg.norm(tw[:3, 2])
    aspect = film_w / film_h
    out = torch.empty((7, count), dtype=torch.float32, device=device)
    log_spp = int(math.log2(spp)) if (spp & (spp - 1)) == 0 else None
    for c0 in range(0, count, chunk):
        c1 = min(count, c0 + chunk)
        idx = torch.arange(start + c0, start + c1, dtype=torch.int64, device=device)
        if pixels is not None:   # local index -> index in the full wavefront
            idx = pixels[idx // spp] * spp + idx % spp
        v0, v1 = tea32(torch.full_like(idx, seed), idx)
        jx = (v0 >> 9).to(torch.float64) * (1.0 / (1 << 23))
        jy = (v1 >> 9).to(torch.float64) * (1.0 / (1 << 23))
        pix = (idx >> log_spp) if log_spp is not None else (idx // spp)
        py = pix // film_w
        px = pix - py * film_w
        sx = (px.to(torch.float64) + jx) / film_w
        sy = (py.to(torch.float64) + jy) / film_h
        cx, cy = 1.0 - 2.0 * sx, (1.0 - 2.0 * sy) / aspect
        for k in range(3):
            out[k, c0:c1] = (tw[k, 0] * cx + tw[k, 1] * cy + tw[k, 2] * near + tw[k, 3]).to(torch.float32)
            out[3 + k, c0:c1] = float(d[k])
        out[6, c0:c1] = far - near
    return out


def ray_indices(film_w, film_h, spp, device, pixels=None):
    """Index in the FULL W*H*spp wavefront of every ray of ortho_rays(film_w, film_h, spp, ..., pixels) as an int32
    tensor: the `ray_index` of reparameterize_ray / the `ray_id` of hf_reparam_*, so that a rank's tiles draw the
    auxiliary samples the unpartitioned wavefront draws for the same rays."""
    if pixels is None:
        return torch.arange(film_w * film_h * spp, dtype=torch.int32, device=device)
    pixels = pixels.to(device=device, dtype=torch.int64)
    idx = torch.arange(int(pixels.numel()) * spp, dtype=torch.int64, device=device)
    return (pixels[idx // spp] * spp + idx % spp).to(torch.int32)


def film_positions(film_w, film_h, spp, device, seed=0, pixels=None):
    """Film positions (pixel units, [2, n] float32) of the samples of ortho_rays(film_w, film_h, spp, ..., seed, pixels):
    pixel index + the same jitter -- what a reconstruction filter other than the box needs (hf_film_splat)."""
    total = film_w * film_h * spp
    idx = torch.arange(total, dtype=torch.int64, device=device)
    if pixels is not None:
        pixels = pixels.to(device=device, dtype=torch.int64)
        loc = torch.arange(int(pixels.numel()) * spp, dtype=torch.int64, device=device)
        idx = pixels[loc // spp] * spp + loc % spp
    v0, v1 = tea32(torch.full_like(idx, seed), idx)
    jx = (v0 >> 9).to(torch.float64) * (1.0 / (1 << 23))
    jy = (v1 >> 9).to(torch.float64) * (1.0 / (1 << 23))
    pix = idx // spp
    py = pix // film_w
    px = pix - py * film_w
    return torch.stack([(px.to(torch.float64) + jx), (py.to(torch.float64) + jy)]).to(torch.float32)


def secondary_rays(p, n, seed, light_dir=(0.3, 0.2, 0.9)):
    """Incoherent follow-up rays from hit points p [3,n] with normals n [3,n]:
    one cosine-hemisphere bounce and one shadow ray toward a directional light.
    Returns (bounce [7,n], shadow [7,n]); origins are offset by 1e-4 along n."""
    dev = p.device
    m = p.shape[1]
    idx = torch.arange(m, dtype=torch.int64, device=dev)
    v0, v1 = tea32(torch.full_like(idx, seed + 1), idx)
    u1 = (v0 >> 9).to(torch.float32) * (1.0 / (1 << 23))
    u2 = (v1 >> 9).to(torch.float32) * (1.0 / (1 << 23))
    r, phi = torch.sqrt(u1), 2 * math.pi * u2
    lx, ly, lz = r * torch.cos(phi), r * torch.sin(phi), torch.sqrt(torch.clamp(1 - u1, min=0.0))
    a = torch.where(n[0:1].abs() > 0.9, torch.tensor([[0.0], [1.0], [0.0]], device=dev),
                    torch.tensor([[1.0], [0.0], [0.0]], device=dev)).expand(3, m)
    s = torch.linalg.cross(n, a, dim=0)
    s = s / torch.linalg.norm(s, dim=0, keepdim=True)
    t = torch.linalg.cross(n, s, dim=0)
    d = s * lx + t * ly + n * lz
    o = p + 1e-4 * n
    inf = torch.full((1, m), math.inf, device=dev)
    bounce = torch.cat([o, d, inf]).to(torch.float32).contiguous()
    L = torch.tensor(light_dir, dtype=torch.float32, device=dev)
    L = (L / torch.linalg.norm(L)).reshape(3, 1).expand(3, m)
    shadow = torch.cat([o, L, inf]).to(torch.float32).contiguous()
    return bounce, shadow
