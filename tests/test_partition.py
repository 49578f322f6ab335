"""Row (e): the image-tile partition of one wavefront over ranks (BASELINE configs[3]).
CPU: shard lists are disjoint, cover the film, are balanced, and generate exactly the rays of the full
wavefront.  GPU (one device, ranks run one after the other as "virtual ranks"): the shards' results
concatenate to the full wavefront's bit for bit and the shard gradients sum to the full gradient."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("film", [(64, 64), (100, 37), (32, 96)])
def test_partition_is_disjoint_cover(world, film):
    import hf_amd
    W, H = film
    parts = hf_amd.workload.partition_tiles(W, H, world, tile=32)
    assert len(parts) == world
    allpix = torch.cat(parts)
    assert allpix.numel() == W * H and torch.equal(torch.sort(allpix).values, torch.arange(W * H))
    # interleaved blocks: no rank holds more than one block above its fair share
    nblocks = ((W + 31) // 32) * ((H + 31) // 32)
    assert max(p.numel() for p in parts) <= ((nblocks + world - 1) // world) * 32 * 32
    # pixels of a block are consecutive in the list (waves stay spatially coherent)
    for p in parts:
        if p.numel():
            b = (p // W // 32) * ((W + 31) // 32) + (p % W) // 32
            assert (torch.diff(b) >= 0).all()


def test_sharded_rays_equal_full_wavefront():
    import hf_amd
    W = H = 64; spp = 4
    full = hf_amd.workload.ortho_rays(W, H, spp, "cpu")
    for world in (2, 3):
        parts = hf_amd.workload.partition_tiles(W, H, world)
        for r in range(world):
            sh = hf_amd.workload.ortho_rays(W, H, spp, "cpu", pixels=parts[r])
            idx = (parts[r][:, None] * spp + torch.arange(spp)[None, :]).reshape(-1)
            assert torch.equal(sh, full[:, idx])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_virtual_ranks_reproduce_single_gpu(hf, world):
    """Every rank's tile shard through the HIP path, one after the other on one device: concatenated pi/si equal
    the single-launch results bit for bit, and the sum of the shard gradient textures equals the full gradient."""
    dev = torch.device("cuda", 0)
    N, W, spp = 512, 128, 16
    shape = hf.Heightfield(heightfield=hf.workload.sine_heights(N, N, device=dev), max_height=0.5)
    flags = hf.RayFlags.All

    def run(rays):
        ray = hf.Ray3f(rays[0:3].contiguous(), rays[3:6].contiguous(), rays[6].contiguous())
        pi = shape.ray_intersect_preliminary(ray)
        si = shape.compute_surface_interaction(ray, pi, flags)
        g = torch.zeros((18, rays.shape[1]), device=dev)
        hit = pi.is_valid()
        g[0] = hit.float(); g[1:4] = si.n * hit           # dL/dt = 1, dL/dp = n
        gh = shape.adjoint(ray, pi, g, flags)
        return pi, si, gh

    full = hf.workload.ortho_rays(W, W, spp, dev)
    pi_f, si_f, gh_f = run(full)
    assert float(pi_f.is_valid().float().mean()) > 0.1
    parts = hf.workload.partition_tiles(W, W, world)
    gh_sum = torch.zeros_like(gh_f)
    for r in range(world):
        rays = hf.workload.ortho_rays(W, W, spp, dev, pixels=parts[r])
        pi, si, gh = run(rays)
        idx = (parts[r].to(dev)[:, None] * spp + torch.arange(spp, device=dev)[None, :]).reshape(-1)
        assert torch.equal(pi.t, pi_f.t[idx]) and torch.equal(pi.prim_index, pi_f.prim_index[idx])
        assert torch.equal(pi.prim_uv, pi_f.prim_uv[:, idx])
        assert torch.equal(si.p, si_f.p[:, idx]) and torch.equal(si.n, si_f.n[:, idx])
        gh_sum += gh
    err = float(torch.linalg.norm((gh_sum - gh_f).double()) / torch.linalg.norm(gh_f.double()))
    assert err < 1e-5, err     # float atomics: order of the additions differs, nothing else


@pytest.mark.gpu
def test_adjoint_row_band_is_the_rows_that_received_gradient(hf):
    """hf_adjoint_rows: the reported band [lo, hi) is exactly the span of the texture rows of the hit triangles'
    vertices (cell row cy -> vertex rows cy, cy + 1), accumulated over several launches -- what a multi-GPU host may
    restrict its all-reduce to (BASELINE configs[3]: one all-reduce of the gradient texture per step)."""
    import torch
    dev = torch.device("cuda", 0)
    N, film, spp = 512, 128, 4
    h = hf.workload.sine_heights(N, N, device=dev)
    shape = hf.Heightfield(heightfield=h, max_height=0.5)
    rays = hf.workload.ortho_rays(film, film, spp, dev)
    ray = hf.Ray3f(rays[0:3], rays[3:6], rays[6])
    pi = shape.ray_intersect_preliminary(ray)
    cy = (pi.prim_index.to(torch.int64) >> 1) // (N - 1)
    keep = pi.is_valid() & (cy >= 100) & (cy < 237)               # only the hits of a band of cell rows take part
    assert int(keep.sum()) > 1000
    g = torch.zeros((18, len(ray)), device=dev); g[0] = 1.0
    band = shape.new_row_band()
    grad = torch.zeros((N, N), device=dev)
    half = len(ray) // 2
    for sl in (slice(0, half), slice(half, len(ray))):            # two launches, one band
        r2 = hf.Ray3f(rays[0:3, sl].contiguous(), rays[3:6, sl].contiguous(), rays[6, sl].contiguous())
        p2 = shape.ray_intersect_preliminary(r2)
        shape.adjoint(r2, p2, g[:, sl].contiguous(), active=keep[sl].contiguous(), grad_heightfield=grad, row_band=band)
    lo, hi = (int(x) for x in band.cpu())
    assert lo == int(cy[keep].min()) and hi == int(cy[keep].max()) + 2
    assert float(grad[:lo].abs().max()) == 0.0 and float(grad[hi:].abs().max()) == 0.0 and float(grad[lo:hi].abs().max()) > 0
    untouched = shape.new_row_band()                              # a launch without hits leaves the band empty
    shape.adjoint(ray, pi, g, active=torch.zeros_like(keep), row_band=untouched)
    assert [int(x) for x in untouched.cpu()] == [N, 0]
