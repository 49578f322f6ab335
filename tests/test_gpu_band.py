"""HIP traversal == the oracle's BAND brute force at the grid sizes BASELINE.json names.

The band intersector (oracle/hf_oracle.c: trace_band; pinned to the brute force over all cells by
tests/test_oracle_band.py) shares no mip, margin or constant with the hierarchical walks: float64 geometry, every
cell within +/-2 cells of the ray's xy segment, the same fp32 triangle test and tie rule.  This is the reference's
"accelerated == naive on the real scene" (src/render/tests/test_kdtrees.py:52-82) at sizes where the brute force
over every cell cannot run: bit-exact prim_index / t / prim_uv and ray_test on
  * configs[1]: 1024^2 grid, 512^2 @16spp   -- 2^18 rays (16384 whole pixels)
  * configs[3]: 4096^2 grid, 1024^2 @64spp  -- 2^18 rays (4096 whole pixels, so the coherent row sweep is what runs)
  * the bounce rays of configs[3] (one cosine-hemisphere ray per primary hit, incoherent: per-lane walks from the root)
  * mixed rays incl. origins 50 units away and grazing packets on the 4096^2 grid and on white-noise heights
"""
import numpy as np
import pytest
import torch

import common
from test_oracle_band import mixed_rays

pytestmark = pytest.mark.gpu


def _check(hf, shape, f, r_np, rays_t=None):
    rt = torch.from_numpy(r_np).cuda() if rays_t is None else rays_t
    ray = hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous())
    pi = shape.ray_intersect_preliminary(ray)
    t, u, v, prim = f.ray_intersect_preliminary(r_np, band=True, nthreads=16)
    assert np.array_equal(prim, pi.prim_index.cpu().numpy().view(np.uint32)), \
        f"{int((prim != pi.prim_index.cpu().numpy().view(np.uint32)).sum())} prim_index mismatches vs the band brute force"
    assert np.array_equal(t.view(np.uint32), pi.t.cpu().numpy().view(np.uint32))
    assert np.array_equal(u.view(np.uint32), pi.prim_uv[0].cpu().numpy().view(np.uint32))
    assert np.array_equal(v.view(np.uint32), pi.prim_uv[1].cpu().numpy().view(np.uint32))
    assert np.array_equal(shape.ray_test(ray).cpu().numpy(), np.isfinite(t))
    return t


def _pixel_sample(rng, n_rays, spp, n_pix):
    """indices of whole pixels (spp consecutive rays each), sorted: coherent waves stay coherent"""
    pix = np.sort(rng.choice(n_rays // spp, n_pix, replace=False))
    return (pix[:, None] * spp + np.arange(spp)[None, :]).reshape(-1)


@pytest.mark.parametrize("N,film,spp,npix", [(1024, 512, 16, 16384), (4096, 1024, 64, 4096)])
def test_primary_rays_equal_band_brute_force(hf, oracle, N, film, spp, npix):
    dev = torch.device("cuda", 0)
    h = hf.workload.sine_heights(N, N, device=dev)
    shape = hf.Heightfield(heightfield=h, max_height=0.5)
    f = oracle.OracleField(h.cpu().numpy(), max_height=0.5)
    rng = np.random.default_rng(N)
    R = film * film * spp
    idx = _pixel_sample(rng, R, spp, npix)
    # (the wavefront is generated in chunks and the sampled rays gathered from it)
    idx_t = torch.from_numpy(idx).to(dev)
    chunk = 1 << 24
    parts = []
    for c0 in range(0, R, chunk):
        sel = idx_t[(idx_t >= c0) & (idx_t < c0 + chunk)]
        if sel.numel():
            w = hf.workload.ortho_rays(film, film, spp, dev, start=c0, count=min(chunk, R - c0))
            parts.append(w[:, sel - c0])
            del w
    rays = torch.cat(parts, 1).contiguous()
    assert rays.shape[1] == npix * spp >= 1 << 18
    t = _check(hf, shape, f, rays.cpu().numpy(), rays)
    assert 0.15 < np.isfinite(t).mean() < 0.35


def test_bounce_rays_equal_band_brute_force(hf, oracle):
    """the incoherent secondary rays of SURVEY 8d on the 4096^2 grid: one cosine-hemisphere bounce per primary hit"""
    N, film, spp = 4096, 1024, 16
    dev = torch.device("cuda", 0)
    h = hf.workload.sine_heights(N, N, device=dev)
    shape = hf.Heightfield(heightfield=h, max_height=0.5)
    f = oracle.OracleField(h.cpu().numpy(), max_height=0.5)
    rays = hf.workload.ortho_rays(film, film, spp, dev)
    si = shape.ray_intersect(hf.Ray3f(rays[0:3], rays[3:6], rays[6]))
    hit_idx = torch.nonzero(si.is_valid()).squeeze(1)
    rng = np.random.default_rng(3)
    sel = torch.from_numpy(np.sort(rng.choice(hit_idx.numel(), 1 << 17, replace=False))).to(dev)
    hit_idx = hit_idx[sel]
    bounce, shadow = hf.workload.secondary_rays(si.p[:, hit_idx], si.n[:, hit_idx], seed=0)
    t = _check(hf, shape, f, bounce.cpu().numpy(), bounce)
    assert 0.2 < np.isfinite(t).mean() < 0.99
    _check(hf, shape, f, shadow.cpu().numpy(), shadow)


@pytest.mark.parametrize("terrain", ["sine4096", "noise1024"])
def test_mixed_rays_equal_band_brute_force(hf, oracle, terrain):
    rng = np.random.default_rng(11)
    if terrain == "sine4096":
        from test_oracle_band import baseline_heights
        h, mh = baseline_heights(4096), 0.5
    else:
        h, mh = rng.uniform(0, 1, (1024, 1024)).astype(np.float32), 0.05
    shape = hf.Heightfield(heightfield=torch.from_numpy(h), max_height=mh)
    f = oracle.OracleField(h, max_height=mh)
    _check(hf, shape, f, mixed_rays(rng, 1 << 17, mh))


def test_far_origin_needle_regression_on_the_gpu(hf, oracle):
    """the round-3 fuzz find (tests/test_oracle_band.py::test_far_origin_needle_regression) and its regime through the
    HIP kernels: the single ray, and 40 000 rays from 50 units away onto 129^2 white-noise heights against the brute
    force over ALL cells"""
    from test_oracle_band import _fuzz_scene
    h, mh, tw = _fuzz_scene(78, 359)
    f = oracle.OracleField(h, max_height=mh, to_world=tw)
    shape = hf.Heightfield(heightfield=torch.from_numpy(h), max_height=mh, to_world=torch.from_numpy(tw))
    r = np.array([[26.356414794921875, -7.695851802825928, 5.930713653564453,
                   -0.9977114200592041, 0.2779437005519867, -0.2164042592048645, np.inf]], np.float32).T
    rr = np.repeat(r, 64, 1)
    rt = torch.from_numpy(rr).cuda()
    pi = shape.ray_intersect_preliminary(hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous()))
    t, u, v, prim = f.ray_intersect_preliminary(r, naive=True, nthreads=16)
    assert int(prim[0]) == 8133 and np.all(pi.prim_index.cpu().numpy().view(np.uint32) == 8133)
    assert np.all(pi.t.cpu().numpy() == t[0])
    rng = np.random.default_rng(12)
    N, mhn, dist, n = 129, 1.0, 50.0, 40000
    hn = rng.uniform(0, 1, (N, N)).astype(np.float32)
    c = rng.uniform(-1, 1, (2, n)); dirs = rng.normal(size=(3, n))
    dirs[2] = -np.abs(dirs[2]) * rng.uniform(0.05, 1.0, n); dirs /= np.linalg.norm(dirs, axis=0)
    o = np.concatenate([c, np.full((1, n), mhn * 0.5)]) - dirs * dist
    rays = np.concatenate([o, dirs, np.full((1, n), np.inf)]).astype(np.float32)
    fn = oracle.OracleField(hn, max_height=mhn)
    sn = hf.Heightfield(heightfield=torch.from_numpy(hn), max_height=mhn)
    rt = torch.from_numpy(rays).cuda()
    pi = sn.ray_intersect_preliminary(hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous()))
    t, u, v, prim = fn.ray_intersect_preliminary(rays, naive=True, nthreads=16)
    assert np.array_equal(prim, pi.prim_index.cpu().numpy().view(np.uint32))
    assert np.array_equal(t.view(np.uint32), pi.t.cpu().numpy().view(np.uint32))


def test_noise_hit_above_the_bound_regression_on_the_gpu(hf, oracle):
    """the second round-3 fuzz find (tests/test_oracle_band.py::test_noise_hit_above_the_bound_regression) through the HIP
    kernels: closest hit, any hit and the fused record of the single ray (a wave of copies and a lone ray)"""
    from test_oracle_band import _fuzz_scene
    h, mh, tw = _fuzz_scene(301, 166)
    f = oracle.OracleField(h, max_height=mh, to_world=tw)
    shape = hf.Heightfield(heightfield=torch.from_numpy(h), max_height=mh, to_world=torch.from_numpy(tw))
    r = np.array([[24.74660873413086, -6.202888011932373, 2.355172634124756,
                   -0.8037796020507812, 0.23658400774002075, -0.07390778511762619, np.inf]], np.float32).T
    t, u, v, prim = f.ray_intersect_preliminary(r, band=True, nthreads=16)
    assert int(prim[0]) == 6444120
    for copies in (64, 1):
        rt = torch.from_numpy(np.repeat(r, copies, 1)).cuda()
        ray = hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous())
        pi = shape.ray_intersect_preliminary(ray)
        assert np.all(pi.prim_index.cpu().numpy().view(np.uint32) == 6444120) and np.all(pi.t.cpu().numpy() == t[0])
        assert bool(shape.ray_test(ray).all())
        si = shape.ray_intersect(ray)
        assert np.all(si.t.cpu().numpy() == t[0]) and np.all(si.prim_index.cpu().numpy().view(np.uint32) == 6444120)


def test_walk_needle_term_regression_on_the_gpu(hf, oracle):
    """the six rays of tests/test_oracle_band.py::test_walk_needle_term_regression (white noise, N = 4096, origins 8
    units away: the rays on which the oracle's walk -- before its round-4 needle term -- or the band brute force lose
    to the brute force over all 16.7 M cells) through the HIP kernels: closest hit == the full brute force's, as a
    wave of copies and as lone rays in one launch"""
    from test_oracle_band import _far_origin_sweep_case
    h, mh, r_all = _far_origin_sweep_case(4)
    idx = np.array([95386, 138178, 279488, 286422, 288405, 379107])
    r = np.ascontiguousarray(r_all[:, idx])
    f = oracle.OracleField(h, max_height=mh)
    t, u, v, prim = f.ray_intersect_preliminary(r, naive=True, nthreads=16)
    assert prim.tolist() == [18618549, 5812674, 20465175, 14063183, 11430767, 13598460]
    shape = hf.Heightfield(heightfield=torch.from_numpy(h), max_height=mh)
    for copies in (64, 1):
        rt = torch.from_numpy(np.repeat(r, copies, 1)).cuda()
        pi = shape.ray_intersect_preliminary(hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous()))
        got = pi.prim_index.cpu().numpy().view(np.uint32).reshape(6, copies)
        assert np.array_equal(got, np.repeat(prim, copies).reshape(6, copies)), \
            f"rays {np.nonzero((got != prim[:, None]).any(1))[0].tolist()} of the six differ from the full brute force: {got[:, 0].tolist()} vs {prim.tolist()}"
        assert np.array_equal(pi.t.cpu().numpy().view(np.uint32), np.repeat(t.view(np.uint32), copies))
