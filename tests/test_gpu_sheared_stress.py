"""Stress of the sheared bounds: terrains on which a plane fit is excellent (steep
smooth slopes), useless (white noise, cliffs) or degenerate (constant, huge / tiny max_height),
with rays that graze the surface, run along grid lines, or come as coherent 64-ray packets (the
shared upper-level walk).  Every ray is compared with the oracle: prim_index, t and the any-hit
mask bit for bit -- a bound that is not conservative shows up as a missing or a different hit.
"""
import numpy as np
import pytest
import torch

import common

pytestmark = pytest.mark.gpu


def _terrain(kind, W, H, rng):
    u = np.arange(W) / (W - 1.0); v = np.arange(H)[:, None] / (H - 1.0)
    if kind == "steep":      # slopes of several cells per cell, smooth: the case the planes are made for
        return (0.5 + 0.45 * np.sin(2 * np.pi * 11 * u) * np.cos(2 * np.pi * 9 * v)).astype(np.float32)
    if kind == "noise":      # no plane fits
        return rng.uniform(0, 1, (H, W)).astype(np.float32)
    if kind == "cliffs":     # piecewise constant with vertical walls
        return ((np.floor(u * 13) + np.floor(v * 7)) % 3 / 2.0 + 0 * u).astype(np.float32)
    if kind == "ramp":       # one exact plane: residual ranges are ~0, everything hangs on the margins
        return (0.1 + 0.8 * (0.7 * u + 0.3 * v)).astype(np.float32)
    if kind == "ridge":      # smooth slope + sharp ridge lines on node boundaries
        return (0.2 + 0.6 * np.abs(((u * 8) % 1.0) - 0.5) + 0.2 * v).astype(np.float32)
    raise ValueError(kind)


def _grazing_rays(f_o, n, rng, W, H):
    """rays that start a hair above a surface vertex and leave nearly parallel to the local slope"""
    i = rng.integers(0, H, n); j = rng.integers(0, W, n)
    p = np.stack([f_o.vertex(int(a), int(b)) for a, b in zip(i, j)], 1).astype(np.float64)   # [3, n]
    ang = rng.uniform(0, 2 * np.pi, n)
    d = np.stack([np.cos(ang), np.sin(ang), rng.normal(0, 0.05, n)])
    back = rng.uniform(0.05, 0.6, n)
    o = p - d * back + np.array([[0.0], [0.0], [1.0]]) * rng.uniform(-2e-3, 2e-3, n)
    return np.concatenate([o, d, np.full((1, n), np.inf)]).astype(np.float32)


def _packet_rays(n_pix, rng, zmax):
    """64 nearly identical rays per 'pixel' (coherent waves), oblique, like a sensor's samples"""
    c = rng.uniform(-0.9, 0.9, (2, n_pix))
    dirs = rng.normal(size=(3, n_pix)); dirs[2] = -np.abs(dirs[2]) * 0.7 - 0.2
    dirs /= np.linalg.norm(dirs, axis=0)
    o = np.concatenate([c, np.full((1, n_pix), zmax * 0.5)]) - dirs * 2.5
    o = np.repeat(o, 64, 1) + rng.uniform(-2e-3, 2e-3, (3, n_pix * 64))
    d = np.repeat(dirs, 64, 1) * (1 + rng.uniform(-1e-3, 1e-3, (1, n_pix * 64)))
    maxt = np.full((1, n_pix * 64), np.inf)
    # partially dead packets: lanes with a short / negative maxt, a non-finite origin or direction
    k = rng.uniform(size=n_pix * 64)
    maxt[0, k < 0.10] = rng.uniform(0.5, 3.5, int((k < 0.10).sum()))
    maxt[0, (k >= 0.10) & (k < 0.13)] = -1.0
    o[0, (k >= 0.13) & (k < 0.15)] = np.nan
    d[2, (k >= 0.15) & (k < 0.17)] = np.inf
    return np.concatenate([o, d, maxt]).astype(np.float32)


def _compare(hf, f_o, f_g, r):
    rt = torch.from_numpy(r).cuda()
    ray = hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous())
    pi = f_g.ray_intersect_preliminary(ray)
    t, u, v, prim = f_o.ray_intersect_preliminary(r, nthreads=16)
    pg, tg = pi.prim_index.cpu().numpy().view(np.uint32), pi.t.cpu().numpy()
    bad = np.nonzero((prim != pg) | (t != tg))[0]
    assert bad.size == 0, f"{bad.size} rays differ, first {bad[:5]}: oracle {prim[bad[:5]]} {t[bad[:5]]} gpu {pg[bad[:5]]} {tg[bad[:5]]}"
    assert np.array_equal(f_g.ray_test(ray).cpu().numpy(), np.isfinite(t))
    return np.isfinite(t).mean()


@pytest.mark.parametrize("kind", ["steep", "noise", "cliffs", "ramp", "ridge"])
@pytest.mark.parametrize("W,H,max_height", [(257, 193, 0.5), (130, 300, 6.0), (513, 513, 1e-3)])
def test_sheared_bounds_are_conservative(hf, oracle, kind, W, H, max_height):
    rng = np.random.default_rng(sum(map(ord, kind)) * 1000 + W)
    h = _terrain(kind, W, H, rng)
    tw = common.affine(W) if W == 130 else None
    f_o = oracle.OracleField(h, max_height=max_height, to_world=tw)
    props = dict(heightfield=torch.from_numpy(h), max_height=max_height)
    if tw is not None:
        props["to_world"] = torch.from_numpy(tw)
    f_g = hf.Heightfield(props)
    zmax = max_height
    xs = np.array([f_o.vertex(0, j)[0] for j in range(0, W, max(1, W // 24))])
    ys = np.array([f_o.vertex(i, 0)[1] for i in range(0, H, max(1, H // 24))])
    # packets first: 64-aligned, so that each one is exactly one wave
    r_obj = [_packet_rays(600, rng, zmax), common.random_rays(60000, rng, zmax), common.inside_rays(30000, rng, zmax)]
    if tw is None:
        r_obj += [_grazing_rays(f_o, 20000, rng, W, H), common.structured_rays(xs, ys, -0.2 * zmax, 1.2 * zmax)]
    r = common.to_world_rays(np.concatenate(r_obj, 1), tw)
    frac = _compare(hf, f_o, f_g, r)
    assert 0.05 < frac <= 1.0
