"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): prim_index / hit mask bit-exact; t, prim_uv, SI fields and
height gradients within 1e-5 relative.  (t and prim_uv are in fact compared bit-exact
here as well, because both sides use the same explicit operation order.)
"""
import numpy as np
import pytest
import torch

import common

pytestmark = pytest.mark.gpu
REL = 1e-5


def _mk(hf, oracle, h, max_height=0.5, to_world=None, flip=False):
    f_o = oracle.OracleField(h, max_height=max_height, to_world=to_world, flip_normals=flip)
    props = dict(heightfield=torch.from_numpy(h), max_height=max_height, flip_normals=flip)
    if to_world is not None:
        props["to_world"] = torch.from_numpy(np.asarray(to_world))
    f_g = hf.Heightfield(props)
    return f_o, f_g


def _ray(hf, r):
    rt = torch.from_numpy(r).cuda()
    return hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous())


def _check_prelim(hf, f_o, f_g, r, naive=False):
    t, u, v, prim = f_o.ray_intersect_preliminary(r, naive=naive)
    pi = f_g.ray_intersect_preliminary(_ray(hf, r))
    tg, uvg, pg = pi.t.cpu().numpy(), pi.prim_uv.cpu().numpy(), pi.prim_index.cpu().numpy().view(np.uint32)
    assert np.array_equal(np.isfinite(t), np.isfinite(tg)), "hit mask differs"
    assert np.array_equal(prim, pg), f"prim_index differs on {(prim != pg).sum()} rays"
    hit = np.isfinite(t)
    assert np.array_equal(t[hit], tg[hit]) or np.allclose(t[hit], tg[hit], rtol=REL, atol=0)
    assert np.allclose(u, uvg[0], rtol=REL, atol=1e-6) and np.allclose(v, uvg[1], rtol=REL, atol=1e-6)
    st = f_g.ray_test(_ray(hf, r)).cpu().numpy()
    assert np.array_equal(st, hit), "ray_test != ray_intersect_preliminary().is_valid()"
    return hit.mean()


@pytest.mark.parametrize("W,H", [(2, 2), (3, 5), (17, 9), (64, 64), (100, 37)])
@pytest.mark.parametrize("kind", ["rand", "sine", "stairs", "flat"])
def test_prelim_random_and_structured(hf, oracle, W, H, kind):
    rng = np.random.default_rng(W * 1000 + H)
    h = common.heights(kind, W, H, rng)
    f_o, f_g = _mk(hf, oracle, h)
    r = np.concatenate([common.random_rays(4000, rng), common.inside_rays(2000, rng)], 1)
    frac = _check_prelim(hf, f_o, f_g, r, naive=(W * H <= 64 * 64))
    assert frac > 0.2
    xs = np.array([f_o.vertex(0, j)[0] for j in range(W)]); ys = np.array([f_o.vertex(i, 0)[1] for i in range(H)])
    _check_prelim(hf, f_o, f_g, common.structured_rays(xs, ys), naive=(W * H <= 33 * 33))


def test_prelim_transformed(hf, oracle):
    rng = np.random.default_rng(7)
    tw = common.affine(1)
    h = common.heights("sine", 129, 65, rng)
    f_o, f_g = _mk(hf, oracle, h, max_height=0.4, to_world=tw)
    r = common.to_world_rays(np.concatenate([common.random_rays(20000, rng, 0.4), common.inside_rays(5000, rng, 0.4)], 1), tw)
    assert _check_prelim(hf, f_o, f_g, r) > 0.3
    assert np.allclose(f_o.bbox(), f_g.bbox().reshape(-1).numpy(), rtol=1e-6, atol=1e-6)


def test_mips_match_oracle(hf, oracle):
    rng = np.random.default_rng(3)
    for (W, H) in [(2, 2), (5, 3), (64, 64), (257, 100)]:
        h = common.heights("rand", W, H, rng)
        f_o, f_g = _mk(hf, oracle, h, max_height=0.7)
        assert f_g.num_levels() == max(f_o.num_levels(), 1)
        for l in range(1, f_o.num_levels() + 1):
            assert np.array_equal(f_o.mip(l), f_g.mip(l).numpy()), f"mip level {l}"
        if f_o.num_levels() == 0:   # single cell: the one stored level is the cell's range
            z = (h * np.float32(0.7)).astype(np.float32)
            assert np.array_equal(f_g.mip(1).numpy()[0, 0], [z.min(), z.max()])


def test_active_mask_and_miss_records(hf, oracle):
    rng = np.random.default_rng(11)
    h = common.heights("sine", 33, 33, rng)
    f_o, f_g = _mk(hf, oracle, h)
    r = common.random_rays(3000, rng)
    active = rng.uniform(size=3000) < 0.5
    t, u, v, prim = f_o.ray_intersect_preliminary(r, active=active)
    pi = f_g.ray_intersect_preliminary(_ray(hf, r), active=torch.from_numpy(active).cuda())
    assert np.array_equal(t, pi.t.cpu().numpy()) and np.array_equal(prim, pi.prim_index.cpu().numpy().view(np.uint32))
    assert np.all(np.isinf(t[~active]))
    si_o = f_o.compute_surface_interaction(r, t, u, v, prim, oracle.RAY_ALL | oracle.RAY_BOUNDARYTEST, active=active)
    si_g = pi.compute_surface_interaction(_ray(hf, r), hf.RayFlags.All | hf.RayFlags.BoundaryTest,
                                          torch.from_numpy(active).cuda())
    miss = ~np.isfinite(t)
    assert np.all(np.isinf(si_g.t.cpu().numpy()[miss]))
    assert np.all(si_g.p.cpu().numpy()[:, miss] == 0) and np.all(si_g.boundary_test.cpu().numpy()[miss] == 1e8)
    assert np.allclose(si_g.wi.cpu().numpy()[:, miss], -r[3:6, miss])
    assert np.allclose(si_o["wi"], si_g.wi.cpu().numpy(), rtol=REL, atol=1e-6)


def _si_fields(si):
    return {"t": si.t, "p": si.p, "n": si.n, "uv": si.uv, "sh_n": si.sh_frame.n, "dp_du": si.dp_du,
            "dp_dv": si.dp_dv, "boundary_test": si.boundary_test, "sh_s": si.sh_frame.s,
            "sh_t": si.sh_frame.t, "wi": si.wi}


@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("mode", ["default", "follow", "detach"])
def test_surface_interaction_and_adjoint(hf, oracle, mode, flip):
    rng = np.random.default_rng(5)
    tw = common.affine(2)
    h = common.heights("sine", 65, 48, rng)
    f_o, f_g = _mk(hf, oracle, h, max_height=0.45, to_world=tw, flip=flip)
    r = common.to_world_rays(common.random_rays(20000, rng, 0.45), tw)
    flags = oracle.RAY_ALL | oracle.RAY_BOUNDARYTEST | {"default": 0, "follow": oracle.RAY_FOLLOWSHAPE,
                                                        "detach": oracle.RAY_DETACHSHAPE}[mode]
    t, u, v, prim = f_o.ray_intersect_preliminary(r)
    si_o = f_o.compute_surface_interaction(r, t, u, v, prim, flags)
    ray = _ray(hf, r)
    pi = f_g.ray_intersect_preliminary(ray)
    si_g = pi.compute_surface_interaction(ray, flags)
    fused = f_g.ray_intersect(ray, flags)
    for name, val in _si_fields(si_g).items():
        a, b = si_o[name], val.detach().cpu().numpy()
        assert np.allclose(a, b, rtol=REL, atol=2e-6), f"{mode} SI field {name}: max abs diff {np.abs(a - b).max()}"
        c = _si_fields(fused)[name].detach().cpu().numpy()
        assert np.array_equal(b, c, equal_nan=True), f"fused ray_intersect differs from prelim+SI in {name}"
    # adjoint with a random upstream gradient (seed 12345, cf. src/conftest.py:27-30)
    grng = np.random.default_rng(12345)
    g = {nm: grng.normal(size=(c, r.shape[1])).astype(np.float32) for nm, c in oracle.GRAD_FIELDS}
    gh_o, go_o, gd_o = f_o.adjoint(r, t, u, v, prim, g, flags, ray_grads=True)
    gblock = torch.from_numpy(np.concatenate([g[nm].reshape(c, -1) for nm, c in oracle.GRAD_FIELDS])).cuda()
    gh_g, go_g, gd_g = f_g.adjoint(ray, pi, gblock, flags, ray_grads=True)
    scale = max(np.abs(gh_o).max(), 1e-20)
    assert np.abs(gh_o - gh_g.cpu().numpy()).max() <= 2e-5 * scale + 1e-12, \
        f"grad heights: {np.abs(gh_o - gh_g.cpu().numpy()).max()} vs scale {scale}"
    l2 = np.linalg.norm(gh_o - gh_g.cpu().numpy()) / max(np.linalg.norm(gh_o), 1e-20)
    assert l2 <= REL or np.linalg.norm(gh_o) == 0
    assert np.allclose(go_o, go_g.cpu().numpy(), rtol=1e-4, atol=1e-4 * np.abs(go_o).max())
    assert np.allclose(gd_o, gd_g.cpu().numpy(), rtol=1e-4, atol=1e-4 * np.abs(gd_o).max())
    if mode == "detach":
        assert np.all(gh_g.cpu().numpy() == 0)


def test_autograd_backward_matches_adjoint(hf, oracle):
    rng = np.random.default_rng(9)
    h = common.heights("sine", 40, 40, rng)
    f_o, f_g = _mk(hf, oracle, h)
    r = common.random_rays(5000, rng)
    ray = _ray(hf, r)
    f_g.heightfield.requires_grad_(True)
    si = f_g.ray_intersect(ray, hf.RayFlags.All)
    valid = si.is_valid()
    loss = si.t[valid].sum() + (si.n[2][valid] * 0.5).sum() + (si.p[0][valid] * si.uv[1][valid]).sum()
    loss.backward()
    t, u, v, prim = f_o.ray_intersect_preliminary(r)
    si_o = f_o.compute_surface_interaction(r, t, u, v, prim)
    hit = np.isfinite(t)
    g = {"t": hit.astype(np.float32)[None], "n": np.stack([0 * hit, 0 * hit, 0.5 * hit]).astype(np.float32),
         "p": np.stack([si_o["uv"][1] * hit, 0 * hit, 0 * hit]).astype(np.float32),
         "uv": np.stack([0 * hit, si_o["p"][0] * hit]).astype(np.float32)}
    gh_o = f_o.adjoint(r, t, u, v, prim, g)
    gh_g = f_g.heightfield.grad.cpu().numpy()
    assert np.linalg.norm(gh_o - gh_g) <= 1e-5 * np.linalg.norm(gh_o)


def test_error_behaviour(hf):
    with pytest.raises(hf.HfError) as e:
        hf.Heightfield(heightfield=torch.zeros(1, 5))
    assert e.value.code == 1
    f = hf.Heightfield(heightfield=torch.rand(8, 8))
    ray = hf.Ray3f(torch.tensor([[0.0], [0.0], [2.0]]).cuda(), torch.tensor([[0.0], [0.0], [-1.0]]).cuda())
    pi = f.ray_intersect_preliminary(ray)
    with pytest.raises(hf.HfError) as e:   # mesh.cpp:709-711
        pi.compute_surface_interaction(ray, hf.RayFlags.All | hf.RayFlags.DetachShape | hf.RayFlags.FollowShape)
    assert e.value.code == 4 and "DetachShape | FollowShape" in str(e.value)
    with pytest.raises(RuntimeError):
        hf.Heightfield(heightfield=torch.rand(8, 8), bogus=1)
    # empty wavefront is legal
    empty = hf.Ray3f(torch.zeros(3, 0).cuda(), torch.zeros(3, 0).cuda(), torch.zeros(0).cuda())
    assert f.ray_intersect_preliminary(empty).t.numel() == 0
    assert f.ray_test(empty).numel() == 0
    # NaN / inf rays are misses, not hangs
    bad = hf.Ray3f(torch.tensor([[float("nan")], [0.0], [2.0]]).cuda(), torch.tensor([[0.0], [float("inf")], [-1.0]]).cuda())
    assert not f.ray_intersect_preliminary(bad).is_valid().any()


def test_wi_and_shading_frame_carry_gradients(hf, oracle):
    """finalize_surface_interaction is AD-attached in the reference (interaction.h:257-267, 476-499): a loss on
    si.wi.z = <-d, sh_frame.n> must reach the heights and the ray direction.  d(sum wi.z)/dh equals the explicit
    adjoint with upstream dL/dsh_n = -d; d(wi.z)/dd = -sh_n (+ the path through the hit point)."""
    rng = np.random.default_rng(17)
    h = common.heights("sine", 48, 40, rng)
    f_o, f_g = _mk(hf, oracle, h)
    r = common.random_rays(6000, rng)
    f_g.heightfield.requires_grad_(True)
    rt = torch.from_numpy(r).cuda()
    d = rt[3:6].contiguous().requires_grad_(True)
    ray = hf.Ray3f(rt[0:3].contiguous(), d, rt[6].contiguous())
    si = f_g.ray_intersect(ray, hf.RayFlags.All)
    hit = si.is_valid()
    assert float(hit.float().mean()) > 0.2
    # primal: the rebuilt rows agree with the kernel's own
    pi = f_g.ray_intersect_preliminary(ray)
    with torch.no_grad():
        si_k = f_g.compute_surface_interaction(hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous()), pi)
    assert torch.allclose(si.wi[:, hit], si_k.wi[:, hit], rtol=1e-5, atol=1e-6)
    assert torch.allclose(si.sh_frame.s[:, hit], si_k.sh_frame.s[:, hit], rtol=1e-5, atol=1e-6)
    assert torch.allclose(si.sh_frame.t[:, hit], si_k.sh_frame.t[:, hit], rtol=1e-5, atol=1e-6)
    si.wi[2][hit].sum().backward()
    g_auto = f_g.heightfield.grad.clone()
    assert float(g_auto.abs().max()) > 0
    g = torch.zeros((18, r.shape[1]), device="cuda")
    g[9:12] = -rt[3:6] * hit                      # dL/dsh_n = -d on the hit lanes
    g_exp = f_g.adjoint(hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous()), pi, g)
    err = float(torch.linalg.norm((g_auto - g_exp).double()) / torch.linalg.norm(g_exp.double()))
    assert err < 1e-5, err
    # explicit dependence on the direction: -sh_n, plus what flows through sh_n(p(d)) = 0 for flat triangles
    gd = d.grad[:, hit]
    assert torch.allclose(gd, -si.sh_frame.n.detach()[:, hit], rtol=1e-4, atol=1e-5)
    # the tangent frame is differentiable too: a loss on sh_frame.s reaches the heights
    f_g.heightfield.grad = None
    si2 = f_g.ray_intersect(hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous()), hf.RayFlags.All)
    si2.sh_frame.s[2][hit].sum().backward()
    assert float(f_g.heightfield.grad.abs().max()) > 0


def test_wide_path_equals_the_ordinary_path(hf, oracle):
    """Fetches of 256 rays that miss the bound as a whole are answered through 16-byte loads and stores behind a
    conservative clip (DESIGN 4.1 "wide path") -- when every row is 16-byte aligned and there is no mask.  The same
    wavefront through rows that are NOT aligned (every buffer shifted by one float: the launcher then keeps to the
    ordinary path) must give the same bytes in every output row, for all three launches; and both equal the oracle.
    The rays: blocks of 256 beside the field, blocks that graze its bound within a few 1e-5 (where the conservative
    clip has to say "maybe" and the exact one decides), blocks that hit, axis-parallel rays, rays with NaN."""
    rng = np.random.default_rng(5)
    h = common.heights("sine", 97, 130, rng)
    mh = 0.4
    shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=mh)
    f = oracle.OracleField(h, max_height=mh)
    nb = 96
    n = nb * 256 + 77
    kind = rng.integers(0, 4, nb + 1)
    cx = np.where(kind == 0, rng.uniform(1.5, 4.0, nb + 1) * rng.choice([-1, 1], nb + 1),      # beside the field
         np.where(kind == 1, (1.0 + rng.uniform(-3e-5, 3e-5, nb + 1)) * rng.choice([-1, 1], nb + 1),   # grazing its xy bound
                  rng.uniform(-0.9, 0.9, nb + 1)))
    c = np.repeat(np.stack([cx, rng.uniform(-0.9, 0.9, nb + 1)]), 256, axis=1)[:, :n] + rng.uniform(-1e-5, 1e-5, (2, n))
    d = np.repeat(np.array([[1e-3], [2e-3], [-1.0]]), n, 1) + rng.normal(size=(3, n)) * np.where(np.repeat(kind, 256)[:n] == 3, 0.3, 1e-4)
    d /= np.linalg.norm(d, axis=0)
    o = np.concatenate([c, np.full((1, n), 2.0)]) - 0.0 * d
    r = np.concatenate([o, d, np.full((1, n), np.inf)]).astype(np.float32)
    r[3:5, 5 * 256:6 * 256] = 0.0; r[5, 5 * 256:6 * 256] = -1.0          # a block of axis-parallel rays
    r[0, 9 * 256 + 3] = np.nan                                            # a NaN in a block beside the field
    r[6, 11 * 256:12 * 256:7] = -1.0                                      # negative maxt
    import ctypes as C
    from hf_amd import _capi
    lib = _capi.lib()

    def run(shift):
        """every row of every buffer starts `shift` floats into its allocation"""
        P = n + 8
        rays = torch.zeros((7, P), device="cuda"); rays[:, shift:shift + n] = torch.from_numpy(r).cuda()
        pib = torch.full((4, P), 7.0, device="cuda"); sib = torch.full((29, P), 7.0, device="cuda")
        hit = torch.full((P * 4,), 9, dtype=torch.uint8, device="cuda")
        base = lambda b, k: b.data_ptr() + 4 * (P * k + shift)
        rs = _capi.hf_rays_t()
        for k in range(3):
            rs.o[k] = base(rays, k); rs.d[k] = base(rays, 3 + k)
        rs.maxt = base(rays, 6)
        pi = _capi.hf_pi_t(); pi.t, pi.prim_uv[0], pi.prim_uv[1], pi.prim_index = (base(pib, k) for k in range(4))
        si = _capi.hf_si_t()
        rows = [base(sib, k) for k in range(29)]
        si.t, si.boundary_test, si.uv[0], si.uv[1] = rows[0], rows[1], rows[2], rows[3]
        for cc in range(3):
            si.p[cc], si.n[cc], si.sh_n[cc], si.dp_du[cc] = rows[4 + cc], rows[7 + cc], rows[10 + cc], rows[13 + cc]
            si.dp_dv[cc], si.sh_s[cc], si.sh_t[cc], si.wi[cc] = rows[16 + cc], rows[19 + cc], rows[22 + cc], rows[25 + cc]
        flags = int(hf.RayFlags.All | hf.RayFlags.BoundaryTest)
        _capi.check(lib.hf_ray_intersect(shape._h, n, C.byref(rs), flags, None, C.byref(pi), C.byref(si), None))
        fused = (pib[:, shift:shift + n].clone(), sib[:28, shift:shift + n].clone())
        pib.fill_(7.0)
        _capi.check(lib.hf_ray_intersect_preliminary(shape._h, n, C.byref(rs), None, C.byref(pi), None))
        prelim = pib[:, shift:shift + n].clone()
        _capi.check(lib.hf_ray_test(shape._h, n, C.byref(rs), None, hit.data_ptr() + 4 * shift, None))
        torch.cuda.synchronize()
        assert bool((pib[:, :shift] == 7.0).all()) and bool((pib[:, shift + n:] == 7.0).all())   # nothing written beside the rows
        return fused, prelim, hit[4 * shift:4 * shift + n].clone()

    import os
    old = os.environ.get("HF_FORCE_GRAB")
    os.environ["HF_FORCE_GRAB"] = "256"   # (a launch this small fetches 64 rays at a time otherwise: no wide path)
    try:
        (pa, sa), qa, ha = run(0)     # 16-byte aligned rows: wide path
        (pb, sb), qb, hb = run(1)     # rows shifted by one float: ordinary path
    finally:
        if old is None:
            del os.environ["HF_FORCE_GRAB"]
        else:
            os.environ["HF_FORCE_GRAB"] = old
    (pc, sc), qc, hc = run(0)         # and the launch as the library sizes it
    assert torch.equal(pa.view(torch.int32), pc.view(torch.int32)) and torch.equal(sa.view(torch.int32), sc.view(torch.int32))
    assert torch.equal(qa.view(torch.int32), qc.view(torch.int32)) and torch.equal(ha, hc)
    assert torch.equal(pa.view(torch.int32), pb.view(torch.int32)) and torch.equal(sa.view(torch.int32), sb.view(torch.int32))
    assert torch.equal(qa.view(torch.int32), qb.view(torch.int32)) and torch.equal(ha, hb)
    t, u, v, prim = f.ray_intersect_preliminary(r, naive=True, nthreads=16)
    assert np.array_equal(prim, qa[3].view(torch.int32).cpu().numpy().view(np.uint32))
    assert np.array_equal(t.view(np.uint32), qa[0].cpu().numpy().view(np.uint32))
    assert np.array_equal(np.isfinite(t), ha.cpu().numpy() != 0)
    hitf = np.isfinite(t).reshape(-1)[: nb * 256].reshape(nb, 256)
    assert 0.2 < (~hitf.any(1)).mean() < 0.8      # a good part of the 256-ray fetches misses as a whole, a good part does not


def test_forced_fetch_sizes_give_the_same_bytes(hf, oracle):
    """The size of a fetch (rays a wave takes from the work counter at a time; include/hf.h: HF_FORCE_GRAB) decides how
    batches are grouped -- which batch asks for its successor's rays ahead, where the wide path may answer 256 rays at
    once -- and must never show in the results: every launch with fetches of 128 / 192 / 256 / 512 rays forced gives the
    bytes of the launch as the library sizes it (64 here), and those equal the oracle's brute force.  (Written for the ray
    pool of round 4 -- incoherent batches of a fetch walked together, profiles/variants/ray_pool.diff, measured and not
    kept -- and kept as a test of the fetch logic.)  The rays: random and inside-the-bound rays (per-lane walks from the
    root), coherent packets between them (beam sweep), rays beside the field, NaN, negative maxt, a mask, a ragged end."""
    import os
    rng = np.random.default_rng(11)
    h = common.heights("rand", 150, 131, rng)
    mh = 0.6
    tw = common.affine(3)
    shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=mh, to_world=torch.from_numpy(tw))
    f = oracle.OracleField(h, max_height=mh, to_world=tw)
    parts = []
    for b in range(40):
        k = rng.integers(0, 4)
        if k == 0:
            parts.append(common.random_rays(64 * int(rng.integers(1, 6)), rng, mh))
        elif k == 1:
            parts.append(common.inside_rays(64 * int(rng.integers(1, 6)), rng, mh))
        elif k == 2:   # a coherent packet: one pixel's worth of nearly equal rays from above
            c = rng.uniform(-0.8, 0.8, (2, 1)); m = 64 * int(rng.integers(1, 3))
            o = np.concatenate([c + rng.uniform(-1e-3, 1e-3, (2, m)), np.full((1, m), 2.0)])
            d = np.array([[0.05], [0.02], [-1.0]]) + rng.normal(size=(3, m)) * 1e-4
            parts.append(np.concatenate([o, d / np.linalg.norm(d, axis=0), np.full((1, m), np.inf)]))
        else:          # rays beside the field between the others: dead items of a pool
            m = 64
            o = np.stack([rng.uniform(2, 3, m), rng.uniform(-1, 1, m), rng.uniform(0, 1, m)])
            parts.append(np.concatenate([o, np.repeat(np.array([[1.0], [0.0], [0.0]]), m, 1), np.full((1, m), np.inf)]))
    parts.append(common.inside_rays(37, rng, mh))   # ragged end
    r = common.to_world_rays(np.concatenate(parts, 1), tw)
    n = r.shape[1]
    r[1, 100] = np.nan; r[6, 300:340:3] = -1.0
    act = rng.uniform(size=n) < 0.9
    rt = torch.from_numpy(r).cuda()
    ray = hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous())
    actt = torch.from_numpy(act).cuda()

    def run():
        out = []
        for a in (None, actt):
            pi = shape.ray_intersect_preliminary(ray, active=a)
            si = shape.ray_intersect(ray, hf.RayFlags.All, active=a)
            st = shape.ray_test(ray, active=a)
            out += [pi.t, pi.prim_index.to(torch.int64), pi.prim_uv[0], pi.prim_uv[1], si.t, si.p, si.n, si.uv, si.dp_du, st.to(torch.float32)]
        torch.cuda.synchronize()
        return [x.detach().clone() for x in out]

    plain = run()   # fetches of 64 rays: no pool beyond the batch
    old = os.environ.get("HF_FORCE_GRAB")
    try:
        for g in ("128", "192", "256", "512"):
            os.environ["HF_FORCE_GRAB"] = g
            for a, b in zip(plain, run()):
                assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b), g
    finally:
        if old is None:
            del os.environ["HF_FORCE_GRAB"]
        else:
            os.environ["HF_FORCE_GRAB"] = old
    t, u, v, prim = f.ray_intersect_preliminary(r, naive=True, nthreads=16)
    assert np.array_equal(prim, plain[1].cpu().numpy().astype(np.uint32))
    assert np.array_equal(t.view(np.uint32), plain[0].cpu().numpy().view(np.uint32))
    assert np.array_equal(np.isfinite(t), plain[9].cpu().numpy() != 0)
    assert 0.2 < np.isfinite(t).mean() < 0.9


def test_coherence_hint_never_changes_a_result(hf, oracle):
    """``coherent`` of Scene::ray_intersect / ray_test / ray_intersect_preliminary (scene.h:117-146, 188-207, 237-259) is
    a performance hint: with ``coherent=False`` (hf_set_ray_coherence: kernels without the beam sweep, every wave walks
    per lane) and ``coherent=True`` every launch gives the bytes of the automatic mode, on packets, incoherent rays and
    dead rays alike, with a mask and without; the handle's mode is restored afterwards, and an unknown mode is refused."""
    from hf_amd import _capi
    rng = np.random.default_rng(21)
    h = common.heights("sine", 300, 270, rng)      # (top level 9: the beam sweep is in play for the packets)
    mh = 0.5
    f_o, f_g = _mk(hf, oracle, h, max_height=mh)
    parts = [common.random_rays(6400, rng, mh), common.inside_rays(6400, rng, mh)]
    for b in range(60):   # packets: one pixel's worth of nearly equal rays each
        c = rng.uniform(-0.9, 0.9, (2, 1))
        o = np.concatenate([c + rng.uniform(-2e-3, 2e-3, (2, 64)), np.full((1, 64), 2.0)])
        d = np.array([[0.3], [0.2], [-1.0]]) + rng.normal(size=(3, 64)) * 1e-4
        parts.append(np.concatenate([o, d / np.linalg.norm(d, axis=0), np.full((1, 64), np.inf)]))
    r = np.concatenate(parts, 1).astype(np.float32)
    r[2, 77] = np.nan
    n = r.shape[1]
    ray = _ray(hf, r)
    act = torch.from_numpy(rng.uniform(size=n) < 0.9).cuda()

    def run(coherent):
        out = []
        for a in (True, act):
            pi = f_g.ray_intersect_preliminary(ray, active=a, coherent=coherent)
            si = f_g.ray_intersect(ray, hf.RayFlags.All | hf.RayFlags.BoundaryTest, active=a, coherent=coherent)
            st = f_g.ray_test(ray, active=a, coherent=coherent)
            out += [pi.t, pi.prim_index.to(torch.float32), pi.prim_uv, si.t, si.p, si.n, si.uv, si.dp_du, si.boundary_test, st.to(torch.float32)]
        torch.cuda.synchronize()
        return [x.detach().clone() for x in out]

    auto = run(None)
    assert f_g.ray_coherence() == f_g.COHERENCE_AUTO
    for flag in (False, True):
        for a, b in zip(auto, run(flag)):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"coherent={flag}"
        assert f_g.ray_coherence() == f_g.COHERENCE_AUTO      # restored
    f_g.set_ray_coherence(f_g.COHERENCE_INCOHERENT)
    for a, b in zip(auto, run(None)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert _capi.lib().hf_set_ray_coherence(f_g._h, 7) == _capi.HF_EINVAL
    assert f_g.ray_coherence() == f_g.COHERENCE_INCOHERENT
    f_g.set_ray_coherence(f_g.COHERENCE_AUTO)
    t, u, v, prim = f_o.ray_intersect_preliminary(r)
    assert np.array_equal(t.view(np.uint32), auto[0].cpu().numpy().view(np.uint32))
    assert np.array_equal(prim, auto[1].cpu().numpy().astype(np.uint32))
    assert 0.3 < np.isfinite(t).mean() < 0.95
