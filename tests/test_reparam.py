"""Warped-area reparameterisation of rays for the heightfield (reparam.py:10-123, 151-333).
CPU: the oracle restatement -- forward and backward mode are transposes of each other; the known answer
the reference tests state for a moving shape (src/render/tests/test_reparameterization.py:29-98: the
derivative of the direction equals the motion of the attached hit point) carried over to a rising
height field.  GPU: hf_reparam_* + hf_ray_intersect + hf_adjoint (host mirror hf_amd.reparameterize_ray)
against the oracle."""
import numpy as np
import pytest


def _scene(oracle, W=33, H=29, seed=0):
    rng = np.random.default_rng(seed)
    u = np.arange(W) / (W - 1.0); v = np.arange(H)[:, None] / (H - 1.0)
    h = (0.5 + 0.3 * np.sin(2 * np.pi * 1.5 * u) * np.cos(2 * np.pi * 1.2 * v) + 0.03 * rng.uniform(-1, 1, (H, W))).astype(np.float32)
    return h, oracle.OracleField(h, max_height=0.5)


def _rays(n, rng):
    tgt = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(-0.8, 0.8, n), np.full(n, 0.25)])
    o = tgt + np.stack([rng.uniform(-0.6, 0.6, n), rng.uniform(-0.6, 0.6, n), rng.uniform(1.0, 2.0, n)])
    d = tgt - o; d /= np.linalg.norm(d, axis=0)
    return o.astype(np.float32), d.astype(np.float32)


@pytest.mark.parametrize("kappa,antithetic", [(30.0, False), (2000.0, True)])
def test_oracle_backward_is_the_transpose_of_forward(oracle, kappa, antithetic):
    h, f = _scene(oracle)
    rng = np.random.default_rng(1)
    o, d = _rays(40, rng)
    dh = rng.normal(size=h.shape)
    gd = rng.normal(size=(3, 40)); gdiv = rng.normal(size=40)
    Vt, div = oracle.reparam_forward(f, o, d, dh, num_rays=6, kappa=kappa, antithetic=antithetic, seed=3)
    gh = oracle.reparam_backward(f, o, d, gd, gdiv, num_rays=6, kappa=kappa, antithetic=antithetic, seed=3)
    dd = d.astype(np.float64)
    PV = Vt - dd * (dd * Vt).sum(0)                      # the backward differentiates normalize(d + V_theta)
    lhs = (gd * PV).sum() + (gdiv * div).sum()
    rhs = (gh * dh).sum()
    assert np.isclose(lhs, rhs, rtol=2e-4, atol=1e-7), (lhs, rhs)
    assert abs(rhs) > 1e-6


def test_oracle_direction_follows_a_rising_surface(oracle):
    """test_reparameterization.py:29-98 with 'the shape translates' -> 'all heights rise': for concentrated
    auxiliary rays the derivative of the direction is the motion of the attached hit point, projected."""
    h = np.full((17, 17), 0.5, np.float32)                # flat, hit far from the border
    f = oracle.OracleField(h, max_height=0.5)
    o = np.array([[0.1], [-0.05], [2.0]], np.float32); d = np.array([[0.2], [0.1], [-1.0]], np.float32); d /= np.linalg.norm(d)
    Vt, div = oracle.reparam_forward(f, o, d, np.ones_like(h, dtype=np.float64), num_rays=32, kappa=1e6, exponent=3.0)
    r = np.concatenate([o, d, [[np.inf]]]).astype(np.float32)
    t, u, v, prim = f.ray_intersect_preliminary(r)
    p = f.compute_surface_interaction(r, t, u, v, prim, oracle.RAY_ALL)["p"].astype(np.float64)
    eps = 1e-4
    new_d = (p + np.array([[0], [0], [0.5 * eps]]) - o); new_d /= np.linalg.norm(new_d)
    fd = (new_d - d.astype(np.float64)) / eps
    assert np.allclose(Vt, fd, atol=1e-2 * np.abs(fd).max())


@pytest.mark.gpu
@pytest.mark.parametrize("kappa,antithetic,num_rays", [(30.0, False, 5), (2000.0, True, 8), (1e5, False, 4)])
def test_gpu_reparameterize_ray_matches_oracle(hf, oracle, kappa, antithetic, num_rays):
    import torch
    h, f = _scene(oracle, seed=2)
    rng = np.random.default_rng(4)
    n = 3000
    o, d = _rays(n, rng)
    active = (rng.uniform(size=n) < 0.9)
    shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=0.5)
    shape.heightfield.requires_grad_(True)
    ray = hf.Ray3f(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
    dirn, det = hf.reparameterize_ray(shape, ray, num_rays=num_rays, kappa=kappa, exponent=3.0, antithetic=antithetic,
                                      seed=7, active=torch.from_numpy(active).cuda())
    assert torch.equal(dirn, ray.d) and bool((det == 1).all())                  # identity in primal mode
    gd = rng.normal(size=(3, n)).astype(np.float32); gdiv = rng.normal(size=n).astype(np.float32)
    ((dirn * torch.from_numpy(gd).cuda()).sum() + (det * torch.from_numpy(gdiv).cuda()).sum()).backward()
    got = shape.heightfield.grad.cpu().numpy().astype(np.float64)
    ref = oracle.reparam_backward(f, o, d, gd, gdiv, num_rays=num_rays, kappa=kappa, exponent=3.0, antithetic=antithetic,
                                  seed=7, active=active)
    assert np.linalg.norm(ref) > 0
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert rel <= 3e-5, rel   # measured 1e-6 .. 9e-6: the weights are (1/(D-1+B))^3, float32 like the reference's Float


@pytest.mark.gpu
@pytest.mark.parametrize("kappa,antithetic,num_rays", [(30.0, False, 5), (2000.0, True, 8), (1e5, False, 4),
                                                       (1e4, False, 1), (500.0, True, 32)])
def test_gpu_fused_loops_equal_the_per_sample_kernels(hf, kappa, antithetic, num_rays):
    """hf_reparam_backward (one kernel for all samples; the default when only the heights are differentiated) against
    hf_reparam_aux_rays / hf_reparam_weights / hf_adjoint per sample: the same arithmetic, so the only difference is
    the order of the float atomics of the scatter."""
    import torch
    from hf_amd import shape as shape_mod
    rng = np.random.default_rng(11)
    W, H = 65, 47
    u = np.arange(W) / (W - 1.0); v = np.arange(H)[:, None] / (H - 1.0)
    h = (0.5 + 0.3 * np.sin(2 * np.pi * 1.5 * u) * np.cos(2 * np.pi * 1.2 * v) + 0.03 * rng.uniform(-1, 1, (H, W))).astype(np.float32)
    n = 20000
    o, d = _rays(n, rng)
    active = torch.from_numpy(rng.uniform(size=n) < 0.9).cuda()
    gd = torch.from_numpy(rng.normal(size=(3, n)).astype(np.float32)).cuda()
    gdiv = torch.from_numpy(rng.normal(size=n).astype(np.float32)).cuda()
    grads = []
    for fused in (True, False):
        shape_mod.REPARAM_FUSED = fused
        try:
            shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=0.5)
            shape.heightfield.requires_grad_(True)
            ray = hf.Ray3f(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
            dirn, det = hf.reparameterize_ray(shape, ray, num_rays=num_rays, kappa=kappa, exponent=3.0,
                                              antithetic=antithetic, seed=7, active=active)
            ((dirn * gd).sum() + (det * gdiv).sum()).backward()
            grads.append(shape.heightfield.grad.double().cpu().numpy())
        finally:
            shape_mod.REPARAM_FUSED = True
    assert np.linalg.norm(grads[1]) > 0
    rel = np.linalg.norm(grads[0] - grads[1]) / np.linalg.norm(grads[1])
    assert rel <= 2e-6, rel


@pytest.mark.gpu
def test_gpu_reparam_trace_equals_aux_rays_plus_ray_intersect(hf):
    """hf_reparam_trace (auxiliary ray generated inside the traversal kernel) is bitwise hf_reparam_aux_rays followed by
    hf_ray_intersect(All | FollowShape | BoundaryTest), inactive lanes included."""
    import ctypes as C
    import torch
    from hf_amd import _capi
    L = _capi.lib()
    rng = np.random.default_rng(21)
    W, H = 130, 97
    u = np.arange(W) / (W - 1.0); v = np.arange(H)[:, None] / (H - 1.0)
    h = (0.5 + 0.3 * np.sin(2 * np.pi * 1.5 * u) * np.cos(2 * np.pi * 1.2 * v) + 0.03 * rng.uniform(-1, 1, (H, W))).astype(np.float32)
    shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=0.5)
    n = 30011
    o, d = _rays(n, rng)
    ot, dt = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    act = torch.from_numpy((rng.uniform(size=n) < 0.85).astype(np.uint8)).cuda()
    p3 = lambda x: (C.c_void_p * 3)(x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr())
    flags = int(hf.RayFlags.All | hf.RayFlags.FollowShape | hf.RayFlags.BoundaryTest)

    def out():
        buf = torch.full((30, n), float("nan"), device="cuda")
        rows = [buf[k].data_ptr() for k in range(30)]
        si = _capi.hf_si_t()
        si.t = rows[0]; si.boundary_test = rows[1]
        for c in range(3):
            si.p[c], si.n[c], si.sh_n[c], si.dp_du[c] = rows[2 + c], rows[5 + c], rows[8 + c], rows[11 + c]
            si.dp_dv[c], si.sh_s[c], si.sh_t[c], si.wi[c] = rows[14 + c], rows[17 + c], rows[20 + c], rows[23 + c]
        si.uv[0], si.uv[1] = rows[26], rows[27]
        pib = torch.full((4, n), float("nan"), device="cuda")
        pi = _capi.hf_pi_t()
        pi.t, pi.prim_uv[0], pi.prim_uv[1], pi.prim_index = (pib[k].data_ptr() for k in range(4))
        return buf, si, pib, pi

    for k, kappa, anti in [(0, 30.0, False), (3, 500.0, True), (2, 1e5, True)]:
        buf1, si1, pib1, pi1 = out()
        _capi.check(L.hf_reparam_trace(shape._h, n, C.byref(p3(ot)), C.byref(p3(dt)), act.data_ptr(), k, kappa, int(anti),
                                       5, None, C.byref(pi1), C.byref(si1), None))
        ad = torch.empty_like(dt); mt = torch.empty(n, device="cuda")
        _capi.check(L.hf_reparam_aux_rays(n, C.byref(p3(ot)), C.byref(p3(dt)), act.data_ptr(), k, kappa, int(anti), 5,
                                          None, C.byref(p3(ad)), mt.data_ptr(), None))
        buf2, si2, pib2, pi2 = out()
        rs = shape._rays_struct(ot, ad, mt)
        _capi.check(L.hf_ray_intersect(shape._h, n, C.byref(rs), flags, None, C.byref(pi2), C.byref(si2), None))
        torch.cuda.synchronize()
        assert torch.equal(pib1.view(torch.int32), pib2.view(torch.int32))
        assert torch.equal(buf1[:28].view(torch.int32), buf2[:28].view(torch.int32))
        assert bool(torch.isfinite(pib1[0]).any()) and not bool(torch.isfinite(pib1[0][act == 0]).any())


@pytest.mark.gpu
@pytest.mark.parametrize("kappa,anti,wi", [(1e5, True, False), (500.0, False, False), (30.0, True, False), (1e5, False, True)])
def test_gpu_reparam_trace_all_equals_the_single_sample_launches(hf, kappa, anti, wi):
    """hf_reparam_trace_all (every sample of a ray in ONE launch, batches whose cones miss the bound answered without a
    sample being drawn) is bitwise num_rays x hf_reparam_trace -- on rays of which most pass far from the field (culled
    batches), some graze its border (the cone test says maybe, the samples decide) and the rest hit; with si.wi asked
    for (no culling: a miss record carries minus the sample's direction) and without."""
    import ctypes as C
    import torch
    from hf_amd import _capi
    L = _capi.lib()
    rng = np.random.default_rng(33)
    W, H = 130, 97
    u = np.arange(W) / (W - 1.0); v = np.arange(H)[:, None] / (H - 1.0)
    h = (0.5 + 0.3 * np.sin(2 * np.pi * 1.5 * u) * np.cos(2 * np.pi * 1.2 * v) + 0.03 * rng.uniform(-1, 1, (H, W))).astype(np.float32)
    shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=0.5)
    K, n = 5, 64 * 700 + 13
    # a wide orthographic bundle: coherent blocks of 64 rays, two thirds of them beside the field
    tgt = np.repeat(rng.uniform(-3.0, 3.0, (2, n // 64 + 1)), 64, axis=1)[:, :n] + rng.uniform(-0.02, 0.02, (2, n))
    dirn = np.array([0.35, 0.2, -0.9]); dirn /= np.linalg.norm(dirn)
    o = (np.concatenate([tgt, np.full((1, n), 0.2)]) - 2.5 * dirn[:, None]).astype(np.float32)
    d = np.repeat(dirn[:, None], n, 1).astype(np.float32)
    ot, dt = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    act = torch.from_numpy((rng.uniform(size=n) < 0.9).astype(np.uint8)).cuda()
    rid = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
    p3 = lambda x: (C.c_void_p * 3)(x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr())

    def out():
        buf = torch.full((K, 12, n), float("nan"), device="cuda")   # per sample: pi (4 rows), si.t, si.p, boundary_test, wi
        def structs(k):
            r = [buf[k, j].data_ptr() for j in range(12)]
            pi = _capi.hf_pi_t(); pi.t, pi.prim_uv[0], pi.prim_uv[1], pi.prim_index = r[0:4]
            si = _capi.hf_si_t(); si.t = r[4]; si.boundary_test = r[8]
            for c in range(3):
                si.p[c] = r[5 + c]
                if wi:
                    si.wi[c] = r[9 + c]
            return pi, si
        return buf, structs

    one, s1 = out()
    pi, si = s1(0)
    _capi.check(L.hf_reparam_trace_all(shape._h, n, C.byref(p3(ot)), C.byref(p3(dt)), act.data_ptr(), K, kappa, int(anti), 9,
                                       rid.data_ptr(), C.byref(pi), C.byref(si), 12 * n, None))
    ref, s2 = out()
    for k in range(K):
        pi, si = s2(k)
        _capi.check(L.hf_reparam_trace(shape._h, n, C.byref(p3(ot)), C.byref(p3(dt)), act.data_ptr(), k, kappa, int(anti), 9,
                                       rid.data_ptr(), C.byref(pi), C.byref(si), None))
    torch.cuda.synchronize()
    rows = 12 if wi else 9
    assert torch.equal(one[:, :rows].view(torch.int32), ref[:, :rows].view(torch.int32))
    hit = torch.isfinite(ref[:, 0])
    assert bool(hit.any()) and not bool(hit.all())
    if kappa >= 500.0:   # (a narrow lobe: the samples stay near their ray)
        assert 0.05 < float(hit.float().mean()) < 0.6, float(hit.float().mean())   # most rays pass beside the field, a good part hits
        blocks = hit[:, : (n // 64) * 64].reshape(K, -1, 64).any(2).any(0)
        assert float(blocks.float().mean()) < 0.5      # ... and more than half of the 64-ray batches have no hit at all


@pytest.mark.gpu
def test_gpu_aux_rays_match_oracle(hf, oracle):
    import ctypes as C
    import torch
    from hf_amd import _capi
    rng = np.random.default_rng(9)
    n = 5000
    o, d = _rays(n, rng)
    ot, dt = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    ad = torch.empty_like(dt); mt = torch.empty(n, device="cuda")
    p3 = lambda x: (C.c_void_p * 3)(x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr())
    for k, kappa, anti in [(0, 30.0, False), (3, 500.0, True), (2, 1e5, True)]:
        _capi.check(_capi.lib().hf_reparam_aux_rays(n, C.byref(p3(ot)), C.byref(p3(dt)), None, k, kappa, int(anti), 11,
                                                    None, C.byref(p3(ad)), mt.data_ptr(), None))
        ref = oracle.reparam_aux_rays(o, d, k, kappa, anti, 11)
        got = ad.cpu().numpy()
        assert np.allclose(got, ref[3:6], atol=2e-6), np.abs(got - ref[3:6]).max()
        assert np.allclose(np.linalg.norm(got, axis=0), 1.0, atol=1e-5) and bool(torch.isinf(mt).all())
    # explicit ray ids: the samples follow the id, not the position in the batch
    ids = rng.permutation(n).astype(np.uint32) + 123456
    idt = torch.from_numpy(ids.view(np.int32)).cuda()
    _capi.check(_capi.lib().hf_reparam_aux_rays(n, C.byref(p3(ot)), C.byref(p3(dt)), None, 1, 200.0, 0, 11,
                                                idt.data_ptr(), C.byref(p3(ad)), mt.data_ptr(), None))
    with oracle.with_ray_ids(ids):
        ref = oracle.reparam_aux_rays(o, d, 1, 200.0, False, 11)
    assert np.allclose(ad.cpu().numpy(), ref[3:6], atol=2e-6)
    assert not np.allclose(ad.cpu().numpy(), oracle.reparam_aux_rays(o, d, 1, 200.0, False, 11)[3:6], atol=1e-3)


def test_oracle_sample_stream_decorrelates_seed_and_pair(oracle):
    """The key of a sample is tea(seed, pair): (seed, pair + 1) and (seed + 1, pair) are different streams (they were
    the same stream when the key was seed + pair)."""
    rng = np.random.default_rng(3)
    _, d = _rays(2000, rng)
    a = oracle.reparam_aux_sample(d, 1, 50.0, False, 4)[0]
    b = oracle.reparam_aux_sample(d, 0, 50.0, False, 5)[0]
    c = oracle.reparam_aux_sample(d, 1, 50.0, False, 4)[0]
    assert np.array_equal(a, c) and np.abs(a - b).max() > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_gpu_reparam_gradient_is_invariant_under_a_partition_of_the_rays(hf, oracle, fused):
    """With ray_index the auxiliary samples of a ray do not depend on where it sits in the batch: the gradient of a
    render split into three shuffled parts equals the gradient of the whole (up to the order of the float atomics),
    and both equal the oracle's with the same ids."""
    import torch
    from hf_amd import shape as shape_mod
    h, f = _scene(oracle, seed=5)
    rng = np.random.default_rng(6)
    n = 6000
    o, d = _rays(n, rng)
    ids = (rng.permutation(1 << 20)[:n]).astype(np.uint32)
    gd = rng.normal(size=(3, n)).astype(np.float32); gdiv = rng.normal(size=n).astype(np.float32)
    need_ray_grads = not fused

    def grad(parts):
        shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=0.5)
        shape.heightfield.requires_grad_(True)
        loss = 0
        for sel in parts:
            oo = torch.from_numpy(o[:, sel]).cuda().requires_grad_(need_ray_grads)
            ray = hf.Ray3f(oo, torch.from_numpy(d[:, sel]).cuda())
            dirn, det = hf.reparameterize_ray(shape, ray, num_rays=6, kappa=300.0, antithetic=True, seed=3,
                                              ray_index=torch.from_numpy(ids[sel].view(np.int32)).cuda())
            loss = loss + (dirn * torch.from_numpy(gd[:, sel]).cuda()).sum() + (det * torch.from_numpy(gdiv[sel]).cuda()).sum()
        loss.backward()
        return shape.heightfield.grad.double().cpu().numpy()

    whole = grad([np.arange(n)])
    perm = rng.permutation(n)
    split = grad([perm[:1000], perm[1000:4100], perm[4100:]])
    with oracle.with_ray_ids(ids):
        ref = oracle.reparam_backward(f, o, d, gd, gdiv, num_rays=6, kappa=300.0, exponent=3.0, antithetic=True, seed=3)
    nr = np.linalg.norm(ref)
    assert nr > 0
    assert np.linalg.norm(whole - split) / nr <= 2e-6
    assert np.linalg.norm(whole - ref) / nr <= 3e-5


# ---- the three set-ups of src/render/tests/test_reparameterization.py:29-40 ---------------------------------
# (rectangle [-1,1]^2 at z = 0, num_rays = 32, kappa = 1e6, exponent = 3; the primary ray targets one SIDE of
# the rectangle, its CENTRE, one CORNER).  The reference moves the mesh by theta * (1,0,0) and checks that the
# derivative of the reparameterised direction equals the motion of the attached hit point,
#     <d/dtheta direction, trans> == <normalize(si.p + trans - o) - d, trans>   (atol 1e-2),
# with the other components insignificant.  A height field's parameter moves vertices along the object z axis,
# so theta lifts all heights (trans = max_height * z) and the rays are oblique, or the lift would be invisible.
_SETUPS = {"side": (0.0, 0.999), "centre": (0.0, 0.0), "corner": (0.99, -0.99)}
_REF_D = np.array([0.25, 0.15, -1.0]) / np.linalg.norm([0.25, 0.15, -1.0])


def _setup_ray(name):
    tx, ty = _SETUPS[name]
    tgt = np.array([tx, ty, 0.25])                       # on the flat surface z = 0.5 * 0.5
    o = tgt - 5.0 * _REF_D
    return o.astype(np.float32)[:, None], _REF_D.astype(np.float32)[:, None]


def _attached_motion(oracle, f, o, d, lift_z):
    r = np.concatenate([o, d, [[np.inf]]]).astype(np.float32)
    t, u, v, prim = f.ray_intersect_preliminary(r)
    assert np.isfinite(t[0])
    p = f.compute_surface_interaction(r, t, u, v, prim, oracle.RAY_ALL)["p"].astype(np.float64)
    eps = 1e-4
    new_d = p + np.array([[0.0], [0.0], [lift_z * eps]]) - o
    new_d /= np.linalg.norm(new_d)
    return (new_d - d.astype(np.float64)) / eps


@pytest.mark.parametrize("name", ["side", "centre", "corner"])
def test_oracle_reference_setups(oracle, name):
    h = np.full((9, 9), 0.5, np.float32)
    f = oracle.OracleField(h, max_height=0.5)
    o, d = _setup_ray(name)
    Vt, div = oracle.reparam_forward(f, o, d, np.ones_like(h, dtype=np.float64), num_rays=32, kappa=1e6, exponent=3.0)
    fd = _attached_motion(oracle, f, o, d, 0.5)
    trans = np.array([[0.0], [0.0], [1.0]])
    assert abs(float((Vt * trans).sum()) - float((fd * trans).sum())) < 1e-2          # the reference's assertion
    assert np.allclose(Vt, fd, atol=1e-2)                                               # and every component


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["side", "centre", "corner"])
def test_gpu_reference_setups(hf, oracle, name):
    """same set-ups through hf_amd.reparameterize_ray: the forward derivative along 'all heights rise' is read
    from reverse mode, V_theta[k] = <dL/dh, 1> for the upstream gradient e_k on the direction."""
    import torch
    h = np.full((9, 9), 0.5, np.float32)
    f = oracle.OracleField(h, max_height=0.5)
    o, d = _setup_ray(name)
    shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=0.5)
    shape.heightfield.requires_grad_(True)
    ray = hf.Ray3f(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
    Vt = np.zeros((3, 1))
    for k in range(3):
        shape.heightfield.grad = None
        dirn, det = hf.reparameterize_ray(shape, ray, num_rays=32, kappa=1e6, exponent=3.0, seed=0)
        dirn[k].sum().backward()
        Vt[k, 0] = float(shape.heightfield.grad.double().sum())
    fd = _attached_motion(oracle, f, o, d, 0.5)
    assert np.allclose(Vt, fd, atol=1e-2), (Vt.ravel(), fd.ravel())
    ref, _ = oracle.reparam_forward(f, o, d, np.ones_like(h, dtype=np.float64), num_rays=32, kappa=1e6, exponent=3.0)
    assert np.allclose(Vt, ref, atol=2e-4 * max(1.0, np.abs(ref).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("kappa,antithetic,num_rays", [(30.0, False, 5), (2000.0, True, 8)])
def test_gpu_reparameterize_ray_gradients_of_the_ray(hf, oracle, kappa, antithetic, num_rays):
    """reparam.py:296-325 also back-propagates to ray.o and ray.d.  The host mirror's analytic chain (hf_adjoint's
    grad_o / grad_d of the FollowShape hit, minus grad_p, through Frame3f(d)) against the oracle's float64 central
    differences of <gVd, V_direct(o, d)>; rays near the border so that some auxiliary rays miss (V_direct = d)."""
    import torch
    h, f = _scene(oracle, seed=5)
    rng = np.random.default_rng(6)
    n = 1500
    o, d = _rays(n, rng)
    o[0:2] *= 1.25                                   # spread the targets beyond the border
    shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=0.5)
    shape.heightfield.requires_grad_(True)
    ot = torch.from_numpy(o).cuda().requires_grad_(True); dt = torch.from_numpy(d).cuda().requires_grad_(True)
    dirn, det = hf.reparameterize_ray(shape, hf.Ray3f(ot, dt), num_rays=num_rays, kappa=kappa, exponent=3.0,
                                      antithetic=antithetic, seed=7)
    gd = rng.normal(size=(3, n)).astype(np.float32); gdiv = rng.normal(size=n).astype(np.float32)
    ((dirn * torch.from_numpy(gd).cuda()).sum() + (det * torch.from_numpy(gdiv).cuda()).sum()).backward()
    gh_ref, go_ref, gdr_ref = oracle.reparam_backward(f, o, d, gd, gdiv, num_rays=num_rays, kappa=kappa, exponent=3.0,
                                                      antithetic=antithetic, seed=7, ray_grads=True)
    got_h = shape.heightfield.grad.cpu().numpy().astype(np.float64)
    assert np.linalg.norm(got_h - gh_ref) <= 3e-5 * np.linalg.norm(gh_ref)
    go, gdr = ot.grad.cpu().numpy().astype(np.float64), dt.grad.cpu().numpy().astype(np.float64)
    assert np.linalg.norm(go_ref) > 0 and np.linalg.norm(gdr_ref) > 0
    # some auxiliary rays must have missed for the V_direct = d branch to be exercised
    r = oracle.reparam_aux_rays(o, d, 0, kappa, antithetic, 7)
    assert 0.02 < np.isinf(f.ray_intersect_preliminary(r)[0]).mean() < 0.9
    assert np.linalg.norm(go - go_ref) <= 2e-4 * np.linalg.norm(go_ref), np.linalg.norm(go - go_ref) / np.linalg.norm(go_ref)
    assert np.linalg.norm(gdr - gdr_ref) <= 2e-4 * np.linalg.norm(gdr_ref), np.linalg.norm(gdr - gdr_ref) / np.linalg.norm(gdr_ref)
