"""The reference's own known answers, checked against the HIP kernels DIRECTLY (no oracle in between): the driver's
`-m gpu` run then exercises the pins themselves, not only HIP == oracle.  Scene as in the reference tests: one
rectangle-sized cell (2 x 2 vertices, z = 0, max_height = 1, to_world = identity), i.e. two triangles.
  * src/render/tests/test_mesh.py:536-638 (test16), z components of the exact gradient vectors
  * src/render/tests/test_mesh.py:674-735 (test17) / mesh.cpp:695-752: default vs FollowShape vs DetachShape
  * src/render/tests/test_mesh.py:918-958 (test22): boundary-test magnitudes
  * src/render/tests/test_kdtrees.py:52-82: staircase depths t = 2 - step / n
  * src/core/tests/test_random.py:8-16: sample_tea_32 through the workload generator that feeds every GPU test
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cell(hf, heights=None):
    h = torch.zeros((2, 2)) if heights is None else torch.as_tensor(heights, dtype=torch.float32)
    return hf.Heightfield(heightfield=h.cuda(), max_height=1.0)


def _ray(hf, o, d):
    o = torch.tensor(o, dtype=torch.float32).reshape(3, 1).cuda()
    d = torch.tensor(d, dtype=torch.float32).reshape(3, 1).cuda()
    return hf.Ray3f(o, d, torch.full((1,), float("inf")).cuda())


ROWS = {"t": (0, 1), "p": (1, 4), "n": (4, 7), "uv": (7, 9), "sh_n": (9, 12), "dp_du": (12, 15), "dp_dv": (15, 18)}


def _grad(hf, shape, ray, pi, field, comp, flags=None):
    g = torch.zeros((18, 1), device="cuda")
    g[ROWS[field][0] + comp] = 1.0
    return shape.adjoint(ray, pi, g, ray_flags=hf.RayFlags.All if flags is None else flags).cpu().numpy()


def test16_gradient_vectors_z_components(hf):
    shape = _cell(hf)
    ray = _ray(hf, [0.99999, 0.99999, -10.0], [0, 0, 1])
    pi = shape.ray_intersect_preliminary(ray)
    assert abs(float(pi.t[0]) - 10.0) < 1e-5
    z4 = lambda G: G[1, 1]; z1 = lambda G: G[0, 1]; z2 = lambda G: G[1, 0]; z3 = lambda G: G[0, 0]
    G = _grad(hf, shape, ray, pi, "t", 0)
    assert np.allclose([z1(G), z2(G), z3(G), z4(G)], [0, 0, 0, 1], atol=1e-5)
    G = _grad(hf, shape, ray, pi, "p", 2)
    assert np.allclose([z1(G), z2(G), z3(G), z4(G)], [0, 0, 0, 1], atol=1e-5)
    for field in ("n", "sh_n"):
        G = _grad(hf, shape, ray, pi, field, 0)
        assert np.allclose([z1(G), z2(G), z3(G), z4(G)], [0, 0.5, 0, -0.5], atol=1e-5)
        G = _grad(hf, shape, ray, pi, field, 1)
        assert np.allclose([z1(G), z2(G), z3(G), z4(G)], [0.5, 0, 0, -0.5], atol=1e-5)
    for field, comp in (("dp_du", 0), ("dp_du", 1), ("dp_dv", 0), ("dp_dv", 1)):
        assert np.allclose(_grad(hf, shape, ray, pi, field, comp), 0, atol=1e-6)


def test17_default_followshape_detachshape(hf):
    shape = _cell(hf)
    d = np.array([0.3, 0.1, -1.0], np.float32)
    ray = _ray(hf, [-0.5, 0.1, 2.0], d)
    pi = shape.ray_intersect_preliminary(ray)
    assert abs(float(pi.t[0]) - 2.0) < 1e-6
    A = int(hf.RayFlags.All)
    dsum = lambda field, comp, flags: float(_grad(hf, shape, ray, pi, field, comp, flags).sum())   # uniform lift
    assert abs(dsum("t", 0, A) - (-1.0)) < 1e-5
    for k in range(3):
        assert abs(dsum("p", k, A) - (-d[k])) < 1e-5
    assert abs(dsum("uv", 0, A) - (-d[0] * 0.5)) < 1e-5
    Fl = A | int(hf.RayFlags.FollowShape)
    assert np.allclose([dsum("p", 0, Fl), dsum("p", 1, Fl), dsum("p", 2, Fl)], [0, 0, 1], atol=1e-5)
    assert abs(dsum("uv", 0, Fl)) < 1e-6 and abs(dsum("uv", 1, Fl)) < 1e-6
    De = A | int(hf.RayFlags.DetachShape)
    assert dsum("t", 0, De) == 0 and dsum("p", 2, De) == 0
    with pytest.raises(RuntimeError, match="DetachShape \\| FollowShape"):
        shape.ray_intersect(ray, A | int(hf.RayFlags.DetachShape) | int(hf.RayFlags.FollowShape))


def test22_boundary_test_magnitudes(hf):
    shape = _cell(hf)
    flags = int(hf.RayFlags.All) | int(hf.RayFlags.BoundaryTest)

    def B(o):
        si = shape.ray_intersect(_ray(hf, o, [0, 0, 1]), flags)
        return float(si.boundary_test[0]), bool(si.is_valid()[0])
    b, valid = B([2, 0, -1]); assert not valid and b > 1e6
    b, valid = B([0.9999, 0.9999, -1]); assert valid and b < 1e-3
    b, valid = B([0.99999, 0.0, -1]); assert valid and b < 1e-4
    b, valid = B([0.9, 0.0, -1]); assert valid and b > 1e-1


def test_kdtree_staircase_depths(hf):
    """test_kdtrees.py:52-82: n steps of height k / n seen from above: t = 2 - step / n, and ray_test == is_valid"""
    n = 10
    W = 4 * n + 1
    cols = np.minimum(np.arange(W) // 4, n - 1)
    h = np.repeat((cols / n)[None, :], 3, 0).astype(np.float32)          # 3 rows x W columns, plateaus of 4 cells
    shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=1.0)
    xs = np.array([-1 + 2 * (4 * k + 2) / (W - 1) for k in range(n)], np.float32)    # the middle of every plateau
    o = np.stack([xs, np.zeros(n, np.float32), np.full(n, 2.0, np.float32)])
    d = np.stack([np.zeros(n), np.zeros(n), -np.ones(n)]).astype(np.float32)
    ray = hf.Ray3f(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), torch.full((n,), float("inf")).cuda())
    pi = shape.ray_intersect_preliminary(ray)
    assert np.allclose(pi.t.cpu().numpy(), 2.0 - np.arange(n) / n, atol=1e-6)
    assert bool(shape.ray_test(ray).all())


def test_sample_tea_32_known_answers(hf):
    """test_random.py:8-16 through hf_amd.workload.tea32 (the jitter of every synthetic wavefront) and through the
    device's own TEA (hf_reparam_aux_rays draws its samples from sample_tea_32): sample_tea_float32 =
    bits((v1 >> 9) | 0x3f800000) - 1 (random.h:136-140)"""
    import struct
    expected = {(1, 1): 0.5424730777740479, (1, 2): 0.5079904794692993, (1, 3): 0.4171961545944214,
                (1, 4): 0.008385419845581055, (1, 5): 0.8085528612136841, (2, 1): 0.6939879655838013,
                (3, 1): 0.6978365182876587, (4, 1): 0.4897364377975464}
    keys = list(expected)
    a, b = hf.workload.tea32(torch.tensor([k[0] for k in keys]), torch.tensor([k[1] for k in keys]))
    for k, key in enumerate(keys):
        got = struct.unpack("f", struct.pack("I", (int(b[k]) >> 9) | 0x3F800000))[0] - 1.0
        assert got == np.float32(expected[key]), key


def test_boundary_test_all_edges_is_the_mesh_sdf(hf, oracle):
    """HF_RAY_BOUNDARY_ALL_EDGES (libhf extension bit): boundary_test over all three edges of the hit triangle --
    the reference Mesh's per-triangle SDF (mesh.cpp:845-890): in [0, 1] everywhere, equal to the oracle's, and
    <= the default (silhouette edges only), which is exactly 1 on triangles without a silhouette edge."""
    rng = np.random.default_rng(4)
    h = rng.uniform(0, 1, (33, 21)).astype(np.float32)
    shape = hf.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=0.3)
    import common
    r = common.random_rays(20000, rng, 0.3)
    rt = torch.from_numpy(r).cuda()
    ray = hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous())
    base = int(hf.RayFlags.All) | int(hf.RayFlags.BoundaryTest)
    si_all = shape.ray_intersect(ray, base | int(hf.RayFlags.BoundaryAllEdges))
    si_sil = shape.ray_intersect(ray, base)
    v = si_all.is_valid().cpu().numpy()
    b_all, b_sil = si_all.boundary_test.cpu().numpy()[v], si_sil.boundary_test.cpu().numpy()[v]
    assert v.sum() > 5000 and b_all.min() >= 0.0 and b_all.max() <= 1.0 + 1e-5
    assert np.all(b_all <= b_sil + 1e-6) and (b_sil == 1.0).sum() > 100
    f = oracle.OracleField(h, max_height=0.3)
    t, u, vv, prim = f.ray_intersect_preliminary(r)
    rec = f.compute_surface_interaction(r, t, u, vv, prim, base | 0x10000)
    assert np.allclose(si_all.boundary_test.cpu().numpy()[v], rec["boundary_test"][v], rtol=1e-4, atol=1e-5)
