"""Pins the CPU oracle against every known answer the reference's own tests hold for the
semantics this path restates.  The reference has no heightfield, so
each test states which reference test it ports and how the scene maps onto a height grid.
CPU only."""
import struct

import numpy as np
import pytest

import common
import si_numpy as S


def _rays(o, d, maxt=np.inf):
    o = np.asarray(o, np.float32).reshape(-1, 3).T
    d = np.asarray(d, np.float32).reshape(-1, 3).T
    n = max(o.shape[1], d.shape[1])
    o = np.broadcast_to(o, (3, n)); d = np.broadcast_to(d, (3, n))
    return np.concatenate([o, d, np.full((1, n), maxt, np.float32)]).astype(np.float32)


def test_sample_tea_known_answers(oracle):
    """src/core/tests/test_random.py:8-16 (sample_tea_float32 = bits((v1 >> 9) | 0x3f800000) - 1,
    include/mitsuba/core/random.h:136-140)"""
    expected = {(1, 1): 0.5424730777740479, (1, 2): 0.5079904794692993, (1, 3): 0.4171961545944214,
                (1, 4): 0.008385419845581055, (1, 5): 0.8085528612136841, (2, 1): 0.6939879655838013,
                (3, 1): 0.6978365182876587, (4, 1): 0.4897364377975464}
    for (v0, v1), want in expected.items():
        _, w1 = oracle.sample_tea_32(v0, v1, 4)
        got = struct.unpack("f", struct.pack("I", (w1 >> 9) | 0x3F800000))[0] - 1.0
        assert got == np.float32(want)
    # the workload generator's torch restatement agrees with the C one
    import torch
    import hf_amd
    a, b = hf_amd.workload.tea32(torch.tensor([1, 2, 3, 4]), torch.tensor([1, 1, 1, 5]))
    for k, (v0, v1) in enumerate([(1, 1), (2, 1), (3, 1), (4, 5)]):
        assert (int(a[k]), int(b[k])) == oracle.sample_tea_32(v0, v1, 4)


def test_staircase_depth(oracle):
    """src/render/tests/test_kdtrees.py:52-82: 20-step staircase, 127^2 vertical rays from z=2,
    accelerated == brute force, ray_test consistent, t = 2 - step/n_steps.  As a height grid the
    risers are one cell wide, so the analytic value is checked on rays over the flat treads."""
    n_steps, per = 20, 4
    H = n_steps * per + 1
    rows = np.minimum(np.arange(H) // per, n_steps - 1)
    h = np.repeat((rows / n_steps)[:, None], 9, 1).astype(np.float32)   # tread k: rows [4k, 4k+3], riser 4k+3 -> 4k+4
    # scale the shape so that object [-1,1]^2 -> world [0,1]^2 like the reference's staircase mesh
    tw = np.array([[0.5, 0, 0, 0.5], [0, 0.5, 0, 0.5], [0, 0, 1, 0]], np.float32)
    f = oracle.OracleField(h, max_height=1.0, to_world=tw)
    n = 128
    xs = np.arange(n - 1) / (n - 1.0)
    X, Y = np.meshgrid(xs, xs)
    r = _rays(np.stack([X.ravel(), Y.ravel(), np.full(X.size, 2.0)], 1), [0, 0, -1], 100.0)
    t_naive, _, _, prim_naive = f.ray_intersect_preliminary(r, naive=True)
    t, _, _, prim = f.ray_intersect_preliminary(r)
    shadow = f.ray_test(r)
    assert shadow.all() and np.array_equal(shadow, np.isfinite(t_naive))
    assert np.array_equal(t, t_naive) and np.array_equal(prim, prim_naive)
    row_f = Y.ravel() * (H - 1)                       # fractional grid row of each ray
    on_tread = (np.floor(row_f) % per) != per - 1     # not on a riser cell
    on_tread &= np.floor(row_f) < H - 1
    step = np.minimum(np.floor(row_f) // per, n_steps - 1)
    assert np.allclose(t_naive[on_tread], 2.0 - step[on_tread] / n_steps, atol=1e-6)


def test_rectangle_hit_count(oracle):
    """src/shapes/tests/test_rectangle.py:34-59: a flat grid is the rectangle; to_world =
    scale(2, 0.5, 1); 15 rays (a, a, 5) -> (0,0,-1); hit iff |a| <= 0.5; 7 hits."""
    tw = np.array([[2, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, 1, 0]], np.float32)
    for (W, H) in [(2, 2), (5, 4), (33, 17)]:
        f = oracle.OracleField(np.zeros((H, W), np.float32), 1.0, to_world=tw)
        a = np.linspace(-1, 1, 15).astype(np.float32)
        r = _rays(np.stack([a, a, np.full(15, 5.0)], 1), [0, 0, -1])
        found = f.ray_test(r)
        t, u, v, prim = f.ray_intersect_preliminary(r)
        assert np.array_equal(found, np.abs(a) <= 0.5)
        assert np.array_equal(np.isfinite(t), found) and found.sum() == 7
        assert np.allclose(t[found], 5.0)
        # uv = 0.5 * local + 0.5 (rectangle.cpp:312-313)
        si = f.compute_surface_interaction(r, t, u, v, prim)
        assert np.allclose(si["uv"][0][found], 0.5 * (a[found] / 2.0) + 0.5, atol=1e-6)
        assert np.allclose(si["uv"][1][found], 0.5 * (a[found] / 0.5) + 0.5, atol=1e-6)
        assert np.allclose(si["n"][:, found], np.array([[0], [0], [1.0]]), atol=1e-6)


def _unit_cell(oracle, flip=False):
    """the reference's rectangle.obj ([-1,1]^2 at z=0) as a 2x2 height grid: vertices
    v00=(-1,-1) v10=(1,-1) v01=(-1,1) v11=(1,1); reference numbering (from the expected
    vectors of test_mesh.py:569-622): vertex 1=(1,-1), 2=(-1,1), 4=(1,1)."""
    return oracle.OracleField(np.zeros((2, 2), np.float32), 1.0, flip_normals=flip)


def test_mesh_param_gradients_backward(oracle):
    """src/render/tests/test_mesh.py:536-638 (test16), z components: ray (0.99999, 0.99999, -10)
    -> +z hits next to the 4th vertex.  d(si.t)/dz4 = 1, d(si.p.z)/dz4 = 1,
    d(si.n.x)/dz = +0.5 @ vertex 2, -0.5 @ vertex 4; d(si.n.y)/dz = +0.5 @ vertex 1, -0.5 @ vertex 4;
    same for sh_frame.n.  Height gradient == z gradient for max_height = 1, to_world = I."""
    f = _unit_cell(oracle)
    r = _rays([0.99999, 0.99999, -10.0], [0, 0, 1])
    t, u, v, prim = f.ray_intersect_preliminary(r)
    assert np.isfinite(t[0]) and abs(t[0] - 10.0) < 1e-5

    def grad(field, comp):
        g = {field: np.zeros((dict(oracle.GRAD_FIELDS)[field], 1), np.float32)}
        g[field][comp, 0] = 1.0
        return f.adjoint(r, t, u, v, prim, g)   # [[v00, v10], [v01, v11]]

    z4 = lambda G: G[1, 1]; z1 = lambda G: G[0, 1]; z2 = lambda G: G[1, 0]; z3 = lambda G: G[0, 0]
    G = grad("t", 0)
    assert np.allclose([z1(G), z2(G), z3(G), z4(G)], [0, 0, 0, 1], atol=1e-5)
    G = grad("p", 2)
    assert np.allclose([z1(G), z2(G), z3(G), z4(G)], [0, 0, 0, 1], atol=1e-5)
    for field in ("n", "sh_n"):
        G = grad(field, 0)
        assert np.allclose([z1(G), z2(G), z3(G), z4(G)], [0, 0.5, 0, -0.5], atol=1e-5)
        G = grad(field, 1)
        assert np.allclose([z1(G), z2(G), z3(G), z4(G)], [0.5, 0, 0, -0.5], atol=1e-5)
    # dp_du.x / dp_dv.y only depend on x/y of the vertices: no height gradient (zero z entries in test16)
    for field, comp in (("dp_du", 0), ("dp_du", 1), ("dp_dv", 0), ("dp_dv", 1)):
        assert np.allclose(grad(field, comp), 0, atol=1e-6)


def test_follow_vs_default_semantics(oracle):
    """src/render/tests/test_mesh.py:674-735 (test17) and the mode table of mesh.cpp:695-752, for the
    height parameter: raising the whole surface by dz under an oblique ray
      default     : the hit stays on the ray  -> dp = d * dt, uv moves
      FollowShape : the hit is glued          -> dp = (0,0,dz), uv fixed
      DetachShape : no height gradient at all."""
    f = _unit_cell(oracle)
    d = np.array([0.3, 0.1, -1.0], np.float32)
    r = _rays([-0.5, 0.1, 2.0], d)
    t, u, v, prim = f.ray_intersect_preliminary(r)
    assert abs(t[0] - 2.0) < 1e-6
    def dsum(field, comp, flags):
        g = {field: np.zeros((dict(oracle.GRAD_FIELDS)[field], 1), np.float32)}
        g[field][comp, 0] = 1.0
        return f.adjoint(r, t, u, v, prim, g, flags).sum()   # derivative w.r.t. a uniform lift of all heights
    A = oracle.RAY_ALL
    # default: z(t) = 2 - t, so lifting the surface by dz gives dt/dz = -1 and dp = d * dt
    assert abs(dsum("t", 0, A) - (-1.0)) < 1e-5
    for k in range(3):
        assert abs(dsum("p", k, A) - (-d[k])) < 1e-5
    assert abs(dsum("uv", 0, A) - (-d[0] * 0.5)) < 1e-5        # uv = 0.5*x + 0.5
    # FollowShape: glued point, only z moves, uv fixed
    Fl = A | oracle.RAY_FOLLOWSHAPE
    assert np.allclose([dsum("p", 0, Fl), dsum("p", 1, Fl), dsum("p", 2, Fl)], [0, 0, 1], atol=1e-5)
    assert abs(dsum("uv", 0, Fl)) < 1e-6 and abs(dsum("uv", 1, Fl)) < 1e-6
    # DetachShape
    De = A | oracle.RAY_DETACHSHAPE
    assert dsum("t", 0, De) == 0 and dsum("p", 2, De) == 0
    # DetachShape | FollowShape throws (mesh.cpp:709-711)
    with pytest.raises(RuntimeError, match="DetachShape \\| FollowShape"):
        f.compute_surface_interaction(r, t, u, v, prim, A | oracle.RAY_DETACHSHAPE | oracle.RAY_FOLLOWSHAPE)


def test_boundary_test_magnitudes(oracle):
    """src/render/tests/test_mesh.py:918-958 (test22, face normals): miss > 1e6; 1e-4 from a corner
    < 1e-3; 1e-5 from an edge < 1e-4; 0.1 from an edge > 1e-1 (rays from z=-1 along +z)."""
    f = _unit_cell(oracle)
    flags = oracle.RAY_ALL | oracle.RAY_BOUNDARYTEST
    def B(o):
        r = _rays(o, [0, 0, 1])
        t, u, v, prim = f.ray_intersect_preliminary(r)
        return f.compute_surface_interaction(r, t, u, v, prim, flags)["boundary_test"][0], np.isfinite(t[0])
    b, valid = B([2, 0, -1]); assert not valid and b > 1e6
    b, valid = B([0.9999, 0.9999, -1]); assert valid and b < 1e-3
    b, valid = B([0.99999, 0.0, -1]); assert valid and b < 1e-4
    b, valid = B([0.9, 0.0, -1]); assert valid and b > 1e-1


def test_closest_hit_tie_rule(oracle):
    """kdtree.h:2424-2448: `t <= ray.maxt` lets a later primitive with the same t replace the
    earlier one, so an exact tie goes to the highest prim_index.  A vertical ray through an
    interior vertex of a flat grid touches 6 triangles at the same t."""
    f = oracle.OracleField(np.full((5, 5), 0.5, np.float32), 1.0)
    r = _rays([0.0, 0.0, 3.0], [0, 0, -1])          # vertex (row 2, col 2)
    t, u, v, prim = f.ray_intersect_preliminary(r, naive=True)
    t2, _, _, prim2 = f.ray_intersect_preliminary(r)
    assert t[0] == 2.5 and prim[0] == prim2[0]
    # cells around the vertex: (1,1),(2,1),(1,2),(2,2) -> prims 2*(cy*4+cx)+tri; highest index wins
    assert prim[0] == 2 * (2 * 4 + 2) + 0


@pytest.mark.parametrize("W,H", [(2, 2), (3, 5), (17, 9), (64, 64), (100, 37)])
@pytest.mark.parametrize("kind", ["rand", "sine", "stairs", "flat"])
def test_hierarchical_equals_naive(oracle, W, H, kind):
    """test_kdtrees.py:52-82 methodology (accelerated == ray_intersect_naive) on random,
    secondary-ray-like and degenerate (grid-aligned) rays."""
    rng = np.random.default_rng(W * 1000 + H)
    h = common.heights(kind, W, H, rng)
    f = oracle.OracleField(h, max_height=0.5)
    nrand = 1500 if W * H > 2000 else 4000
    r = np.concatenate([common.random_rays(nrand, rng), common.inside_rays(nrand // 2, rng)], 1)
    if W * H <= 33 * 33:
        xs = np.array([f.vertex(0, j)[0] for j in range(W)]); ys = np.array([f.vertex(i, 0)[1] for i in range(H)])
        r = np.concatenate([r, common.structured_rays(xs, ys)], 1)
    a = f.ray_intersect_preliminary(r, naive=True)
    b = f.ray_intersect_preliminary(r)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert np.array_equal(f.ray_test(r), np.isfinite(a[0]))
    assert np.array_equal(f.ray_test(r, naive=True), np.isfinite(a[0]))


def test_bbox_and_maxt(oracle):
    """bbox under a transform (test_rectangle.py:15-31 pattern) and the inclusive maxt test (mesh.h:377)."""
    rng = np.random.default_rng(1)
    h = common.heights("rand", 9, 9, rng)
    tw = common.affine(3)
    f = oracle.OracleField(h, 0.7, to_world=tw)
    A = tw.astype(np.float64)
    corners = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (h.min() * 0.7, h.max() * 0.7)])
    w = corners @ A[:, :3].T + A[:, 3]
    bb = f.bbox()
    assert np.allclose(bb[:3], w.min(0), atol=1e-5) and np.allclose(bb[3:], w.max(0), atol=1e-5)
    g = oracle.OracleField(np.zeros((3, 3), np.float32), 1.0)
    r = _rays([0.3, 0.2, 2.0], [0, 0, -1])
    for maxt, want in ((2.0, True), (np.nextafter(np.float32(2.0), np.float32(0)), False), (0.0, False)):
        r[6] = maxt
        assert g.ray_test(r)[0] == want


def test_heightfield_boundary_test_silhouettes(oracle):
    """for a height field the per-triangle SDF of mesh.cpp:863-890 must only see SILHOUETTE
    edges -- the grid border (the rectangle's border distance, rectangle.cpp:318-319) and edges where the facing
    towards the ray flips (the grazing silhouettes of mesh.cpp:892-898) -- not every interior edge.
      * flat interior, seen from above: no silhouette edge anywhere near -> B = 1 (the incentre value), also exactly
        on interior edges and vertices;
      * the same point near the border: B = distance to the border in triangle units -> 0;
      * a ridge seen from the side: the far slope faces away; on the near slope B -> 0 towards the crest and is 1
        further down; seen from above both slopes face the ray and the crest is no boundary."""
    flags = oracle.RAY_ALL | oracle.RAY_BOUNDARYTEST
    def B(f, o, d):
        r = _rays(o, d)
        t, u, v, prim = f.ray_intersect_preliminary(r)
        assert np.isfinite(t[0])
        return float(f.compute_surface_interaction(r, t, u, v, prim, flags)["boundary_test"][0])
    flat = oracle.OracleField(np.full((9, 9), 0.5, np.float32), 1.0)
    for (x, y) in [(0.1, 0.05), (0.0, 0.0), (0.25, 0.1), (0.125, -0.3)]:       # interior: faces, a vertex, edges
        assert B(flat, [x, y, 3.0], [0, 0, -1]) == 1.0
    assert B(flat, [0.99999, 0.1, 3.0], [0, 0, -1]) < 1e-3                       # border
    assert B(flat, [-0.3, -0.99999, 3.0], [0.05, 0.0, -1]) < 1e-3
    assert 0.05 < B(flat, [0.98, 0.1, 3.0], [0, 0, -1]) < 1.0
    # ridge along y at x = 0: z = 0.5 - |x|  (slopes +-1), 17 columns -> the crest is the vertex column 8
    W = 17
    xs = np.linspace(-1, 1, W)
    ridge = oracle.OracleField(np.repeat((0.5 - 0.5 * np.abs(xs))[None, :], 9, 0).astype(np.float32), 1.0)
    side = np.array([-1.0, 0.0, -0.2]); side /= np.linalg.norm(side)             # from +x, nearly horizontal
    def from_side(x_hit):
        z = 0.5 - 0.5 * abs(x_hit)
        o = np.array([x_hit, 0.03, z]) - 3.0 * side
        return B(ridge, o, side)
    near_crest = from_side(0.004)       # on the near (+x) slope, 0.004 from the crest: one cell = 0.125 wide
    lower = from_side(0.3)
    assert near_crest < 0.1 and lower == 1.0, (near_crest, lower)
    assert B(ridge, [0.004, 0.03, 3.0], [0, 0, -1]) == 1.0                       # from above the crest is no silhouette


def test_boundary_test_next_to_a_silhouette_vertex(oracle):
    """A known limit of the silhouette-edge boundary test, pinned so that it cannot change unnoticed: a
    triangle that touches the silhouette only at a VERTEX has no silhouette edge, so B stays 1 (the incentre value)
    right up to that vertex, while the neighbouring triangle that owns the border edge goes to 0 there.  Hosts that
    need the reference Mesh's SDF over all three edges (mesh.cpp:845-890, B in [0, 1], -> 0 at every vertex) set
    HF_RAY_BOUNDARY_ALL_EDGES (0x10000)."""
    f = oracle.OracleField(np.full((9, 9), 0.5, np.float32), 1.0)      # 8 x 8 cells of 0.25; the silhouette is the border
    sil = oracle.RAY_ALL | oracle.RAY_BOUNDARYTEST
    seen = set()
    for dy in (1e-4, -1e-4):                                            # either side of the grid line y = -0.25, 0.001 from x = 1
        r = _rays([0.999, -0.25 + dy, 3.0], [0, 0, -1])
        t, u, v, prim = f.ray_intersect_preliminary(r)
        assert np.isfinite(t[0]) and (prim[0] >> 1) % 8 == 7            # a cell of the last column
        b_sil = float(f.compute_surface_interaction(r, t, u, v, prim, sil)["boundary_test"][0])
        b_all = float(f.compute_surface_interaction(r, t, u, v, prim, sil | 0x10000)["boundary_test"][0])
        assert 0.0 <= b_all < 0.05                                      # all edges: the vertex is on two of them
        if prim[0] & 1:                                                 # triangle (v11, v01, v10): its right edge IS the border
            assert b_sil < 0.05
        else:                                                           # triangle (v00, v10, v01): touches the border at v10 only
            assert b_sil == 1.0
        seen.add(int(prim[0] & 1))
    assert seen == {0, 1}
