"""Ray-gradient known answers of the reference's rectangle tests, ported to a flat height grid
(a flat grid IS the rectangle [-1,1]^2 at z = 0).

  src/shapes/tests/test_rectangle.py:116-155  test06 (forward mode, ray (-0.3,-0.3,-10) -> +z):
      d si.p / d o.x = [1,0,0]   d si.p / d o.y = [0,1,0]   d si.uv / d o.x = [0.5,0]
      d si.t / d o.z = -1        d si.p / d d.x = [10,0,0]
  src/shapes/tests/test_rectangle.py:158-173  test07 (backward): backward(si.p.x) -> grad(ray.o) = [1,0,0];
      backward(si.t) -> grad(ray.o) = [0,0,-1]
  src/shapes/tests/test_rectangle.py:176-272  test08, the parts a ray can express: with DetachShape the ray
      gradient is all there is; with FollowShape the ray does not move the glued point.

Our adjoint is reverse mode; one backward pass with a one-hot upstream gradient on output component c returns
the row d out_c / d (o, d) of the Jacobian, so the forward-mode columns above are read off three passes.
CPU: the oracle.  GPU twin: hf_adjoint(..., grad_o, grad_d) through the C ABI on the same rays, several grids.
"""
import numpy as np
import pytest

GRIDS = [(2, 2), (5, 4), (33, 17)]


def _ray():
    return np.array([[-0.3], [-0.3], [-10.0], [0.0], [0.0], [1.0], [np.inf]], np.float32)


def _rows_oracle(oracle, W, H, flags=None):
    f = oracle.OracleField(np.zeros((H, W), np.float32), 1.0)
    r = _ray()
    t, u, v, prim = f.ray_intersect_preliminary(r)
    assert np.isfinite(t[0]) and abs(t[0] - 10.0) < 1e-6
    flags = oracle.RAY_ALL if flags is None else flags

    def row(field, comp):
        g = {field: np.zeros((dict(oracle.GRAD_FIELDS)[field], 1), np.float32)}
        g[field][comp, 0] = 1.0
        _, go, gd = f.adjoint(r, t, u, v, prim, g, flags, ray_grads=True)
        return go[:, 0], gd[:, 0]
    return row


def _rows_gpu(hf, W, H, flags=None):
    import torch
    shape = hf.Heightfield(heightfield=torch.zeros(H, W).cuda(), max_height=1.0)
    r = torch.from_numpy(_ray()).cuda()
    flags = hf.RayFlags.All if flags is None else flags

    def row(field, comp):
        o = r[0:3].clone().requires_grad_(True); d = r[3:6].clone().requires_grad_(True)
        si = shape.ray_intersect(hf.Ray3f(o, d, r[6].clone()), flags)
        out = getattr(si, field)
        (out[comp] if out.dim() == 2 else out).sum().backward()
        return o.grad[:, 0].cpu().numpy(), d.grad[:, 0].cpu().numpy()
    return row


def _check_known_answers(row):
    # test06: columns of the Jacobian, read from the rows
    dp_do = np.stack([row("p", k)[0] for k in range(3)])       # [k, j] = d p_k / d o_j
    dp_dd = np.stack([row("p", k)[1] for k in range(3)])
    assert np.allclose(dp_do[:, 0], [1, 0, 0], atol=1e-5)      # d p / d o.x
    assert np.allclose(dp_do[:, 1], [0, 1, 0], atol=1e-5)      # d p / d o.y
    assert np.allclose(dp_dd[:, 0], [10, 0, 0], atol=1e-4)     # d p / d d.x  (t = 10)
    duv_do = np.stack([row("uv", k)[0] for k in range(2)])
    assert np.allclose(duv_do[:, 0], [0.5, 0], atol=1e-5)      # d uv / d o.x
    go, gd = row("t", 0)
    assert abs(go[2] - (-1.0)) < 1e-5                          # d t / d o.z
    # test07: backward
    assert np.allclose(row("p", 0)[0], [1, 0, 0], atol=1e-5)
    assert np.allclose(go, [0, 0, -1], atol=1e-5)
    # the plane is z = 0: the hit cannot leave it, and the normal does not depend on the ray
    assert np.allclose(dp_do[2], 0, atol=1e-5) and np.allclose(dp_dd[2], 0, atol=1e-4)
    for k in range(3):
        assert np.allclose(row("n", k)[0], 0, atol=1e-6) and np.allclose(row("n", k)[1], 0, atol=1e-6)


@pytest.mark.parametrize("W,H", GRIDS)
def test_oracle_rectangle_ray_gradients(oracle, W, H):
    _check_known_answers(_rows_oracle(oracle, W, H))
    # DetachShape (test08 "Test 00" seen from the ray): the ray gradient is unchanged
    _check_known_answers(_rows_oracle(oracle, W, H, oracle.RAY_ALL | oracle.RAY_DETACHSHAPE))


@pytest.mark.parametrize("W,H", GRIDS)
def test_oracle_follow_shape_ray_gradients(oracle, W, H):
    """FollowShape (mesh.cpp:748-752): barycentrics are detached, so p and uv do not follow the ray;
    t = |p - o| / |d| still does: d t / d o.z = -1, d t / d d.z = -t."""
    row = _rows_oracle(oracle, W, H, oracle.RAY_ALL | oracle.RAY_FOLLOWSHAPE)
    for k in range(3):
        assert np.allclose(row("p", k)[0], 0, atol=1e-6) and np.allclose(row("p", k)[1], 0, atol=1e-6)
    go, gd = row("t", 0)
    assert np.allclose(go, [0, 0, -1], atol=1e-5) and abs(gd[2] - (-10.0)) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", GRIDS)
def test_gpu_rectangle_ray_gradients(hf, W, H):
    _check_known_answers(_rows_gpu(hf, W, H))
    _check_known_answers(_rows_gpu(hf, W, H, hf.RayFlags.All | hf.RayFlags.DetachShape))


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", GRIDS)
def test_gpu_follow_shape_ray_gradients(hf, W, H):
    row = _rows_gpu(hf, W, H, hf.RayFlags.All | hf.RayFlags.FollowShape)
    for k in range(3):
        assert np.allclose(row("p", k)[0], 0, atol=1e-6) and np.allclose(row("p", k)[1], 0, atol=1e-6)
    go, gd = row("t", 0)
    assert np.allclose(go, [0, 0, -1], atol=1e-5) and abs(gd[2] - (-10.0)) < 1e-4


# ---- src/render/tests/test_mesh.py:458-531 (test15, forward mode w.r.t. the vertex positions), the part a height
# grid can express: translating every vertex along z == lifting every height (max_height = 1, to_world = I):
#     ray (-0.2,-0.3,-10) -> +z:   d si.t / dz = 1,   d si.p / dz = [0,0,1];   uv does not depend on z.
# Forward mode over a uniform lift = the SUM over all texels of one reverse-mode row.
def _ray15():
    return np.array([[-0.2], [-0.3], [-10.0], [0.0], [0.0], [1.0], [np.inf]], np.float32)


def _check_lift_answers(lift):
    assert abs(lift("t", 0) - 1.0) < 1e-5
    assert np.allclose([lift("p", k) for k in range(3)], [0, 0, 1], atol=1e-5)
    assert np.allclose([lift("uv", k) for k in range(2)], [0, 0], atol=1e-6)


@pytest.mark.parametrize("W,H", GRIDS)
def test_oracle_uniform_lift_known_answers(oracle, W, H):
    f = oracle.OracleField(np.zeros((H, W), np.float32), 1.0)
    r = _ray15()
    t, u, v, prim = f.ray_intersect_preliminary(r)
    assert abs(t[0] - 10.0) < 1e-6

    def lift(field, comp):
        g = {field: np.zeros((dict(oracle.GRAD_FIELDS)[field], 1), np.float32)}
        g[field][comp, 0] = 1.0
        return float(f.adjoint(r, t, u, v, prim, g).sum())
    _check_lift_answers(lift)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", GRIDS)
def test_gpu_uniform_lift_known_answers(hf, W, H):
    import torch
    r = torch.from_numpy(_ray15()).cuda()

    def lift(field, comp):
        h = torch.zeros(H, W, device="cuda", requires_grad=True)
        shape = hf.Heightfield(heightfield=h, max_height=1.0)
        si = shape.ray_intersect(hf.Ray3f(r[0:3], r[3:6], r[6]), hf.RayFlags.All)
        out = getattr(si, field)
        (out[comp] if out.dim() == 2 else out).sum().backward()
        return float(h.grad.sum())
    _check_lift_answers(lift)
