"""Oracle surface interaction + adjoint vs float64 central finite differences (the reference's
methodology, src/integrators/tests/test_ad_integrators.py:1001-1012), all three modes, general
affine to_world, flipped normals, ray gradients.  CPU only."""
import numpy as np
import pytest

import common
import si_numpy as S


@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("mode", ["default", "follow", "detach"])
def test_si_and_adjoint_vs_fd(oracle, mode, flip):
    rng = np.random.default_rng(3)
    W, H, s = 9, 7, 0.6
    h = rng.uniform(0.2, 0.8, (H, W)).astype(np.float32)
    tw = common.affine(4)
    f = oracle.OracleField(h, max_height=s, to_world=tw, flip_normals=flip)
    n = 24
    r = common.to_world_rays(common.random_rays(n, rng, s), tw)
    t, u, v, prim = f.ray_intersect_preliminary(r)
    assert np.isfinite(t).sum() >= n // 2
    flags = S.RAY_ALL | {"default": 0, "follow": S.RAY_FOLLOWSHAPE, "detach": S.RAY_DETACHSHAPE}[mode]
    si = f.compute_surface_interaction(r, t, u, v, prim, flags)
    g = {nm: rng.normal(size=(c, n)).astype(np.float32) for nm, c in S.GRAD_FIELDS}
    gh, go, gd = f.adjoint(r, t, u, v, prim, g, flags, ray_grads=True)
    gh_fd = np.zeros((H, W))
    for k in np.where(np.isfinite(t))[0]:
        gk = {nm: (g[nm][:, k] if c > 1 else g[nm][0, k]) for nm, c in S.GRAD_FIELDS}
        ref = S.surface_interaction(h, s, tw, flip, r[0:3, k], r[3:6, k], prim[k], flags, (u[k], v[k]), h)
        for nm, _ in S.GRAD_FIELDS:   # forward values: 1e-5 relative
            assert np.allclose(si[nm][..., k], ref[nm], rtol=1e-5, atol=2e-6), (nm, k)
        for (i, j), val in S.fd_height_gradient(h, s, tw, flip, r[0:3, k], r[3:6, k], prim[k], flags, gk,
                                                (u[k], v[k])).items():
            gh_fd[i, j] += val
        fo, fdd = S.fd_ray_gradient(h, s, tw, flip, r[0:3, k], r[3:6, k], prim[k], flags, gk, (u[k], v[k]))
        assert np.allclose(go[:, k], fo, rtol=2e-4, atol=2e-4 * (1 + np.abs(fo).max()))
        assert np.allclose(gd[:, k], fdd, rtol=2e-4, atol=2e-4 * (1 + np.abs(fdd).max()))
    if mode == "detach":
        assert np.all(gh == 0)
    else:
        assert np.abs(gh - gh_fd).max() <= 1e-5 * np.abs(gh_fd).max()


def test_closed_form_dt_dh(oracle):
    """dt/dh_k = s * b_k * n_z / (n . d) in object space (to_world = I)."""
    rng = np.random.default_rng(8)
    h = rng.uniform(0.3, 0.6, (6, 6)).astype(np.float32)
    s = 0.8
    f = oracle.OracleField(h, max_height=s)
    r = common.random_rays(50, rng, s * 0.6)
    t, u, v, prim = f.ray_intersect_preliminary(r)
    si = f.compute_surface_interaction(r, t, u, v, prim)
    for k in np.where(np.isfinite(t))[0][:20]:
        g = {"t": np.zeros((1, 50), np.float32)}; g["t"][0, k] = 1
        gh = f.adjoint(r, t, u, v, prim, g)
        ids = S.prim_vertex_ids(6, int(prim[k]))
        b = [1 - u[k] - v[k], u[k], v[k]]
        nd = float(si["n"][:, k] @ r[3:6, k])
        for (i, j), bk in zip(ids, b):
            assert abs(gh[i, j] - s * bk * si["n"][2, k] / nd) <= 1e-4 * (1 + abs(gh[i, j]))
