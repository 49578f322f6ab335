"""float64 numpy restatement of the surface-interaction maths (one ray at a time).

Third, independent statement of src/render/mesh.cpp:672-903 used to check the
C oracle's and the HIP kernels' *gradients* by central finite differences in
float64 (the reference's own methodology: src/integrators/tests/
test_ad_integrators.py:1001-1012).  Pure test helper.
"""
import numpy as np

RAY_UV, RAY_DPDUV, RAY_SHADINGFRAME = 0x2, 0x4, 0x8
RAY_BOUNDARYTEST, RAY_FOLLOWSHAPE, RAY_DETACHSHAPE = 0x40, 0x80, 0x100
RAY_ALL = RAY_UV | RAY_DPDUV | RAY_SHADINGFRAME


def prim_vertex_ids(W, prim):
    cell, tri = prim >> 1, prim & 1
    cx, cy = cell % (W - 1), cell // (W - 1)
    if tri == 0:
        return [(cy, cx), (cy, cx + 1), (cy + 1, cx)]
    return [(cy + 1, cx + 1), (cy + 1, cx), (cy, cx + 1)]


def world_vertices(heights, max_height, to_world, ids):
    H, W = heights.shape
    A = np.asarray(to_world, np.float64).reshape(3, 4)
    P, UV = [], []
    for (i, j) in ids:
        q = np.array([j * 2.0 / (W - 1) - 1.0, i * 2.0 / (H - 1) - 1.0, heights[i, j] * max_height])
        P.append(A[:, :3] @ q + A[:, 3])
        UV.append(np.array([j / (W - 1.0), i / (H - 1.0)]))
    return P, UV


def moeller_trumbore(o, d, p0, p1, p2):
    e1, e2 = p1 - p0, p2 - p0
    pvec = np.cross(d, e2)
    inv_det = 1.0 / np.dot(e1, pvec)
    tvec = o - p0
    u = np.dot(tvec, pvec) * inv_det
    qvec = np.cross(tvec, e1)
    v = np.dot(d, qvec) * inv_det
    t = np.dot(e2, qvec) * inv_det
    return t, u, v


def surface_interaction(heights, max_height, to_world, flip_normals, o, d, prim, flags,
                        uv_fixed=None, heights_geom=None):
    """Differentiable SI of one ray in float64.

    default     : (t,u,v) re-derived by Moeller-Trumbore from the *current* heights/ray
    FollowShape : barycentrics frozen to `uv_fixed`, p glued to the triangle
    DetachShape : geometry evaluated on `heights_geom` (frozen copy); ray still live
    """
    heights = np.asarray(heights, np.float64)
    o = np.asarray(o, np.float64); d = np.asarray(d, np.float64)
    H, W = heights.shape
    ids = prim_vertex_ids(W, int(prim))
    hg = heights_geom if (flags & RAY_DETACHSHAPE) else heights
    P, UV = world_vertices(np.asarray(hg, np.float64), max_height, to_world, ids)
    if flags & RAY_FOLLOWSHAPE:
        b1, b2 = uv_fixed
        t = None
    else:
        t, b1, b2 = moeller_trumbore(o, d, *P)
    b0 = 1.0 - b1 - b2
    p = P[0] * b0 + P[1] * b1 + P[2] * b2
    if flags & RAY_FOLLOWSHAPE:
        t = np.sqrt(np.dot(p - o, p - o) / np.dot(d, d))
    dp0, dp1 = P[1] - P[0], P[2] - P[0]
    N = np.cross(dp0, dp1)
    n = N / np.linalg.norm(N)
    out = {"t": t, "p": p}
    if flags & (RAY_UV | RAY_DPDUV):
        uv = UV[0] * b0 + UV[1] * b1 + UV[2] * b2
    else:
        uv = np.array([b1, b2])
    out["uv"] = uv
    if flags & RAY_DPDUV:
        duv0, duv1 = UV[1] - UV[0], UV[2] - UV[0]
        det = duv0[0] * duv1[1] - duv0[1] * duv1[0]
        out["dp_du"] = (duv1[1] * dp0 - duv0[1] * dp1) / det
        out["dp_dv"] = (-duv1[0] * dp0 + duv0[0] * dp1) / det
    else:
        out["dp_du"] = np.zeros(3); out["dp_dv"] = np.zeros(3)  # coordinate_system(n): not checked
    sgn = -1.0 if flip_normals else 1.0
    out["n"] = sgn * n
    out["sh_n"] = sgn * n
    return out


GRAD_FIELDS = [("t", 1), ("p", 3), ("n", 3), ("uv", 2), ("sh_n", 3), ("dp_du", 3), ("dp_dv", 3)]


def loss(si, g):
    """scalar L = sum_f <g_f, si_f> for upstream gradient dict g."""
    L = 0.0
    for name, _ in GRAD_FIELDS:
        if name in g:
            L += float(np.sum(np.asarray(g[name], np.float64) * si[name]))
    return L


def fd_height_gradient(heights, max_height, to_world, flip_normals, o, d, prim, flags, g,
                       uv_fixed=None, eps=1e-4):
    """central finite differences of loss w.r.t. the 3 vertex heights of `prim`."""
    heights = np.asarray(heights, np.float64)
    H, W = heights.shape
    ids = prim_vertex_ids(W, int(prim))
    grad = {}
    for (i, j) in ids:
        hp = heights.copy(); hp[i, j] += eps
        hm = heights.copy(); hm[i, j] -= eps
        Lp = loss(surface_interaction(hp, max_height, to_world, flip_normals, o, d, prim, flags,
                                      uv_fixed, heights), g)
        Lm = loss(surface_interaction(hm, max_height, to_world, flip_normals, o, d, prim, flags,
                                      uv_fixed, heights), g)
        grad[(i, j)] = (Lp - Lm) / (2 * eps)
    return grad


def fd_ray_gradient(heights, max_height, to_world, flip_normals, o, d, prim, flags, g,
                    uv_fixed=None, eps=1e-5):
    o = np.asarray(o, np.float64); d = np.asarray(d, np.float64)
    go, gd = np.zeros(3), np.zeros(3)
    for k in range(3):
        for arr, out in ((o, go), (d, gd)):
            a = arr.copy(); a[k] += eps
            b = arr.copy(); b[k] -= eps
            args_p = (a, d) if arr is o else (o, a)
            args_m = (b, d) if arr is o else (o, b)
            Lp = loss(surface_interaction(heights, max_height, to_world, flip_normals, *args_p,
                                          prim, flags, uv_fixed, heights), g)
            Lm = loss(surface_interaction(heights, max_height, to_world, flip_normals, *args_m,
                                          prim, flags, uv_fixed, heights), g)
            out[k] = (Lp - Lm) / (2 * eps)
    return go, gd
