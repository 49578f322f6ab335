"""Shared input generators for the parity tests (seeded, deterministic)."""
import numpy as np


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    R = np.eye(3)
    i, j = [(1, 2), (0, 2), (0, 1)][axis]
    R[i, i] = c; R[j, j] = c; R[i, j] = -s; R[j, i] = s
    return R


def affine(seed=0):
    """a generic rotation * anisotropic scale + translation, row-major 3x4 float32"""
    rng = np.random.default_rng(seed)
    A = rot(0, 0.3) @ rot(2, 0.7) @ rot(1, -0.2) @ np.diag(rng.uniform(0.6, 1.6, 3))
    T = rng.uniform(-0.5, 0.5, 3)
    return np.concatenate([A, T[:, None]], 1).astype(np.float32)


def heights(kind, W, H, rng):
    if kind == "flat":
        return np.full((H, W), 0.5, np.float32)
    if kind == "rand":
        return rng.uniform(0, 1, (H, W)).astype(np.float32)
    if kind == "stairs":
        return (np.repeat((np.arange(H) // max(1, H // 5))[:, None], W, 1) / 8.0).astype(np.float32)
    if kind == "sine":
        u = np.arange(W) / (W - 1.0); v = np.arange(H)[:, None] / (H - 1.0)
        return (0.5 + 0.25 * np.sin(2 * np.pi * 2 * u) * np.cos(2 * np.pi * 2 * v)
                + 0.125 * np.sin(2 * np.pi * 7 * (u + v))).astype(np.float32)
    raise ValueError(kind)


def to_world_rays(r_obj, to_world):
    """map object-space rays [7,n] through a 3x4 affine (float64 maths, rounded once)"""
    if to_world is None:
        return r_obj.astype(np.float32)
    A = np.asarray(to_world, np.float64).reshape(3, 4)
    o = A[:, :3] @ r_obj[0:3].astype(np.float64) + A[:, 3:4]
    d = A[:, :3] @ r_obj[3:6].astype(np.float64)
    return np.concatenate([o, d, r_obj[6:7]]).astype(np.float32)


def random_rays(n, rng, zmax=0.5):
    """rays from around the unit box aimed into it (object space)"""
    o = rng.uniform(-1.5, 1.5, (3, n)); o[2] = rng.uniform(-0.5, 1.5, n)
    tgt = rng.uniform(-1, 1, (3, n)); tgt[2] = rng.uniform(0, zmax, n)
    d = tgt - o
    d /= np.linalg.norm(d, axis=0, keepdims=True)
    d *= rng.uniform(0.5, 2.0, n)           # non-unit directions are legal
    return np.concatenate([o, d, np.full((1, n), np.inf)]).astype(np.float32)


def inside_rays(n, rng, zmax=0.5):
    """secondary-ray like: origins inside the bound, random directions, finite maxt"""
    o = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(0, zmax, n)])
    d = rng.normal(size=(3, n)); d /= np.linalg.norm(d, axis=0)
    return np.concatenate([o, d, rng.uniform(0.01, 3, (1, n))]).astype(np.float32)


def structured_rays(xs, ys, zlo=-0.1, zhi=0.6):
    """degenerate families: vertical rays through vertices / cell centres / grid lines,
    axis-parallel horizontal rays along grid lines, diagonal rays through vertices"""
    gx = np.concatenate([xs, (xs[:-1] + xs[1:]) / 2]); gy = np.concatenate([ys, (ys[:-1] + ys[1:]) / 2])
    out = []
    X, Y = np.meshgrid(gx, gy); n = X.size
    one, zero, inf = np.ones(n), np.zeros(n), np.full(n, np.inf)
    out.append(np.stack([X.ravel(), Y.ravel(), np.full(n, 2.0), zero, zero, -one, inf]))
    out.append(np.stack([X.ravel(), Y.ravel(), np.full(n, -2.0), zero, zero, one, inf]))
    zs = np.linspace(zlo, zhi, 15)
    Y2, Z2 = np.meshgrid(gy, zs); n = Y2.size
    one, zero, inf = np.ones(n), np.zeros(n), np.full(n, np.inf)
    out.append(np.stack([np.full(n, -2.0), Y2.ravel(), Z2.ravel(), one, zero, zero, inf]))
    out.append(np.stack([np.full(n, 2.0), Y2.ravel(), Z2.ravel(), -one, zero, zero, inf]))
    X2, Z3 = np.meshgrid(gx, zs); n = X2.size
    one, zero, inf = np.ones(n), np.zeros(n), np.full(n, np.inf)
    out.append(np.stack([X2.ravel(), np.full(n, 2.0), Z3.ravel(), zero, -one, zero, inf]))
    out.append(np.stack([X2.ravel(), np.full(n, -2.0), Z3.ravel(), zero, one, zero, inf]))
    for dxy in [(1, 1), (1, -1), (-1, 1), (-1, -1)]:
        X, Y = np.meshgrid(gx, gy); n = X.size
        d = np.array([dxy[0], dxy[1], -0.3])
        o = np.stack([X.ravel(), Y.ravel(), np.full(n, 0.25)]) - d[:, None] * 3
        out.append(np.concatenate([o, np.repeat(d[:, None], n, 1), np.full((1, n), np.inf)]))
    out = np.concatenate(out, 1).astype(np.float32)
    # the same families with negative zeros in the direction (d = -up is (-0,-0,-1)): 1/(-0) = -inf must not
    # flip the slab tests of the walk
    neg = out.copy()
    z = neg[3:6] == 0
    neg[3:6][z] = np.float32(-0.0)
    return np.concatenate([out, neg], 1)
