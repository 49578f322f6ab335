#!/usr/bin/env python3
"""Generates the committed golden fixtures with the CPU oracle (oracle/hf_oracle.c).

The reference snapshot holds no heightfield outputs to copy, so these
vectors pin the *oracle* (and through it the HIP path) against regressions; the oracle itself
is pinned against the reference's own known answers in tests/test_oracle_reference_answers.py.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hf_amd  # noqa: E402  (workload generator only; no GPU needed)
from oracle import hf_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sine64():
    """configs[0]: 64x64 procedural sine heightfield, 128x128 orthographic sensor @1spp"""
    h = hf_amd.workload.sine_heights(64, 64).numpy()
    rays = hf_amd.workload.ortho_rays(128, 128, 1, "cpu").numpy()
    f = O.OracleField(h, max_height=0.5)
    t, u, v, prim = f.ray_intersect_preliminary(rays)
    tn, un, vn, primn = f.ray_intersect_preliminary(rays, naive=True)
    assert np.array_equal(prim, primn) and np.array_equal(t, tn)
    si = f.compute_surface_interaction(rays, t, u, v, prim, O.RAY_ALL)
    rng = np.random.default_rng(12345)
    g = {nm: rng.normal(size=(c, rays.shape[1])).astype(np.float32) for nm, c in O.GRAD_FIELDS}
    gh = f.adjoint(rays, t, u, v, prim, g, O.RAY_ALL)
    gh_follow = f.adjoint(rays, t, u, v, prim, g, O.RAY_ALL | O.RAY_FOLLOWSHAPE)
    # closed-form upstream gradient of SURVEY 8d: dL/dt = 1, dL/dp = n
    hit = np.isfinite(t)
    gcf = {"t": hit.astype(np.float32)[None], "p": (si["n"] * hit).astype(np.float32)}
    gh_cf = f.adjoint(rays, t, u, v, prim, gcf, O.RAY_ALL)
    np.savez_compressed(os.path.join(OUT, "sine64_ortho128.npz"), heights=h, t=t, u=u, v=v, prim=prim,
                        n=si["n"], uv=si["uv"], p=si["p"], dp_du=si["dp_du"], dp_dv=si["dp_dv"],
                        grad_seed=np.int64(12345), grad_h=gh, grad_h_follow=gh_follow, grad_h_closed=gh_cf)
    print("sine64_ortho128: hits", int(hit.sum()), "of", hit.size)


def transformed():
    """translated / scaled / rotated grid, the three differentiation modes (test_rectangle.py:176-272 analog)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    rng = np.random.default_rng(2024)
    tw = common.affine(5)
    h = common.heights("sine", 33, 21, rng)
    f = O.OracleField(h, max_height=0.4, to_world=tw, flip_normals=True)
    r = common.to_world_rays(np.concatenate([common.random_rays(1500, rng, 0.4), common.inside_rays(500, rng, 0.4)], 1), tw)
    t, u, v, prim = f.ray_intersect_preliminary(r)
    out = dict(heights=h, to_world=tw, rays=r, t=t, u=u, v=v, prim=prim, bbox=f.bbox())
    g = {nm: rng.normal(size=(c, r.shape[1])).astype(np.float32) for nm, c in O.GRAD_FIELDS}
    for nm, c in O.GRAD_FIELDS:
        out["g_" + nm] = g[nm]
    for name, fl in (("default", 0), ("follow", O.RAY_FOLLOWSHAPE), ("detach", O.RAY_DETACHSHAPE)):
        flags = O.RAY_ALL | O.RAY_BOUNDARYTEST | fl
        si = f.compute_surface_interaction(r, t, u, v, prim, flags)
        gh, go, gd = f.adjoint(r, t, u, v, prim, g, flags, ray_grads=True)
        for k, val in si.items():
            out[f"{name}_{k}"] = val
        out[f"{name}_grad_h"] = gh; out[f"{name}_grad_o"] = go; out[f"{name}_grad_d"] = gd
    np.savez_compressed(os.path.join(OUT, "affine33x21_modes.npz"), **out)
    print("affine33x21_modes: hits", int(np.isfinite(t).sum()), "of", t.size)


if __name__ == "__main__":
    sine64()
    transformed()
