#!/usr/bin/env python3
"""Silhouette gradients through the warped-area reparameterisation -- the primary-ray part of
a prb_reparam-style integrator (src/python/python/ad/integrators/prb_reparam.py:317-366: the camera ray is
reparameterised, the pixel value is multiplied by the determinant) on top of the heightfield shape:

    d', det = hf_amd.reparameterize_ray(shape, ray)              # identity in primal mode
    si      = shape.ray_intersect(Ray3f(o, d'), RayFlags.All)    # attached to the heights and to d'
    L       = sum over samples of  f(si) * det / spp             # f = a smooth function of the visible point

Scene: a flat field with a box ridge across it, seen obliquely, so that the ridge's top edge occludes the field
behind it.  theta lifts the ridge.  dL/dtheta has an interior part (the visible ridge top moves with theta) and a
SILHOUETTE part (pixels switch from the field to the ridge), which only the reparameterisation sees.  The script
prints the finite-difference derivative of the rendered sum, the attached-only gradient and the reparameterised one.

    python examples/silhouette_gradient.py [--film 192 --spp 64 --aux 16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hf_amd  # noqa: E402


def scene(n=65, device="cuda"):
    h = torch.full((n, n), 0.2, device=device)
    ridge = torch.zeros((n, n), dtype=torch.bool, device=device)
    ridge[n // 2 - 4:n // 2 + 4, n // 8: n - n // 8] = True        # a box ridge across the field, short of the border
    h[ridge] = 0.7
    return h, ridge


def camera(film, spp, device, seed=0):
    # looking along +y, 35 degrees above the horizon: the ridge hides a strip of the field behind it
    return hf_amd.workload.ortho_rays(film, film, spp, device, seed=seed, origin=(0.0, -2.0, 1.6),
                                      target=(0.0, 0.0, 0.2), scale=(1.2, 1.2, 1.0))


def f_of(si):
    """a smooth function of the visible point: its world height (0 where nothing is hit)"""
    return torch.where(si.is_valid(), si.p[2], torch.zeros_like(si.t))


def render_sum(h, rays, spp):
    shape = hf_amd.Heightfield(heightfield=h, max_height=0.5)
    ray = hf_amd.Ray3f(rays[0:3].contiguous(), rays[3:6].contiguous(), rays[6].contiguous())
    with torch.no_grad():
        return float(f_of(shape.ray_intersect(ray, hf_amd.RayFlags.All)).double().sum()) / spp


def gradients(h, ridge, rays, spp, aux=16, kappa=2e4, exponent=3.0, reparam=True, seed=0):
    """dL/dtheta for theta = 'lift every ridge texel' via reverse mode; returns (value, derivative)"""
    shape = hf_amd.Heightfield(heightfield=h.clone(), max_height=0.5)
    shape.heightfield.requires_grad_(True)
    ray = hf_amd.Ray3f(rays[0:3].contiguous(), rays[3:6].contiguous(), rays[6].contiguous())
    if reparam:
        d, det = hf_amd.reparameterize_ray(shape, ray, num_rays=aux, kappa=kappa, exponent=exponent, seed=seed)
    else:
        d, det = ray.d, torch.ones(len(ray), device=h.device)
    si = shape.ray_intersect(hf_amd.Ray3f(ray.o, d, ray.maxt), hf_amd.RayFlags.All)
    L = (f_of(si) * det).sum() / spp
    L.backward()
    return float(L.detach()), float(shape.heightfield.grad[ridge].double().sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--film", type=int, default=192)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--aux", type=int, default=16)
    ap.add_argument("--kappa", type=float, default=2e4)
    ap.add_argument("--eps", type=float, default=0.02)
    a = ap.parse_args()
    dev = torch.device("cuda")
    h, ridge = scene(device=dev)
    rays = camera(a.film, a.spp, dev)
    hp, hm = h.clone(), h.clone()
    hp[ridge] += a.eps; hm[ridge] -= a.eps
    fd = (render_sum(hp, rays, a.spp) - render_sum(hm, rays, a.spp)) / (2 * a.eps)
    _, g_att = gradients(h, ridge, rays, a.spp, reparam=False)
    _, g_rep = gradients(h, ridge, rays, a.spp, aux=a.aux, kappa=a.kappa, reparam=True)
    print(f"finite differences (eps {a.eps}):      dL/dtheta = {fd:.3f}")
    print(f"attached geometry only:               dL/dtheta = {g_att:.3f}   (misses the silhouette: {g_att / fd:.2f} of FD)")
    print(f"with reparameterize_ray ({a.aux} aux rays): dL/dtheta = {g_rep:.3f}   ({g_rep / fd:.2f} of FD)")
    return fd, g_att, g_rep


if __name__ == "__main__":
    main()
