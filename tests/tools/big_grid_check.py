#!/usr/bin/env python3
"""Largest grids (tool): N x N heights up to the 2^30-vertex limit, 200 k mixed rays against the oracle.
usage: big_grid_check.py N"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hf_amd, common
from oracle import hf_oracle as O
N = int(sys.argv[1])
t0 = time.time()
u = torch.arange(N, device="cuda", dtype=torch.float32) / (N - 1)
h = (0.5 + 0.25 * torch.sin(2 * np.pi * 37 * u)[None, :] * torch.cos(2 * np.pi * 29 * u)[:, None]
     + 0.125 * torch.sin(2 * np.pi * 301 * (u[None, :] + u[:, None])))
shape = hf_amd.Heightfield(heightfield=h, max_height=0.5)
print(f"N={N}: levels {shape.num_levels()}, GPU build {time.time() - t0:.1f} s", flush=True)
hc = h.cpu().numpy(); del h
t0 = time.time()
f = O.OracleField(hc, max_height=0.5)
print(f"oracle build {time.time() - t0:.1f} s", flush=True)
rng = np.random.default_rng(0)
r = np.concatenate([common.random_rays(100000, rng), common.inside_rays(50000, rng)], 1)
# rays into the far corner (largest indices)
c = common.random_rays(50000, rng); c[0:2] = np.abs(c[0:2]); c[3:5] = np.abs(c[3:5])
r = np.concatenate([r, c], 1).astype(np.float32)
rt = torch.from_numpy(r).cuda()
ray = hf_amd.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous())
pi = shape.ray_intersect_preliminary(ray)
t, uu, vv, prim = f.ray_intersect_preliminary(r, nthreads=16)
pg, tg = pi.prim_index.cpu().numpy().view(np.uint32), pi.t.cpu().numpy()
bad = np.nonzero((prim != pg) | (t != tg))[0]
print(f"{r.shape[1]} rays, hit fraction {np.isfinite(t).mean():.2f}, max prim {int(prim[np.isfinite(t)].max())}, mismatches {bad.size}")
# gradient of a few rays through the adjoint
shape.heightfield.requires_grad_(True)
si = shape.ray_intersect(ray, hf_amd.RayFlags.All)
(torch.where(si.is_valid(), si.t, torch.zeros_like(si.t)).sum()).backward()
hit = np.isfinite(t)
gh = f.adjoint(r, t, uu, vv, prim, {"t": hit.astype(np.float32)[None]}, O.RAY_ALL, nthreads=16)
g = shape.heightfield.grad
nz = np.nonzero(gh)
got = g[torch.from_numpy(nz[0]).cuda(), torch.from_numpy(nz[1]).cuda()].cpu().numpy()
print("gradient rel err on the touched texels", float(np.linalg.norm(got - gh[nz]) / np.linalg.norm(gh[nz])), "untouched sum", float(g.abs().sum()) - float(np.abs(got).sum()))
sys.exit(1 if bad.size else 0)
