#!/usr/bin/env python3
"""Tail statistics from an -DHF_WSTATS=5 build: at the points between two converged rounds where at most 8 lanes still
walk, how many such points a batch has, how many lanes walk, and how many pending level-1 / level-2 siblings they hold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hf_amd
from hf_amd import _capi, build
build.LIB_PATH = os.environ["HF_LIB"]; _capi._build.LIB_PATH = os.environ["HF_LIB"]
grid, film, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
shape = hf_amd.Heightfield(heightfield=hf_amd.workload.sine_heights(grid, grid, device=dev), max_height=0.5)
rays = hf_amd.workload.ortho_rays(film, film, spp, dev)
pi = shape.ray_intersect_preliminary(hf_amd.Ray3f(rays[0:3], rays[3:6], rays[6]))
trav = (pi.t != float("inf")).reshape(-1, 64)
z = torch.zeros(1, device=dev, dtype=torch.float64)
t = torch.where(trav, pi.t.double().reshape(-1, 64), z).max(1).values
u = torch.where(trav, pi.prim_uv[0].double().reshape(-1, 64), z).max(1).values
w = trav.any(1)
t, u = t[w], u[w]
nb = len(t)
pts, walkers = float((t % 4096).sum()) / nb, float(torch.floor(t / 4096).sum()) / nb
b1, b2 = float((u % 4096).sum()) / nb, float(torch.floor(u / 4096).sum()) / nb
print(f"batches {nb}: tail points per batch {pts:.2f}, walkers per point {walkers / max(pts, 1e-9):.2f}, pending level-1 siblings per walker "
      f"{b1 / max(walkers, 1e-9):.2f}, pending level-2 siblings per walker {b2 / max(walkers, 1e-9):.2f}")
