#!/usr/bin/env python3
"""Lane-count histogram of the visits and cell rounds from an -DHF_WSTATS=4 build (at most 8 / 9..24 / more lanes)."""
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
def split(x): return [float((x % 1024).sum()) / nb, float((torch.floor(x / 1024) % 1024).sum()) / nb, float(torch.floor(x / 1048576).sum()) / nb]
c, v = split(t), split(u)
print(f"batches {nb}: cell rounds per batch with <=8 / 9..24 / >24 lanes: {c[0]:.2f} / {c[1]:.2f} / {c[2]:.2f}; visits: {v[0]:.2f} / {v[1]:.2f} / {v[2]:.2f}")
