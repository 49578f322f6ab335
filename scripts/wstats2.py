#!/usr/bin/env python3
"""Tail statistics from an -DHF_WSTATS=2 -DHF_WSTATS_THR=n build: the share of the per-lane walk (visits, cell rounds,
hand-offs) that runs while fewer than n lanes of the batch are still unfinished."""
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
v = torch.where(trav, pi.prim_uv[1].double().reshape(-1, 64), z).max(1).values
w = trav.any(1)
t, u, v = t[w], u[w], v[w]
s = lambda x: float(x.sum())
print(f"batches {int(w.sum())}: visits {s(t % 4096) / len(t):.2f} (tail {s(torch.floor(t / 4096)) / len(t):.2f}), "
      f"cell rounds {s(u % 4096) / len(t):.2f} (tail {s(torch.floor(u / 4096)) / len(t):.2f}), "
      f"hand-offs {s(v % 4096) / len(t):.2f} (tail {s(torch.floor(v / 4096)) / len(t):.2f}); "
      f"batches with a tail {float((torch.floor(v / 4096) > 0).double().mean()):.3f}")
