#!/usr/bin/env python3
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
u = pi.prim_uv[0].reshape(-1, 64); v = pi.prim_uv[1].reshape(-1, 64); n = torch.where(trav, pi.t.reshape(-1, 64), torch.zeros(1, device=dev))
w = trav.any(1)
uw = (u * trav).max(1).values[w]; vw = (v * trav).max(1).values[w]; nw = n.max(1).values[w]
print(f"packet expansions per batch {float(nw.mean()):.1f}; load-wait cycles/expansion {float(uw.sum()/nw.sum()):.0f}; mask cycles/expansion {float(vw.sum()/nw.sum()):.0f}")
