#!/usr/bin/env python3
"""Hand-off statistics from an -DHF_WSTATS=3 build: participants per hand-off, the share of them that took part in an
earlier hand-off of the same batch, and the distribution of hand-offs per batch."""
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
nb = len(t)
vis, part = t % 4096, torch.floor(t / 4096)
cel, rep = u % 4096, torch.floor(u / 4096)
print(f"batches {nb}: hand-offs {float(v.sum()) / nb:.2f}, participants per hand-off {float(part.sum()) / float(v.sum()):.1f}, "
      f"of which repeaters {float(rep.sum()) / float(part.sum()):.3f}; visits per hand-off {float(vis.sum()) / float(v.sum()):.2f}, cell rounds per hand-off {float(cel.sum()) / float(v.sum()):.2f}")
for k in range(0, 9):
    sel = v == k if k < 8 else v >= 8
    if int(sel.sum()):
        print(f"  {k}{'+' if k == 8 else ''} hand-offs: {float(sel.double().mean()):.3f} of the batches, visits {float(vis[sel].mean()):.1f}, cell rounds {float(cel[sel].mean()):.1f}, participants {float(part[sel].mean()):.1f}, repeaters {float(rep[sel].mean()):.1f}")
