#!/usr/bin/env python3
"""Cycle split of the coherent walk from an -DHF_TIMING build (u = whole walk, v = per-lane subtree part)."""
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
u = pi.prim_uv[0].reshape(-1, 64); v = pi.prim_uv[1].reshape(-1, 64)
w = trav.any(1)
uw = (u * trav).max(1).values[w]; vw = (v * trav).max(1).values[w]
print(f"batches with work {int(w.sum())} of {w.numel()}")
print(f"cycles per batch: walk mean {float(uw.mean()):.0f} (p50 {float(uw.median()):.0f}, max {float(uw.max()):.0f}); subtree part mean {float(vw.mean()):.0f} = {float(vw.sum()/uw.sum())*100:.1f}%")
print(f"sum walk cycles {float(uw.sum()):.3e}")
