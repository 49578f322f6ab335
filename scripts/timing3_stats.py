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
z = torch.zeros(1, device=dev)
u = torch.where(trav, pi.prim_uv[0].reshape(-1, 64), z); v = torch.where(trav, pi.prim_uv[1].reshape(-1, 64), z); t = torch.where(trav, pi.t.reshape(-1, 64), z)
w = trav.any(1)
A = u.max(1).values[w]; B = v.max(1).values[w]; T = t.max(1).values[w]
print(f"batches {int(w.sum())}: walk {float(T.mean()):.0f} cyc/batch; subtree phase A (walk) {float(A.mean()):.0f} ({float(A.sum()/T.sum())*100:.1f}%), phase B (cells) {float(B.mean()):.0f} ({float(B.sum()/T.sum())*100:.1f}%), packet+handoff rest {float((T-A-B).mean()):.0f} ({float((T-A-B).sum()/T.sum())*100:.1f}%)")
