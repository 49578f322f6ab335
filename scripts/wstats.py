#!/usr/bin/env python3
"""Wave-level dynamic block-execution counts per batch from an -DHF_WSTATS build."""
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
u = torch.where(trav, pi.prim_uv[0].double().reshape(-1, 64), z).max(1).values
v = torch.where(trav, pi.prim_uv[1].double().reshape(-1, 64), z).max(1).values
t = torch.where(trav, pi.t.double().reshape(-1, 64), z).max(1).values
pr = torch.where(trav, pi.prim_index.to(torch.int64).reshape(-1, 64) & 0xFFFFFFFF, torch.zeros(1, device=dev, dtype=torch.int64)).max(1).values
w = trav.any(1)
u, v, t, pr = u[w], v[w], t[w], pr[w]
def m(x): return float(x.mean())
print(f"batches {int(w.sum())}")
print(f"row sweep: rows with a lane inside {m(t % 1024):.1f}, nodes box-tested {m(torch.floor(t / 1024) % 1024):.1f}, hand-offs with work {m(torch.floor(t / 1048576)):.1f}")
print(f"subtree (max over lanes of per-call sums; lanes run the same wave-level loop): iterations {m(u % 4096):.1f}, hand-offs after the hoisted visit (HF_HOIST builds) {m(torch.floor(u / 4096)):.1f}, visits {m(v % 4096):.1f}, cell rounds {m(torch.floor(v / 4096)):.1f}")
print(f"lanes per visit {float((pr & 0xFFFF).double().mean()) / max(m(v % 4096), 1e-9):.1f}, lanes per cell round {float((pr >> 16).double().mean()) / max(m(torch.floor(v / 4096)), 1e-9):.1f}")
