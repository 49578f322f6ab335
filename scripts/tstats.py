#!/usr/bin/env python3
"""Per-batch cycle breakdown from an -DHF_TSTATS build (diagnostic): total, setup, per-lane subtree walk."""
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
def mx(x): return torch.where(trav, x.double().reshape(-1, 64), z).max(1).values
tot, sub, setup = mx(pi.t), mx(pi.prim_uv[0]), mx(pi.prim_uv[1])
w = trav.any(1)
print(f"batches {int(w.sum())}: cycles/batch total {float(tot[w].mean()):.0f}, setup+coherence {float(setup[w].mean()):.0f}, "
      f"subtree walks {float(sub[w].mean()):.0f}, shared walk {float((tot - sub - setup)[w].mean()):.0f}")
tt = tot[w]
q = torch.quantile(tt[torch.randperm(tt.numel(), device=tt.device)[:1000000]], torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9, 0.99], dtype=torch.float64, device=tt.device))
print("total cycles/batch quantiles 10/25/50/75/90/99 %:", [int(x) for x in q.tolist()])
srt = torch.sort(tt, descending=True).values
cs = torch.cumsum(srt, 0) / srt.sum()
for frac in (0.01, 0.05, 0.1, 0.25, 0.5):
    k = int(frac * srt.numel())
    print(f"  the most expensive {100 * frac:.0f} % of the batches take {100 * float(cs[k]):.1f} % of the cycles")
hits = trav.sum(1)[w].double()
for lo, hi in ((1, 16), (16, 48), (48, 64), (64, 65)):
    m = (hits >= lo) & (hits < hi)
    if m.any():
        print(f"  batches with {lo}..{hi - 1} hit lanes: {100 * float(m.double().mean()):.1f} % of batches, mean cycles {float(tt[m].mean()):.0f}")
