#!/usr/bin/env python3
"""Per-batch cycle breakdown of the traversal kernel from an -DHF_TSTATS build (HF_LIB selects it): shader cycles per
phase of a traversing batch of the bench wavefront, stamped by lane 0 of every wave (TSTAMP in csrc/hf_kernels.hip).
usage: HF_LIB=scratch_so/libhf_X_ts.so scripts/tstats.py grid film spp [fused]   (fused: the fused launch, RayFlags.All)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hf_amd
from hf_amd import _capi, build
build.LIB_PATH = os.environ["HF_LIB"]; _capi._build.LIB_PATH = os.environ["HF_LIB"]
grid, film, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
shape = hf_amd.Heightfield(heightfield=hf_amd.workload.sine_heights(grid, grid, device=dev), max_height=0.5)
rays = hf_amd.workload.ortho_rays(film, film, spp, dev)
ray = hf_amd.Ray3f(rays[0:3], rays[3:6], rays[6])
pi = shape.ray_intersect(ray, hf_amd.RayFlags.All) if len(sys.argv) > 4 and sys.argv[4] == "fused" else shape.ray_intersect_preliminary(ray)
t = pi.t.reshape(-1, 64).double()
lane = (pi.prim_index.reshape(-1, 64).to(torch.int64) & 0xFFFFFFFF)
trav = (torch.isfinite(t[:, 0]) & (lane[:, 1] == 1) & (lane[:, 63] == 63))   # batches that exported (any lane alive)
c = t[trav][:, :16]
names = ["set-up, clip, coherence", "beam set-up", "passes (node loads, beam test)", "candidates (box, record, push)", "visit rounds", "cell rounds",
         "-", "rest of the traversal + fold", "output of the previous batch (stores, SI)"]
c[:, 8] = torch.where(c[:, 8] > 1e6, torch.full_like(c[:, 8], float("nan")), c[:, 8])   # a wave's first batch: no previous one
c[:, 8] = torch.nan_to_num(c[:, 8], nan=float(torch.nanmean(c[:, 8])))
tot = float(c[:, :9].sum(1).mean())
print(f"batches {int(trav.sum())}: {tot:.0f} cycles per batch (stamped; the stamps cost ~10 %)")
for k, nm in enumerate(names):
    if nm != "-":
        print(f"  {nm:45s} {float(c[:, k].mean()):8.0f}  {100 * float(c[:, k].mean()) / tot:5.1f} %")

# the wide path (fetches of 256 rays that miss as a whole): lane 0 of such a fetch leaves three cycle counts in prim_uv[0] of its
# first rays -- waiting for the grab number, for the rays (incl. the conservative clip), issuing the 30 + 4 stores
u = (pi.prim_uv[0] if hasattr(pi, "prim_uv") else None)
if u is not None:
    u4 = u.reshape(-1, 256)[:, :4].double()
    wide = u4[:, 3] == 1.0
    if bool(wide.any()):
        w3 = u4[wide]
        print(f"wide fetches {int(wide.sum())} of {u4.shape[0]}: cycles waiting for the grab number {float(w3[:, 0].mean()):.0f} (median {float(w3[:, 0].median()):.0f}), "
              f"rays + clip {float(w3[:, 1].mean()):.0f} (median {float(w3[:, 1].median()):.0f}), stores issued {float(w3[:, 2].mean()):.0f} (median {float(w3[:, 2].median()):.0f})")
