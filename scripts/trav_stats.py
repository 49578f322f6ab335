#!/usr/bin/env python3
"""Traversal statistics of the bench workload from an HF_STATS build (scripts/stats_build.sh)."""
import ctypes as C, os, sys, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hf_amd
from hf_amd import _capi, build
lib_stats = os.environ.get("HF_STATS_LIB", "gpurun_out/libhf_stats.so")
build.LIB_PATH = lib_stats
_capi._build.LIB_PATH = lib_stats
grid, film, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
shape = hf_amd.Heightfield(heightfield=hf_amd.workload.sine_heights(grid, grid, device=dev), max_height=0.5)
rays = hf_amd.workload.ortho_rays(film, film, spp, dev)
ray = hf_amd.Ray3f(rays[0:3], rays[3:6], rays[6])
pi = shape.ray_intersect_preliminary(ray)
trav = pi.t != float("inf")
u, v = pi.prim_uv[0][trav], pi.prim_uv[1][trav]
hit = pi.t[trav] >= 0
nexp, nleafp = u % 1000, torch.floor(u / 1000)
ncell, niter = v % 1000, torch.floor(v / 1000)
print(f"rays {trav.numel()}  traversing (inside bbox) {float(trav.float().mean()):.3f}  hit {float((pi.t >= 0).float().mean() ):.3f}")
for name, x in (("expansions", nexp), ("leaf parents", nleafp), ("cells tested", ncell), ("inner iters", niter)):
    for lab, sel in (("all", slice(None)), ("hit", hit), ("nohit", ~hit)):
        y = x[sel]
        if y.numel():
            q = torch.quantile(y[:: max(1, y.numel() // 1000000)].float(), torch.tensor([0.5, 0.95, 0.99], device=dev))
            print(f"  {name:13s} {lab:6s} mean {float(y.mean()):7.2f} p50 {float(q[0]):6.1f} p95 {float(q[1]):6.1f} p99 {float(q[2]):6.1f} max {float(y.max()):6.0f}")
# intra-wave divergence: waves of 64 consecutive rays
full_n = torch.zeros_like(pi.t); full_n[trav] = niter
w = full_n.reshape(-1, 64)
wmax, wmean = w.max(1).values, w.mean(1)
sel = wmax > 0
print(f"waves with work {float(sel.float().mean()):.3f}; sum(wave max)/sum(lane iters) = {float(wmax[sel].sum() * 64 / w[sel].sum()):.2f}")
full_e = torch.zeros_like(pi.t); full_e[trav] = nexp
w = full_e.reshape(-1, 64); wmax = w.max(1).values
print(f"expansions: sum(wave max)*64/sum = {float(wmax[sel].sum() * 64 / w[sel].sum()):.2f}")
act = trav.reshape(-1, 64).float().mean(1)
print(f"active lane fraction in waves with work {float(act[sel].mean()):.3f}")
