#!/usr/bin/env python3
"""Wave-level execution statistics per traversing batch from an -DHF_WSTATS=<mode> build (HF_LIB selects it; the
diagnostic build exports its counters in place of the hit record -- see WSTATS_EXPORT in csrc/hf_kernels.hip).

usage: HF_LIB=scratch_so/libhf_X_ws.so scripts/wstats.py [--mode M] grid film spp
  mode 1 (-DHF_WSTATS / =1): passes, box tests, hand-offs, iterations, visits, cell rounds, lanes per visit / cell round
  mode 4: lane-count histogram of visits and cell rounds (<= 8 / 9..24 / more)
  (modes 2, 3, 5 belonged to the per-lane walk below the hand-off level, which the item walk replaced in round 4)
  --aux KAPPA: the statistics of auxiliary rays (sample 0 of hf_reparam_aux_rays) instead of the primary rays
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hf_amd
from hf_amd import _capi, build

ap = argparse.ArgumentParser()
ap.add_argument("--mode", type=int, default=1)
ap.add_argument("--aux", type=float, default=0.0, help="kappa: trace auxiliary ray 0 of every ray (hf_reparam_aux_rays) instead of the ray itself")
ap.add_argument("--bounce", action="store_true", help="the statistics of one cosine bounce per primary hit (SURVEY 8d's secondary rays; -DHF_WSTATS_ROOT builds export the per-lane root walk's counters)")
ap.add_argument("--gen-bounce", default="", help="(internal) write the bounce rays of the workload to this file with the library as built in-tree, and exit")
ap.add_argument("grid", type=int); ap.add_argument("film", type=int); ap.add_argument("spp", type=int)
a = ap.parse_args()
if a.bounce and not a.gen_bounce:
    # the bounce rays come from the primary hits of the PRODUCTION library (a diagnostic build's hit records are counters)
    import subprocess, tempfile
    rays_file = os.path.join(tempfile.gettempdir(), "wstats_bounce_rays.pt")
    env = {k: v for k, v in os.environ.items() if k != "HF_LIB"}
    subprocess.run([sys.executable, os.path.abspath(__file__), "--gen-bounce", rays_file, str(a.grid), str(a.film), str(a.spp)], env=env, check=True)
if not a.gen_bounce:
    build.LIB_PATH = os.environ["HF_LIB"]; _capi._build.LIB_PATH = os.environ["HF_LIB"]
dev = torch.device("cuda", 0)
shape = hf_amd.Heightfield(heightfield=hf_amd.workload.sine_heights(a.grid, a.grid, device=dev), max_height=0.5)
rays = hf_amd.workload.ortho_rays(a.film, a.film, a.spp, dev)
if a.aux > 0:
    import ctypes as C
    ad = torch.empty((3, rays.shape[1]), device=dev); mt = torch.empty(rays.shape[1], device=dev)
    p3 = lambda x: (C.c_void_p * 3)(x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr())
    _capi.check(_capi.lib().hf_reparam_aux_rays(rays.shape[1], C.byref(p3(rays[0:3])), C.byref(p3(rays[3:6])), None, 0, a.aux, 0, 0, None,
                                                C.byref(p3(ad)), mt.data_ptr(), None))
    rays = torch.cat([rays[0:3], ad, mt[None]])
if a.bounce and not a.gen_bounce:
    rays = torch.load(rays_file, weights_only=True).to(dev)
    os.remove(rays_file)
if a.gen_bounce:
    si0 = shape.ray_intersect(hf_amd.Ray3f(rays[0:3].contiguous(), rays[3:6].contiguous(), rays[6].contiguous()), hf_amd.RayFlags.All)
    hit_idx = torch.nonzero(torch.isfinite(si0.t)).squeeze(1)
    hit_idx = hit_idx[: (hit_idx.numel() // 64) * 64]
    rays, _ = hf_amd.workload.secondary_rays(si0.p[:, hit_idx], si0.n[:, hit_idx], seed=0)
    torch.save(rays.cpu(), a.gen_bounce)
    sys.exit(0)
pi = shape.ray_intersect_preliminary(hf_amd.Ray3f(rays[0:3].contiguous(), rays[3:6].contiguous(), rays[6].contiguous()))
trav = (pi.t != float("inf")).reshape(-1, 64)
z = torch.zeros(1, device=dev, dtype=torch.float64)
def mx(x): return torch.where(trav, x.double().reshape(-1, 64), z).max(1).values
w = trav.any(1)
t, u, v = mx(pi.t)[w], mx(pi.prim_uv[0])[w], mx(pi.prim_uv[1])[w]
pr = torch.where(trav, pi.prim_index.to(torch.int64).reshape(-1, 64) & 0xFFFFFFFF, torch.zeros(1, device=dev, dtype=torch.int64)).max(1).values[w]
nb = len(t)
m = lambda x: float(x.mean())
s = lambda x: float(x.sum())
fl = torch.floor
if a.mode == 1:
    # (a batch exports its counters only when it took the beam sweep: prim_index then holds lane counts, not a primitive)
    # (a real hit has barycentrics in [0, 1]; the exported iteration count is at least 1)
    tk = u >= 1.0
    frac = float(tk.double().mean())
    print(f"traversing batches {nb}, of which with at least one round of the work list {frac:.3f}; per traversing batch (the others count as zero):")
    t, u, v, pr = t[tk], u[tk], v[tk], pr[tk]
    _m = m
    m = lambda x: frac * _m(x)
    print(f"beam sweep: passes {m(t % 1024):.1f}, nodes box-tested per lane {m(fl(t / 1024) % 1024):.1f}, nodes with a taker {m(fl(t / 1048576)):.1f}")
    print(f"per-lane walk: iterations {m(u % 4096):.1f}, hand-offs after the hoisted visit {m(fl(u / 4096)):.1f}, visits {m(v % 4096):.1f}, cell rounds {m(fl(v / 4096)):.1f}")
    print(f"lanes per visit {float((pr & 0xFFFF).double().mean()) / max(_m(v % 4096), 1e-9):.1f}, lanes per cell round {float((pr >> 16).double().mean()) / max(_m(fl(v / 4096)), 1e-9):.1f}")
elif a.mode == 4:
    def split(x): return [s(x % 1024) / nb, s(fl(x / 1024) % 1024) / nb, s(fl(x / 1048576)) / nb]
    c, vv = split(t), split(u)
    print(f"batches {nb}: cell rounds per batch with <=8 / 9..24 / >24 lanes: {c[0]:.2f} / {c[1]:.2f} / {c[2]:.2f}; visits: {vv[0]:.2f} / {vv[1]:.2f} / {vv[2]:.2f}")
