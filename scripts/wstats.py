#!/usr/bin/env python3
"""Wave-level execution statistics per traversing batch from an -DHF_WSTATS=<mode> build (HF_LIB selects it; the
diagnostic build exports its counters in place of the hit record -- see WSTATS_EXPORT in csrc/hf_kernels.hip).

usage: HF_LIB=scratch_so/libhf_X_ws.so scripts/wstats.py [--mode M] grid film spp
  mode 1 (-DHF_WSTATS / =1): passes, box tests, hand-offs, iterations, visits, cell rounds, lanes per visit / cell round
  mode 2 (-DHF_WSTATS=2 -DHF_WSTATS_THR=n): share of the walk that runs while fewer than n lanes of the batch are unfinished
  mode 3: participants per hand-off, repeaters, distribution of hand-offs per batch
  mode 4: lane-count histogram of visits and cell rounds (<= 8 / 9..24 / more)
  mode 5: pending level-1 / level-2 siblings held by the last (<= 8) walkers
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hf_amd
from hf_amd import _capi, build

ap = argparse.ArgumentParser()
ap.add_argument("--mode", type=int, default=1)
ap.add_argument("grid", type=int); ap.add_argument("film", type=int); ap.add_argument("spp", type=int)
a = ap.parse_args()
build.LIB_PATH = os.environ["HF_LIB"]; _capi._build.LIB_PATH = os.environ["HF_LIB"]
dev = torch.device("cuda", 0)
shape = hf_amd.Heightfield(heightfield=hf_amd.workload.sine_heights(a.grid, a.grid, device=dev), max_height=0.5)
rays = hf_amd.workload.ortho_rays(a.film, a.film, a.spp, dev)
pi = shape.ray_intersect_preliminary(hf_amd.Ray3f(rays[0:3], rays[3:6], rays[6]))
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
    print(f"batches {nb}")
    print(f"beam sweep: passes {m(t % 1024):.1f}, nodes box-tested per lane {m(fl(t / 1024) % 1024):.1f}, nodes with a taker {m(fl(t / 1048576)):.1f}")
    print(f"per-lane walk: iterations {m(u % 4096):.1f}, hand-offs after the hoisted visit {m(fl(u / 4096)):.1f}, visits {m(v % 4096):.1f}, cell rounds {m(fl(v / 4096)):.1f}")
    print(f"lanes per visit {float((pr & 0xFFFF).double().mean()) / max(m(v % 4096), 1e-9):.1f}, lanes per cell round {float((pr >> 16).double().mean()) / max(m(fl(v / 4096)), 1e-9):.1f}")
elif a.mode == 2:
    print(f"batches {nb}: visits {s(t % 4096) / nb:.2f} (tail {s(fl(t / 4096)) / nb:.2f}), cell rounds {s(u % 4096) / nb:.2f} (tail {s(fl(u / 4096)) / nb:.2f}), "
          f"hand-offs {s(v % 4096) / nb:.2f} (tail {s(fl(v / 4096)) / nb:.2f}); batches with a tail {float((fl(v / 4096) > 0).double().mean()):.3f}")
elif a.mode == 3:
    vis, part, cel, rep = t % 4096, fl(t / 4096), u % 4096, fl(u / 4096)
    print(f"batches {nb}: hand-offs {s(v) / nb:.2f}, participants per hand-off {s(part) / s(v):.1f}, of which repeaters {s(rep) / s(part):.3f}; "
          f"visits per hand-off {s(vis) / s(v):.2f}, cell rounds per hand-off {s(cel) / s(v):.2f}")
    for k in range(0, 9):
        sel = v == k if k < 8 else v >= 8
        if int(sel.sum()):
            print(f"  {k}{'+' if k == 8 else ''} hand-offs: {float(sel.double().mean()):.3f} of the batches, visits {m(vis[sel]):.1f}, cell rounds {m(cel[sel]):.1f}, "
                  f"participants {m(part[sel]):.1f}, repeaters {m(rep[sel]):.1f}")
elif a.mode == 4:
    def split(x): return [s(x % 1024) / nb, s(fl(x / 1024) % 1024) / nb, s(fl(x / 1048576)) / nb]
    c, vv = split(t), split(u)
    print(f"batches {nb}: cell rounds per batch with <=8 / 9..24 / >24 lanes: {c[0]:.2f} / {c[1]:.2f} / {c[2]:.2f}; visits: {vv[0]:.2f} / {vv[1]:.2f} / {vv[2]:.2f}")
elif a.mode == 5:
    pts, walkers = s(t % 4096) / nb, s(fl(t / 4096)) / nb
    b1, b2 = s(u % 4096) / nb, s(fl(u / 4096)) / nb
    print(f"batches {nb}: tail points per batch {pts:.2f}, walkers per point {walkers / max(pts, 1e-9):.2f}, pending level-1 siblings per walker "
          f"{b1 / max(walkers, 1e-9):.2f}, pending level-2 siblings per walker {b2 / max(walkers, 1e-9):.2f}")
