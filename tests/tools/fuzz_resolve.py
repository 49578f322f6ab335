#!/usr/bin/env python3
"""Re-examine one scene of a fuzz run (tests/tools/fuzz_parity.py): the oracle's hierarchical walk and its band brute
force on the scene's rays, every ray on which they differ resolved by the FULL brute force over all cells; with a GPU,
the HIP kernels as a third party.  usage: FUZZ_MAXDIM=... fuzz_resolve.py seed scene rays [cap]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np, torch
from oracle import hf_oracle as O
import fuzz_parity

seed0, sc, nrays = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 64
O.build()
nt = min(16, os.cpu_count() or 1)
h, mh, tw, kind, r, _ = fuzz_parity.make_scene(seed0, sc, nrays)
print(f"scene {sc} of seed {seed0}: {h.shape[1]}x{h.shape[0]} {kind} mh={mh:.3g} tw={'y' if tw is not None else 'n'}, {r.shape[1]} rays")
f = O.OracleField(h, max_height=mh, to_world=tw)
tb, _, _, pb = f.ray_intersect_preliminary(r, band=True, nthreads=nt)
tw_, _, _, pw = f.ray_intersect_preliminary(r, nthreads=nt)
pg = None
if torch.cuda.is_available():
    import hf_amd
    props = dict(heightfield=torch.from_numpy(h), max_height=mh)
    if tw is not None:
        props["to_world"] = torch.from_numpy(tw)
    rt = torch.from_numpy(r).cuda()
    pi = hf_amd.Heightfield(props).ray_intersect_preliminary(hf_amd.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous()))
    pg, tg = pi.prim_index.cpu().numpy().view(np.uint32), pi.t.cpu().numpy()
dis = (pw != pb) | (tw_ != tb)
if pg is not None:
    dis |= (pg != pb) | (pg != pw)
bad = np.nonzero(dis)[0][:cap]
print(f"walk != band on {int(((pw != pb) | (tw_ != tb)).sum())} rays" + (f", gpu != band {int((pg != pb).sum())}, gpu != walk {int((pg != pw).sum())}" if pg is not None else ""))
if bad.size:
    tn, _, _, pn = f.ray_intersect_preliminary(np.ascontiguousarray(r[:, bad]), naive=True, nthreads=nt)
    print(f"resolved by the full brute force ({bad.size} rays): band right {int((pn == pb[bad]).sum())}, walk right {int((pn == pw[bad]).sum())}"
          + (f", gpu right {int((pn == pg[bad]).sum())}" if pg is not None else ""))
    for k, i in enumerate(bad[:8]):
        print(f"  ray {i}: origin distance {float(np.linalg.norm(r[0:3, i])):.1f}, full {pn[k]}, band {pb[i]}, walk {pw[i]}" + (f", gpu {pg[i]}" if pg is not None else ""))
