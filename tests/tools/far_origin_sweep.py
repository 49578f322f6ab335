"""Validity domain of the fp32 hit test for DISTANT ray origins ("far origins").

The per-triangle Moeller-Trumbore test is evaluated in float32 (as the reference's Mesh does); its noise grows with
the distance between the ray origin and the triangle, so on needle terrain (white noise, cells much smaller than the
distance) a far ray can be reported by a triangle ~1 cell beside the ideal ray.  The traversal margin follows that
noise (setup_ray, hf_kernels.hip); this sweep measures what is left: for white-noise fields of N^2 texels and ray
origins `dist` object units away it counts rays where the GPU, the oracle's hierarchical walk and the oracle's band
brute force disagree, and resolves each disagreement with the FULL brute force (every triangle of the field).

usage (GPU box):  python tests/tools/far_origin_sweep.py [rays_per_case [dump_dir]]   (dump_dir: the disagreeing rays as .npz)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hf_amd                                 # noqa: E402  (tests/hf_amd.py: the package under its short name)
from oracle import hf_oracle as O             # noqa: E402


def main(n=400000, cap=64, dump=None):
    O.build()
    rng = np.random.default_rng(1)
    nt = min(16, os.cpu_count() or 1)
    print("N dist max_height | rays | gpu!=band walk!=band gpu!=walk | of the first %d: band==full gpu==full walk==full" % cap)
    for N, dist, mh in [(1024, 8, 1.0), (1024, 50, 1.0), (2048, 50, 0.2), (4096, 3, 0.5), (4096, 8, 0.5), (4096, 50, 0.5), (4096, 200, 0.5)]:
        h = rng.uniform(0, 1, (N, N)).astype(np.float32)
        f = O.OracleField(h, max_height=mh)
        shape = hf_amd.Heightfield(heightfield=torch.from_numpy(h).cuda(), max_height=mh)
        c = rng.uniform(-1, 1, (2, n)); dirs = rng.normal(size=(3, n))
        dirs[2] = -np.abs(dirs[2]) * rng.uniform(0.05, 1.0, n); dirs /= np.linalg.norm(dirs, axis=0)
        o = np.concatenate([c, np.full((1, n), mh * 0.5)]) - dirs * dist
        r = np.concatenate([o, dirs, np.full((1, n), np.inf)]).astype(np.float32)
        tb, _, _, pb = f.ray_intersect_preliminary(r, band=True, nthreads=nt)
        tw, _, _, pw = f.ray_intersect_preliminary(r, nthreads=nt)
        rt = torch.from_numpy(r).cuda()
        pi = shape.ray_intersect_preliminary(hf_amd.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous()))
        pg = pi.prim_index.cpu().numpy().view(np.uint32)
        gb, wb, gw = pg != pb, pw != pb, pg != pw
        bad = np.nonzero(gb | wb)[0][:cap]
        line = f"{N} {dist} {mh} | {n} | {int(gb.sum())} {int(wb.sum())} {int(gw.sum())}"
        if bad.size:
            tn, _, _, pn = f.ray_intersect_preliminary(np.ascontiguousarray(r[:, bad]), naive=True, nthreads=nt)
            line += f" | {bad.size}: {int((pn == pb[bad]).sum())} {int((pn == pg[bad]).sum())} {int((pn == pw[bad]).sum())}"
            if dump:   # the disagreeing rays with every checker's answer (regression vectors: tests/test_oracle_band.py)
                os.makedirs(dump, exist_ok=True)
                np.savez(os.path.join(dump, f"far_origin_N{N}_d{dist}.npz"), index=bad, rays=r[:, bad], full_prim=pn, full_t=tn,
                         band_prim=pb[bad], walk_prim=pw[bad], gpu_prim=pg[bad])
                lost = bad[(pn != pw[bad]) | (pn != pg[bad])]
                if lost.size:
                    line += f" | walk or GPU != full on rays {lost.tolist()}"
        print(line, flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 400000, dump=sys.argv[2] if len(sys.argv) > 2 else None)
