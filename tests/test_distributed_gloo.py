"""N > 1 path on CPU: world_size-2 gloo.  ONE wavefront is cut into image tiles (hf_amd.workload.partition_tiles,
bench.py's strong-scaling partition, BASELINE configs[3]); every rank traces its tiles, accumulates a private
dL/dheight texture, one all-reduce sums them (hf_amd.allreduce_gradient).  No GPU here, so the per-rank compute is
the CPU oracle; the check is that tiles + all-reduce reproduce the single-process gradient of the whole wavefront.
The same partition through the HIP kernels is checked on one device by tests/test_partition.py (virtual ranks)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


FILM, SPP, TILE = 48, 2, 8


def _grad_for_rank(rank, world):
    """gradient texture of rank `rank` of `world` (world = 1: the whole wavefront)"""
    import hf_amd
    from oracle import hf_oracle as O
    h = hf_amd.workload.sine_heights(32, 32).numpy()
    pixels = hf_amd.workload.partition_tiles(FILM, FILM, world, tile=TILE)[rank]
    r = hf_amd.workload.ortho_rays(FILM, FILM, SPP, "cpu", pixels=pixels).numpy()
    f = O.OracleField(h, max_height=0.5)
    t, u, v, prim = f.ray_intersect_preliminary(r, nthreads=1)
    si = f.compute_surface_interaction(r, t, u, v, prim, nthreads=1)
    hit = np.isfinite(t)
    g = {"t": hit.astype(np.float32)[None], "p": (si["n"] * hit).astype(np.float32)}
    return f.adjoint(r, t, u, v, prim, g, nthreads=1)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hf_amd
    g = torch.from_numpy(_grad_for_rank(rank, world).copy())
    hf_amd.allreduce_gradient(g)
    if rank == 0:
        np.save(out, g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gradient_allreduce(tmp_path):
    sys.path.insert(0, ROOT)
    world, port = 2, 29500 + (os.getpid() % 1000)
    out = str(tmp_path / "g.npy")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    want = _grad_for_rank(0, 1).astype(np.float64)          # the un-partitioned wavefront
    got = np.load(out)
    assert np.abs(want).max() > 0
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    assert np.abs(_grad_for_rank(0, 2)).max() > 0 and np.abs(_grad_for_rank(1, 2)).max() > 0   # both ranks contribute


def test_allreduce_is_identity_without_process_group():
    sys.path.insert(0, ROOT)
    import hf_amd
    g = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    assert torch.equal(hf_amd.allreduce_gradient(g.clone()), g)
