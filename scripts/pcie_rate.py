#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path (GPU box): what the bench wavefront would cost a caller whose rays and records live in
HOST memory -- rays up (28 B per ray), the launch, the fused record down (104 B per ray) -- next to the resident figure the
bench reports; and the latency of the 16-ray host-pointer packet entry (hf_ray_intersect_preliminary_packet).
usage: python scripts/pcie_rate.py [rays, default 2^24]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, hf_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
dev = torch.device("cuda", 0)
shape = hf_amd.Heightfield(heightfield=hf_amd.workload.sine_heights(4096, 4096, device=dev), max_height=0.5)
rays_d = hf_amd.workload.ortho_rays(1024, 1024, 64, dev, start=(1 << 25), count=n)   # a slice from the middle of the bench wavefront
rays_h = torch.empty((7, n), dtype=torch.float32).pin_memory(); rays_h.copy_(rays_d)
out_h = torch.empty((26, n), dtype=torch.float32).pin_memory()                        # the fused record: 104 B per ray
def step():
    r = rays_h.to(dev, non_blocking=True)
    si = shape.ray_intersect(hf_amd.Ray3f(r[0:3], r[3:6], r[6]), hf_amd.RayFlags.All)
    rec = torch.cat([si.t[None], si.p, si.n, si.uv, si.dp_du, si.dp_dv, si.sh_frame.s, si.sh_frame.t, si.sh_frame.n, si.wi, si.boundary_test[None]][:10])[:26]
    out_h[:rec.shape[0]].copy_(rec, non_blocking=True)
    torch.cuda.synchronize()
for _ in range(2): step()
t0 = time.perf_counter()
for _ in range(5): step()
dt = (time.perf_counter() - t0) / 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
r = rays_h.to(dev); torch.cuda.synchronize()
e0.record()
for _ in range(5): shape.ray_intersect(hf_amd.Ray3f(r[0:3], r[3:6], r[6]), hf_amd.RayFlags.All)
e1.record(); torch.cuda.synchronize()
res = e0.elapsed_time(e1) / 5
up = time.perf_counter(); rays_h.to(dev); torch.cuda.synchronize(); up = time.perf_counter() - up
print(f"{n} rays: host -> device -> host {dt * 1e3:.1f} ms = {n / dt / 1e6:.0f} Mrays/s (rays up at {28 * n / up / 1e9:.1f} GB/s); resident launch {res:.3f} ms = {n / res / 1e3:.0f} Mrays/s")
o = np.random.default_rng(0).uniform(-1, 1, (3, 16)).astype(np.float32); o[2] = 2.0
d = np.tile(np.array([[0.01], [0.02], [-1.0]], np.float32), (1, 16))
for _ in range(20): shape.ray_intersect_preliminary_packet(o, d)
t0 = time.perf_counter()
for _ in range(200): shape.ray_intersect_preliminary_packet(o, d)
print(f"16-ray host packet: {(time.perf_counter() - t0) / 200 * 1e6:.0f} us per call")
