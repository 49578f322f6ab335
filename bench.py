#!/usr/bin/env python3
"""Headline benchmark: Mrays/s, forward + adjoint, on an N x N heightfield.

One step = one pass of the hot path over one ray wavefront:
    forward  : hf_ray_intersect  (hierarchical traversal + fused surface interaction, RayFlags::All)
    adjoint  : hf_adjoint        (reverse mode of the SI, atomic scatter of dL/dheight)
               [+ one RCCL all-reduce of the N x N gradient texture when world_size > 1]
Workload (BASELINE.json configs[3] / SURVEY.md 8d): 4096^2 procedural sine heightfield,
1024x1024 orthographic sensor @ 64 spp = ONE wavefront of 67 108 864 rays.  At N > 1 the wavefront is cut into
32x32-pixel image tiles, tile b -> rank b % N (hf_amd.workload.partition_tiles; src/render/integrator.cpp:
130-140), i.e. STRONG scaling: total work fixed, `value` = 67.1 M rays / step time.  Heights and acceleration
data are replicated; every rank accumulates a private gradient texture and the textures are summed with one
all-reduce per step, issued asynchronously so that it overlaps the next step's forward + adjoint (double-
buffered texture).  `--scaling weak` keeps the round-1 mode (every rank traces its own 64 spp of the image).

Launch:  python bench.py [--gpus N --steps K --warmup W]
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# contract figures, SURVEY.md section 8(d)
FWD_BYTES_PER_RAY = 104.0   # 28 ray in + 72 SI out + 4 prim_index out
ADJ_BYTES_PER_RAY = 128.0   # 28 ray + 16 pi + 72 upstream in + 12 atomically added
GRID_BYTES_PER_TEXEL = 20.0 / 3.0  # heights 4 B + mips ~8/3 B, read once per pass
HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0       # same guide: measured float4 copy (79 % of spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=4096, help="heightfield resolution N (N x N)")
    ap.add_argument("--film", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time (0 = skip)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = image tiles of ONE wavefront (configs[3]); weak = a full wavefront per rank")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import hf_amd
    from hf_amd import _capi
    from hf_amd.shape import _DIFF_ROWS, _fill, _rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    # HF_BENCH_BACKEND=gloo rehearses the multi-rank path on fewer GPUs than ranks (ranks share devices; the
    # all-reduce then goes through the host).  The measured configuration is nccl = RCCL, one rank per GPU.
    backend = os.environ.get("HF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    collective = None
    if world > 1:
        # what the collective saw, as the process group itself reports it: every rank's (host, device index, device name,
        # PCI bus id) gathered on rank 0 -- under nccl (= RCCL) the ranks must hold DISTINCT devices
        import socket
        props = torch.cuda.get_device_properties(local)
        mine = {"rank": rank, "host": socket.gethostname(), "device": local, "name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", ""))}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        if backend == "nccl":
            ids = [(g["host"], g["device"]) for g in gathered]
            assert len(set(ids)) == world, f"ranks share a device under nccl: {ids}"
        try:
            nccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            nccl_version = None
        collective = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": nccl_version,
                      "ranks": gathered, "hip": torch.version.hip, "torch": torch.__version__}

    N, Wf, spp = args.grid, args.film, args.spp
    R_total = Wf * Wf * spp
    strong = world > 1 and args.scaling == "strong"
    lib = _capi.lib()

    # ---- inputs, resident in HBM before any timed region --------------------------------
    heights = hf_amd.workload.sine_heights(N, N, device=dev)
    shape = hf_amd.Heightfield(heightfield=heights, max_height=0.5)
    if strong:   # this rank's image tiles of the one wavefront
        my_pixels = hf_amd.workload.partition_tiles(Wf, Wf, world)[rank]
        rays = hf_amd.workload.ortho_rays(Wf, Wf, spp, dev, pixels=my_pixels)   # [7, R]
    else:
        rays = hf_amd.workload.ortho_rays(Wf, Wf, spp, dev, seed=rank)          # [7, R]
    R = rays.shape[1]
    t = torch.empty(R, dtype=torch.float32, device=dev)
    uv = torch.empty((2, R), dtype=torch.float32, device=dev)
    prim = torch.empty(R, dtype=torch.int32, device=dev)
    si = torch.empty((18, R), dtype=torch.float32, device=dev)           # t,p,n,uv,sh_n,dp_du,dp_dv = 72 B/ray
    gsi = torch.zeros((18, R), dtype=torch.float32, device=dev)          # upstream dL/dsi, 72 B/ray
    grads = [torch.zeros((N, N), dtype=torch.float32, device=dev) for _ in range(2 if world > 1 else 1)]
    pending = [None, None]                                               # outstanding all-reduce per buffer
    stream = torch.cuda.current_stream(dev).cuda_stream

    r_s = shape._rays_struct(rays[0:3], rays[3:6], rays[6])
    pi_s = shape._pi_struct(t, uv, prim)
    si_s = _fill(_capi.hf_si_t(), _DIFF_ROWS, _rows(si, R))
    g_s = _fill(_capi.hf_si_grad_t(), _DIFF_ROWS, _rows(gsi, R))
    flags = int(hf_amd.RayFlags.All)
    step_no = [0]

    def forward():
        _capi.check(lib.hf_ray_intersect(shape._h, R, C.byref(r_s), flags, None, C.byref(pi_s), C.byref(si_s), stream))

    band = torch.tensor([N, 0], dtype=torch.int32, device=dev)  # rows that receive gradient (hf_adjoint_rows)
    rows = [0, N]                                                # the rows the all-reduce covers: set after the warm-up
    serial = [False]                                             # True: wait for the all-reduce inside the step

    def adjoint(events=None):
        b = step_no[0] % len(grads)
        step_no[0] += 1
        if pending[b] is not None:        # the all-reduce that used this buffer two steps ago
            pending[b].wait(); pending[b] = None
        grad_h = grads[b]
        grad_h.zero_()
        if events:                        # the kernel alone: not the wait for an earlier all-reduce, not the memset
            events[0].record()
        _capi.check(lib.hf_adjoint_rows(shape._h, R, C.byref(r_s), C.byref(pi_s), flags, None, C.byref(g_s),
                                        grad_h.data_ptr(), None, None, band.data_ptr(), stream))
        if events:
            events[1].record()
        if world > 1:
            # one collective per step over the rows any rank touched (contiguous in memory); overlapped with the next
            # step's kernels -- or waited for at once (`serial`: what a step needs whose optimiser consumes the
            # gradient before the next forward)
            pending[b] = dist.all_reduce(grad_h[rows[0]:rows[1]], async_op=True)
            if serial[0]:
                pending[b].wait(); pending[b] = None
        return grad_h

    def drain():
        for b in range(len(pending)):
            if pending[b] is not None:
                pending[b].wait(); pending[b] = None

    # closed-form upstream gradient of SURVEY 8d: dL/dt = 1, dL/dp = n, rest 0
    forward()
    torch.cuda.synchronize()
    hit = torch.isfinite(si[0])
    gsi[0] = hit.to(torch.float32)
    gsi[1:4] = si[4:7] * hit
    hit_frac = float(hit.float().mean())
    del hit

    ev = lambda: torch.cuda.Event(enable_timing=True)
    fwd_ev, adj_ev = [], []

    def step(record):
        if record:
            a, b, c, d = ev(), ev(), ev(), ev()
            a.record(); forward(); b.record(); adjoint((c, d))
            fwd_ev.append((a, b)); adj_ev.append((c, d))
        else:
            forward(); adjoint()

    for _ in range(max(args.warmup, 1)):
        step(False)
    drain()
    torch.cuda.synchronize()
    # the rows ANY rank touched (static scene and camera: the band of the warm-up steps holds for the timed ones;
    # a caller whose heights move re-measures it): union over ranks
    lo_hi = band.to(torch.int64).clone()
    if world > 1:
        lo_t, hi_t = lo_hi[0:1].clone(), lo_hi[1:2].clone()
        dist.all_reduce(lo_t, op=dist.ReduceOp.MIN); dist.all_reduce(hi_t, op=dist.ReduceOp.MAX)
        lo_hi = torch.cat([lo_t, hi_t])
    rows[0], rows[1] = int(lo_hi[0]), max(int(lo_hi[1]), int(lo_hi[0]))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    drain()                               # the last all-reduce belongs to the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # the band kept accumulating during the timed steps: a launch that touched a row outside [rows[0], rows[1]) would
    # have had that row left out of the all-reduce -- check instead of assuming (static scene: must hold)
    chk = band.to(torch.int64).clone()
    if world > 1:
        lo_c, hi_c = chk[0:1].clone(), chk[1:2].clone()
        dist.all_reduce(lo_c, op=dist.ReduceOp.MIN); dist.all_reduce(hi_c, op=dist.ReduceOp.MAX)
        chk = torch.cat([lo_c, hi_c])
    assert int(chk[0]) >= rows[0] and int(chk[1]) <= rows[1], f"gradient rows {chk.tolist()} outside the all-reduced band {rows}"
    ms_step = 1e3 * elapsed / args.steps
    grad_h = grads[(step_no[0] - 1) % len(grads)]
    grad_sample = grad_h.clone()          # the reduced gradient of the last timed step (checksums below)
    # N > 1: the same steps with the all-reduce waited for inside the step (no overlap with the next forward) --
    # reported beside the headline, which overlaps it
    serial_ms = None
    if world > 1:
        serial[0] = True
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(False)
        drain()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        st = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
        serial_ms = 1e3 * float(st.item()) / args.steps
        serial[0] = False
    if world > 1:   # whole-job ray count: the one wavefront (strong) or one wavefront per rank (weak)
        rr = torch.tensor([float(R)], dtype=torch.float64, device=dev)
        dist.all_reduce(rr)
        total_rays = float(rr.item())
        assert not strong or int(total_rays) == R_total
    else:
        total_rays = float(R)
    value = total_rays / (elapsed / args.steps) / 1e6  # Mrays/s, whole job
    fwd_ms = sum(a.elapsed_time(b) for a, b in fwd_ev) / len(fwd_ev)
    adj_ms = sum(a.elapsed_time(b) for a, b in adj_ev) / len(adj_ev)

    # gradient sanity (size-independent property): grad is finite and non-zero
    gnorm = float(torch.linalg.norm(grad_sample.double()))
    assert math.isfinite(gnorm) and gnorm > 0
    gsum = float(grad_sample.double()[::7, ::5].sum())   # a second, position-dependent checksum of the reduced gradient

    out = None
    if rank == 0:
        # roofline of the dominant kernel.  ALGORITHMIC bytes (SURVEY 8d contract figures):
        #   forward, every ray: 28 B ray in + 72 B SI + 4 B prim_index out, + the grid once per pass
        #   adjoint: a ray that hit reads ray 28 + pi 16 + upstream 72 and adds 12 B; a ray that missed is
        #            recognised from pi.t (4 B) and skipped -- counting 128 B for it would credit bytes never moved
        n_hit = hit_frac * R
        fwd_bytes = R * FWD_BYTES_PER_RAY + N * N * GRID_BYTES_PER_TEXEL
        adj_bytes = n_hit * ADJ_BYTES_PER_RAY + (R - n_hit) * 4.0 + N * N * 4.0
        dom_ms, dom_bytes, dom_name = (fwd_ms, fwd_bytes, "hf_trace_kernel<2>") if fwd_ms >= adj_ms else \
                                      (adj_ms, adj_bytes, "hf_adjoint_kernel")
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        # counter-measured HBM traffic of that kernel, only when the profile it came from was taken at this commit
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1:   # the profile is of the N = 1 launch (the whole wavefront)
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(dom_name, {}).get("hbm_bytes_per_launch")
                traffic_src = tj.get("source")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "frac_of_measured_copy_peak": round(achieved / HBM_COPY_GBS, 5),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": round(dom_ms, 4),
                    "fwd_ms": round(fwd_ms, 4), "adj_ms": round(adj_ms, 4),
                    "fwd_frac": round(fwd_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "adj_frac": round(adj_bytes / (adj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "algorithmic_bytes": {"forward": fwd_bytes, "adjoint": adj_bytes}}
        extras = None
        if world == 1 and os.environ.get("HF_BENCH_EXTRAS", "1") != "0":  # profile_round.sh switches them off
            try:   # the headline line must not depend on the extra timings
                extras = other_launches(torch, hf_amd, _capi, lib, shape, rays, r_s, pi_s, si_s, si, R, stream, flags)
            except Exception as e:
                extras = {"error": repr(e)}
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            try:
                cpu = cpu_baseline(args, heights.cpu().numpy(), rays, gsi, R)
            except Exception as e:   # e.g. the oracle library could not be built on this host
                cpu = {"error": repr(e)}
        out = {"metric": "Mrays/s forward+adjoint on 4096^2 heightfield", "value": round(value, 2),
               "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_step, 4), "higher_is_better": True,
               "scaling": "strong" if (strong or world == 1) else "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{N}x{N} sine heightfield, {Wf}x{Wf} orthographic sensor @{spp}spp "
                                      f"= {int(total_rays)} rays per step, forward(ray_intersect, RayFlags.All)"
                                      f"+adjoint(dL/dheight)",
                          "rays_rank0": R, "rays_per_step": int(total_rays), "hit_fraction_rank0": round(hit_frac, 4),
                          "parallelism": (f"one wavefront in 32x32-pixel tiles, tile b -> rank b % {world}; "
                                          if strong else f"one wavefront per rank ({world}); ") +
                                         "heights replicated, 1 async all-reduce of the gradient texture per step"},
               "allreduce": {"rows": [rows[0], rows[1]], "bytes": (rows[1] - rows[0]) * N * 4,
                             "overlap": "with the next step's kernels (headline); serial_* = the all-reduce waited for inside "
                                        "the step: the number an optimiser that consumes the gradient before the next forward sees",
                             "serial_ms_per_step": None if serial_ms is None else round(serial_ms, 4),
                             "serial_value": None if serial_ms is None else round(total_rays / (serial_ms * 1e-3) / 1e6, 2)},
               "collective": collective,
               "library": library_id(_capi),
               "grad_l2": gnorm, "grad_checksum": gsum,
               "roofline": roofline, "cpu_baseline": cpu, "other_launches_ms": extras}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def library_id(_capi):
    """which libhf.so the numbers come from (HF_LIB can point the package at an A/B build: the line says so)"""
    import hashlib
    path = os.environ.get("HF_LIB") or _capi._build.LIB_PATH
    return {"path": os.path.relpath(path, ROOT) if path.startswith(ROOT) else path,
            "sha256_16": hashlib.sha256(open(path, "rb").read()).hexdigest()[:16],
            "hf_lib_override": bool(os.environ.get("HF_LIB"))}


def _rows_of(buf, n):
    return [buf.data_ptr() + 4 * n * k for k in range(buf.shape[0])]


def other_launches(torch, hf_amd, _capi, lib, shape, rays, r_s, pi_s, si_s, si, R, stream, flags, iters=10):
    """Not part of `value`: the other entry points of the path on the same wavefront, and on the incoherent
    secondary rays of SURVEY 8d (one cosine bounce + one shadow ray per primary hit), HIP-event ms per launch."""
    dev = si.device
    hit8 = torch.empty(R, dtype=torch.uint8, device=dev)
    hit_idx = torch.nonzero(torch.isfinite(si[0])).squeeze(1)
    bounce, shadow = hf_amd.workload.secondary_rays(si[1:4][:, hit_idx], si[4:7][:, hit_idx], seed=0)
    Rs = bounce.shape[1]
    b_s = shape._rays_struct(bounce[0:3], bounce[3:6], bounce[6])
    s_s = shape._rays_struct(shadow[0:3], shadow[3:6], shadow[6])
    fn = {
        "ray_intersect_preliminary": lambda: lib.hf_ray_intersect_preliminary(shape._h, R, C.byref(r_s), None, C.byref(pi_s), stream),
        "ray_test": lambda: lib.hf_ray_test(shape._h, R, C.byref(r_s), None, hit8.data_ptr(), stream),
        "compute_surface_interaction": lambda: lib.hf_compute_surface_interaction(shape._h, R, C.byref(r_s), C.byref(pi_s), flags, None, C.byref(si_s), stream),
        "parameters_changed(mip rebuild)": lambda: (shape.parameters_changed(["heightfield"]), 0)[1],
    }
    out = {}
    for name, f in fn.items():
        _capi.check(f()); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            _capi.check(f())
        e1.record(); torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / iters, 4)
    # next-row kernels: direct lighting of the wavefront under 4 directional lights, and its adjoint
    K = 4
    L = (_capi.hf_dir_light_t * K)()
    for k, (x, y, z) in enumerate([(0.5, 0.2, 0.84), (-0.5, 0.3, 0.81), (0.1, -0.6, 0.79), (0.0, 0.0, 1.0)]):
        nrm = math.sqrt(x * x + y * y + z * z)
        L[k].to_light[0], L[k].to_light[1], L[k].to_light[2], L[k].irradiance = x / nrm, y / nrm, z / nrm, math.pi
    spp = 64 if R % 64 == 0 else 1
    img = torch.empty((K, R // spp), dtype=torch.float32, device=dev)
    gimg = torch.ones_like(img)
    gn = torch.empty((3, R), dtype=torch.float32, device=dev)
    rows = _rows_of(si, R)
    shn = (C.c_void_p * 3)(*rows[9:12]); gnp = (C.c_void_p * 3)(*_rows_of(gn, R))
    dd = (C.c_void_p * 3)(r_s.d[0], r_s.d[1], r_s.d[2])
    more = {
        "direct_lighting(4 lights)": lambda: lib.hf_direct_lighting(R, spp, C.byref(shn), C.byref(dd), rows[0], K, L, 1.0, None, img.data_ptr(), stream),
        "direct_lighting_adjoint": lambda: lib.hf_direct_lighting_adjoint(R, spp, C.byref(shn), C.byref(dd), rows[0], K, L, 1.0, None, gimg.data_ptr(), C.byref(gnp), stream),
    }
    for name, f in more.items():
        _capi.check(f()); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            _capi.check(f())
        e1.record(); torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / iters, 4)
    # backward of mitsuba.ad.reparameterize_ray on the wavefront (reparam.py's defaults: 4 auxiliary
    # rays per ray, kappa 1e5, exponent 3): 4 x hf_reparam_trace + hf_reparam_backward through the host mirror
    try:
        ray = hf_amd.Ray3f(rays[0:3], rays[3:6])
        gdir = torch.randn(3, R, device=dev); gdv = torch.randn(R, device=dev)
        was = shape.heightfield.requires_grad
        shape.heightfield.requires_grad_(True)

        def rp():
            shape.heightfield.grad = None
            dd_, det = hf_amd.reparameterize_ray(shape, ray, num_rays=4, kappa=1e5, exponent=3.0)
            torch.autograd.backward((dd_, det), (gdir, gdv))   # (the upstream gradients handed over as they are: no loss kernels of the caller in the timing)
        rp(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            rp()
        e1.record(); torch.cuda.synchronize()
        out["reparameterize_ray_backward(4 aux rays)"] = round(e0.elapsed_time(e1) / 2, 3)
        # the reference's default for prb_reparam is 16 auxiliary rays per ray (prb_reparam.py:237): 39 GB of auxiliary hits
        def rp16():
            shape.heightfield.grad = None
            dd_, det = hf_amd.reparameterize_ray(shape, ray, num_rays=16, kappa=1e5, exponent=3.0)
            torch.autograd.backward((dd_, det), (gdir, gdv))   # (the upstream gradients handed over as they are: no loss kernels of the caller in the timing)
        rp16(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rp16(); e1.record(); torch.cuda.synchronize()
        out["reparameterize_ray_backward(16 aux rays)"] = round(e0.elapsed_time(e1), 3)
        torch.cuda.empty_cache()
        shape.heightfield.grad = None
        shape.heightfield.requires_grad_(was)
        del ray, gdir, gdv
    except Exception as e:
        out["reparameterize_ray_backward(4 aux rays)"] = {"error": repr(e)}
    # secondary rays last: they overwrite pi / si of the primary wavefront
    sec = {
        "bounce_rays_ray_intersect": lambda: lib.hf_ray_intersect(shape._h, Rs, C.byref(b_s), flags, None, C.byref(pi_s), C.byref(si_s), stream),
        # ... with the hint an integrator passes for its secondary rays (coherent = false, scene.h:117-146 -> hf_set_ray_coherence)
        "bounce_rays_ray_intersect(coherent=false)": lambda: (lib.hf_set_ray_coherence(shape._h, 1),
                                                              lib.hf_ray_intersect(shape._h, Rs, C.byref(b_s), flags, None, C.byref(pi_s), C.byref(si_s), stream),
                                                              lib.hf_set_ray_coherence(shape._h, 0))[1],
        "shadow_rays_ray_test": lambda: lib.hf_ray_test(shape._h, Rs, C.byref(s_s), None, hit8.data_ptr(), stream),
    }
    for name, f in sec.items():
        _capi.check(f()); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            _capi.check(f())
        e1.record(); torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / iters, 4)
    out["secondary_rays"] = int(Rs)
    # BASELINE configs[4] stand-in (examples/inverse_heights.py): 100 Adam steps, loss = multi-light renders only,
    # end-to-end wall-clock on this GPU (trace -> shade -> loss -> adjoints -> hf_adam_step incl. rebuild)
    try:
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import inverse_heights
        inverse_heights.run(grid=64, film=64, spp=1, steps=3, verbose=False)            # code objects, allocator
        # (a) recovery at the size where 100 shading-only Adam steps can recover heights (the per-texel fit of a
        #     gradient-domain loss does not propagate low frequencies on finer grids within 100 steps)
        hist, err, wall = inverse_heights.run(grid=64, film=128, spp=4, steps=100, lr=0.04, verbose=False)
        rec = {"grid": 64, "rays_per_step": 128 * 128 * 4, "adam_steps": 100, "wall_clock_s": round(wall, 4),
               "loss_first": round(hist[0], 6), "loss_last": round(hist[-1], 6),
               "centred_height_error_first": round(inverse_heights.run.start_centred_error, 5),
               "centred_height_error_last": round(inverse_heights.run.last_centred_error, 5)}
        # (b) the same loop at configs[1]'s size (timing only): the eager autograd loop, and ONE step captured into a HIP
        #     graph and replayed 100 times (inverse_heights.run_captured: no host work per step; hf_adam_step_scheduled)
        hist, err, wall = inverse_heights.run(grid=1024, film=512, spp=16, steps=100, lr=0.005, verbose=False)
        hist_c, err_c, tm = inverse_heights.run_captured(grid=1024, film=512, spp=16, steps=100, lr=0.005)
        out["configs4_inverse_loop"] = {"recovery": rec,
                                        "timing": {"grid": 1024, "rays_per_step": 512 * 512 * 16, "adam_steps": 100,
                                                   "wall_clock_s": round(tm["wall_clock_s"], 4),
                                                   "ms_per_step": round(tm["wall_ms_per_step"], 4),
                                                   "kernel_ms_sum_per_step": round(tm["gpu_ms_per_step"], 4),
                                                   "host_overhead_ms_per_step": round(max(0.0, tm["wall_ms_per_step"] - tm["gpu_ms_per_step"]), 4),
                                                   "how": "one optimisation step captured into a HIP graph, replayed 100 times",
                                                   "loss_first": round(hist_c[0], 6), "loss_last": round(hist_c[-1], 6),
                                                   "eager_autograd_loop": {"wall_clock_s": round(wall, 4), "ms_per_step": round(10.0 * wall, 4),
                                                                           "host_overhead_ms_per_step": round(max(0.0, 10.0 * wall - tm["gpu_ms_per_step"]), 4),
                                                                           "loss_first": round(hist[0], 6), "loss_last": round(hist[-1], 6)}}}
    except Exception as e:   # the headline number must not depend on the example
        out["configs4_inverse_loop"] = {"error": repr(e)}
    return out


def cpu_baseline(args, heights_np, rays, gsi, R):
    """CPU restatement of the reference's scalar path (oracle; the reference llvm_ad_rgb
    heightfield cannot be built -- no source, no Dr.Jit) on a bounded strided sample."""
    import numpy as np
    from oracle import hf_oracle as O
    # the GPU box grants a CPU share of 16 threads per GPU; never oversubscribe it
    cores = min(os.cpu_count() or 1, int(os.environ.get("HF_CPU_THREADS", "16")))
    f = O.OracleField(heights_np, max_height=0.5)
    grid_flags = O.RAY_ALL

    def run(k):
        stride = max(1, R // k)
        idx = slice(0, stride * k, stride)
        r = rays[:, idx].cpu().numpy()
        g = gsi[:, idx].cpu().numpy()
        grads = {"t": g[0:1], "p": g[1:4]}
        t0 = time.perf_counter()
        t, u, v, prim = f.ray_intersect_preliminary(r, nthreads=cores)
        f.compute_surface_interaction(r, t, u, v, prim, grid_flags, nthreads=cores)
        f.adjoint(r, t, u, v, prim, grads, grid_flags, nthreads=cores)
        return time.perf_counter() - t0, r.shape[1]

    run(1 << 16)                               # warm up the thread pool
    dt, k = run(1 << 20)                       # calibration
    rate = k / dt
    k2 = int(min(R, max(1 << 20, rate * args.cpu_seconds)))
    dt2, k2 = run(k2)
    return {"value": round(k2 / dt2 / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{k2} rays, every {max(1, R // k2)}-th ray of the wavefront, forward (traversal+SI) + adjoint, "
                      f"{dt2:.1f} s, OpenMP over rays",
            "note": "CPU restatement of the reference scalar path (reference llvm_ad_rgb unavailable offline)"}


if __name__ == "__main__":
    main()
