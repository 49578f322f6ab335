#!/usr/bin/env python3
"""Run individual libhf kernels on the bench workload (for rocprofv3 / A-B timing).
usage: python tools/prof_kernels.py [--grid 4096 --film 1024 --spp 64 --iters 5] kinds...
kinds: fwd prelim si adj test miss mips sec_fwd sec_test (incoherent bounce / shadow rays from the primary hits; sec_*_inc: with the
coherent = false hint) fwd_inc prelim_inc (the primary wavefront with that hint: what a wrong hint costs)"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hf_amd
from hf_amd import _capi, build
if os.environ.get('HF_LIB'):
    build.LIB_PATH = os.environ['HF_LIB']; _capi._build.LIB_PATH = os.environ['HF_LIB']
from hf_amd.shape import _DIFF_ROWS, _fill, _rows

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=4096)
ap.add_argument("--film", type=int, default=1024)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--pad", type=int, default=0, help="floats of padding between the SoA rows of every buffer (row pitch = R + pad; 0 = contiguous rows, 2^28 B apart on the bench wavefront)")
ap.add_argument("kinds", nargs="*", default=["fwd", "prelim", "si", "adj", "test", "miss", "mips", "sec_fwd", "sec_test"])
a = ap.parse_args()
dev = torch.device("cuda", 0)
N, R = a.grid, a.film * a.film * a.spp
lib = _capi.lib()
shape = hf_amd.Heightfield(heightfield=hf_amd.workload.sine_heights(N, N, device=dev), max_height=0.5)
rays_c = hf_amd.workload.ortho_rays(a.film, a.film, a.spp, dev)
P = R + a.pad   # row pitch in elements: every [k, R] buffer below is the first R columns of a [k, P] allocation
def rowbuf(k, zero=False):
    b = (torch.zeros if zero else torch.empty)((k, P), device=dev)
    return b[:, :R]
def rowptrs(v):   # device addresses of the rows of such a view
    return [v.data_ptr() + 4 * P * k for k in range(v.shape[0])]
def rays_struct(v):
    p = rowptrs(v); r = _capi.hf_rays_t()
    for k in range(3): r.o[k] = p[k]; r.d[k] = p[3 + k]
    r.maxt = p[6]
    return r
rays = rowbuf(7); rays.copy_(rays_c)
if a.pad == 0: rays = rays_c
else: del rays_c
pib = rowbuf(4)  # pi: t, u, v, prim_index (as bits) -- four rows of one allocation (separate [R] tensors at pad 0: the same 2^28 B pitch)
t, uv, prim = pib[0], pib[1:3], pib[3].view(torch.int32)
si = rowbuf(18); gsi = rowbuf(18, zero=True)
hit8 = torch.empty(R, dtype=torch.uint8, device=dev)
grad_h = torch.zeros((N, N), device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
r_s = rays_struct(rays)
pi_s = _capi.hf_pi_t(); pp = rowptrs(pib); pi_s.t = pp[0]; pi_s.prim_uv[0] = pp[1]; pi_s.prim_uv[1] = pp[2]; pi_s.prim_index = pp[3]
si_s = _fill(_capi.hf_si_t(), _DIFF_ROWS, rowptrs(si)); g_s = _fill(_capi.hf_si_grad_t(), _DIFF_ROWS, rowptrs(gsi))
flags = int(hf_amd.RayFlags.All)
miss = rowbuf(7); miss.copy_(rays); miss[5] = 1.0; miss[3] = 0; miss[4] = 0; miss[2] = 5.0   # pointing up, above the box
m_s = rays_struct(miss)
fn = {
    "fwd": lambda: _capi.check(lib.hf_ray_intersect(shape._h, R, C.byref(r_s), flags, None, C.byref(pi_s), C.byref(si_s), st)),
    "prelim": lambda: _capi.check(lib.hf_ray_intersect_preliminary(shape._h, R, C.byref(r_s), None, C.byref(pi_s), st)),
    "si": lambda: _capi.check(lib.hf_compute_surface_interaction(shape._h, R, C.byref(r_s), C.byref(pi_s), flags, None, C.byref(si_s), st)),
    "adj": lambda: _capi.check(lib.hf_adjoint(shape._h, R, C.byref(r_s), C.byref(pi_s), flags, None, C.byref(g_s), grad_h.data_ptr(), None, None, st)),
    "test": lambda: _capi.check(lib.hf_ray_test(shape._h, R, C.byref(r_s), None, hit8.data_ptr(), st)),
    "miss": lambda: _capi.check(lib.hf_ray_intersect(shape._h, R, C.byref(m_s), flags, None, C.byref(pi_s), C.byref(si_s), st)),
    "mips": lambda: shape.parameters_changed(["heightfield"]),
}
def _inc(f):   # the same launch with the handle set to incoherent rays (hf_set_ray_coherence)
    def g():
        _capi.check(lib.hf_set_ray_coherence(shape._h, 1)); f(); _capi.check(lib.hf_set_ray_coherence(shape._h, 0))
    return g
fn["fwd_inc"] = _inc(fn["fwd"]); fn["prelim_inc"] = _inc(fn["prelim"]); fn["test_inc"] = _inc(fn["test"])
fn["fwd"](); torch.cuda.synchronize()
h = torch.isfinite(si[0]); gsi[0] = h.float(); gsi[1:4] = si[4:7] * h
if any(k.startswith("sec") for k in a.kinds):
    # secondary rays of SURVEY 8d: one cosine-hemisphere bounce + one shadow ray per primary hit
    hit_idx = torch.nonzero(h).squeeze(1)
    bounce, shadow = hf_amd.workload.secondary_rays(si[1:4][:, hit_idx], si[4:7][:, hit_idx], seed=0)
    Rs = bounce.shape[1]
    b_s = shape._rays_struct(bounce[0:3], bounce[3:6], bounce[6]); s_s = shape._rays_struct(shadow[0:3], shadow[3:6], shadow[6])
    fn["sec_fwd"] = lambda: _capi.check(lib.hf_ray_intersect(shape._h, Rs, C.byref(b_s), flags, None, C.byref(pi_s), C.byref(si_s), st))
    fn["sec_test"] = lambda: _capi.check(lib.hf_ray_test(shape._h, Rs, C.byref(s_s), None, hit8.data_ptr(), st))
    fn["sec_fwd_inc"] = _inc(fn["sec_fwd"]); fn["sec_test_inc"] = _inc(fn["sec_test"])   # coherent = false: what an integrator passes for its secondary rays
    nrays = {"sec_fwd": Rs, "sec_test": Rs, "sec_fwd_inc": Rs, "sec_test_inc": Rs}
else:
    nrays = {}
if any(k.startswith("alive") for k in a.kinds):
    # the rays that enter the (identity to_world) bound, compacted in wavefront order: what a traversal kernel
    # behind a cull + compact pre-pass would see
    bb = shape.bbox().to(dev)
    lo = torch.tensor([-1.0 - 1e-4, -1.0 - 1e-4, float(bb[0, 2]) - 1e-5], device=dev)[:, None]
    hi = torch.tensor([1.0 + 1e-4, 1.0 + 1e-4, float(bb[1, 2]) + 1e-5], device=dev)[:, None]
    inv = 1.0 / rays[3:6]
    t1, t2 = (lo - rays[0:3]) * inv, (hi - rays[0:3]) * inv
    tin = torch.minimum(t1, t2).max(0).values.clamp(min=0); tout = torch.minimum(torch.maximum(t1, t2).min(0).values, rays[6])
    alive_idx = torch.nonzero(tin <= tout).squeeze(1)
    ar = rays[:, alive_idx].contiguous(); Ra = ar.shape[1]
    a_s = shape._rays_struct(ar[0:3], ar[3:6], ar[6])
    fn["alive_fwd"] = lambda: _capi.check(lib.hf_ray_intersect(shape._h, Ra, C.byref(a_s), flags, None, C.byref(pi_s), C.byref(si_s), st))
    fn["alive_prelim"] = lambda: _capi.check(lib.hf_ray_intersect_preliminary(shape._h, Ra, C.byref(a_s), None, C.byref(pi_s), st))
    nrays.update({"alive_fwd": Ra, "alive_prelim": Ra})
for nm, org in (("fwd_ymajor", (0.15, 2.6, 0.9)), ("fwd_xmajor", (2.6, 0.15, 0.9)), ("fwd_steep", (0.3, 0.2, 3.0))):
    if nm in a.kinds:   # other view directions: grazing along y / along x (long paths through the grid), steep from above
        rr = hf_amd.workload.ortho_rays(a.film, a.film, a.spp, dev, origin=org, target=(0.0, 0.0, 0.1), scale=(1.3, 1.3, 1.0))
        rs_ = shape._rays_struct(rr[0:3], rr[3:6], rr[6])
        fn[nm] = (lambda rs_=rs_, rr=rr: _capi.check(lib.hf_ray_intersect(shape._h, R, C.byref(rs_), flags, None, C.byref(pi_s), C.byref(si_s), st)))
if os.environ.get("HF_REPARAM_ONE_LAUNCH") == "0":   # A/B: num_rays x hf_reparam_trace instead of hf_reparam_trace_all
    from hf_amd import shape as _shape_mod
    _shape_mod.REPARAM_ONE_LAUNCH = False
if "reparam16" in a.kinds:   # the reference's default for prb_reparam: 16 auxiliary rays (prb_reparam.py:237)
    hfp = shape.heightfield.requires_grad_(True)
    ray_o16 = hf_amd.Ray3f(rays[0:3], rays[3:6], rays[6])
    gdir16 = torch.randn(3, R, device=dev); gdv16 = torch.randn(R, device=dev)
    def _reparam16():
        shape.heightfield.grad = None
        dd, det = hf_amd.reparameterize_ray(shape, ray_o16, num_rays=16, kappa=1e5, exponent=3.0)
        torch.autograd.backward((dd, det), (gdir16, gdv16))
    fn["reparam16"] = _reparam16
if "reparam" in a.kinds:
    # backward of reparameterize_ray (4 auxiliary rays per primary ray: 8 fused traces + 8 weight kernels + 4 adjoints)
    hfp = shape.heightfield.requires_grad_(True)
    ray_o = hf_amd.Ray3f(rays[0:3], rays[3:6], rays[6])
    gdir = torch.randn(3, R, device=dev); gdv = torch.randn(R, device=dev)
    def _reparam():
        shape.heightfield.grad = None
        dd, det = hf_amd.reparameterize_ray(shape, ray_o, num_rays=4, kappa=float(os.environ.get("HF_PROF_KAPPA", "1e5")), exponent=3.0)
        torch.autograd.backward((dd, det), (gdir, gdv))
    fn["reparam"] = _reparam
def clocks():
    """current sclk / mclk of the first card as sysfs reports them (read right after the timed launches; informative only)"""
    if not os.environ.get("HF_PROF_CLOCKS"): return ""
    import glob
    out = []
    for nm in ("pp_dpm_sclk", "pp_dpm_mclk"):
        try:
            f = sorted(glob.glob(f"/sys/class/drm/card*/device/{nm}"))[0]
            cur = [l.split()[1] for l in open(f) if l.strip().endswith("*")]
            out.append(f"{nm[7:]} {cur[0] if cur else '?'}")
        except Exception as e:
            out.append(f"{nm[7:]} n/a")
    return "  [" + ", ".join(out) + "]"
for k in a.kinds:
    fn[k](); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn[k]()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    nr = nrays.get(k, R)
    print(f"{k:8s} {ms:9.3f} ms  {nr / ms / 1e3:10.1f} Mrays/s  ({nr} rays){clocks()}", flush=True)
