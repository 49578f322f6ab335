"""ctypes binding of the CPU oracle (oracle/hf_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/hf_oracle.h.  Importers:
tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhf_oracle.so")

RAY_MINIMAL, RAY_UV, RAY_DPDUV, RAY_SHADINGFRAME = 0x1, 0x2, 0x4, 0x8
RAY_BOUNDARYTEST, RAY_FOLLOWSHAPE, RAY_DETACHSHAPE = 0x40, 0x80, 0x100
RAY_ALL = RAY_UV | RAY_DPDUV | RAY_SHADINGFRAME

SI_FIELDS = [("t", 1), ("p", 3), ("n", 3), ("uv", 2), ("sh_n", 3), ("dp_du", 3),
             ("dp_dv", 3), ("boundary_test", 1), ("sh_s", 3), ("sh_t", 3), ("wi", 3)]
GRAD_FIELDS = [("t", 1), ("p", 3), ("n", 3), ("uv", 2), ("sh_n", 3), ("dp_du", 3), ("dp_dv", 3)]


def build(force=False):
    """Compile libhf_oracle.so with the committed Makefile."""
    src = os.path.join(_HERE, "hf_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libhf_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_fp = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.hfo_create.restype = C.c_void_p
        L.hfo_create.argtypes = [C.c_int, C.c_int, _fp, C.c_float, _fp, _fp, C.c_int]
        L.hfo_destroy.argtypes = [C.c_void_p]
        L.hfo_set_heights.argtypes = [C.c_void_p, _fp]
        L.hfo_num_levels.argtypes = [C.c_void_p]
        L.hfo_get_mip.argtypes = [C.c_void_p, C.c_int, _fp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.hfo_bbox.argtypes = [C.c_void_p, _fp]
        L.hfo_vertex.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp]
        L.hfo_intersect_batch.argtypes = [C.c_void_p, C.c_int64, C.POINTER(_fp), _u8p, C.c_int,
                                          C.c_int, _fp, _fp, _fp, _u32p]
        L.hfo_ray_test_batch.argtypes = [C.c_void_p, C.c_int64, C.POINTER(_fp), _u8p, C.c_int,
                                         C.c_int, _u8p]
        L.hfo_compute_si_batch.restype = C.c_int
        L.hfo_compute_si_batch.argtypes = [C.c_void_p, C.c_int64, C.POINTER(_fp), _fp, _fp, _fp,
                                           _u32p, _u8p, C.c_uint32, C.c_int, C.POINTER(_fp)]
        L.hfo_adjoint_batch.restype = C.c_int
        L.hfo_adjoint_batch.argtypes = [C.c_void_p, C.c_int64, C.POINTER(_fp), _fp, _fp, _fp,
                                        _u32p, _u8p, C.c_uint32, C.c_int, C.POINTER(_fp), _fp,
                                        C.POINTER(_fp), C.POINTER(_fp)]
        L.hfo_make_sine_heights.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, _fp]
        L.hfo_sample_tea_32.argtypes = [C.c_uint32, C.c_uint32, C.c_int, _u32p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_fp)


def invert_affine(m):
    """Inverse of a row-major 3x4 affine matrix, computed in float64 then rounded."""
    m = np.asarray(m, dtype=np.float64).reshape(3, 4)
    a = np.eye(4)
    a[:3, :] = m
    return np.linalg.inv(a)[:3, :].astype(np.float32)


class OracleField:
    """Scalar CPU heightfield (oracle).  rays: float32 array [7, n] = ox,oy,oz,dx,dy,dz,maxt."""

    def __init__(self, heights, max_height=1.0, to_world=None, to_object=None, flip_normals=False):
        h = _f32(heights)
        assert h.ndim == 2
        self.H, self.W = h.shape
        self.to_world = _f32(np.eye(4)[:3] if to_world is None else to_world).reshape(3, 4)
        self.to_object = (invert_affine(self.to_world) if to_object is None
                          else _f32(to_object).reshape(3, 4))
        self.max_height = float(max_height)
        self._h = lib().hfo_create(self.W, self.H, _p(h), self.max_height,
                                   _p(self.to_world), _p(self.to_object), int(flip_normals))
        if not self._h:
            raise ValueError("hfo_create failed (need W,H >= 2)")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.hfo_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def set_heights(self, heights):
        h = _f32(heights)
        assert h.shape == (self.H, self.W)
        lib().hfo_set_heights(self._h, _p(h))

    def num_levels(self):
        return lib().hfo_num_levels(self._h)

    def mip(self, level):
        w, h = C.c_int(), C.c_int()
        n = lib().hfo_get_mip(self._h, level, None, C.byref(w), C.byref(h))
        out = np.empty((h.value, w.value, 2), np.float32)
        lib().hfo_get_mip(self._h, level, _p(out), C.byref(w), C.byref(h))
        assert n == w.value * h.value
        return out

    def bbox(self):
        out = np.empty(6, np.float32)
        lib().hfo_bbox(self._h, _p(out))
        return out

    def vertex(self, i, j):
        out = np.empty(3, np.float32)
        lib().hfo_vertex(self._h, i, j, _p(out))
        return out

    @staticmethod
    def _rays(rays):
        r = _f32(rays)
        assert r.ndim == 2 and r.shape[0] == 7
        arr = (_fp * 7)(*[_p(r[k]) for k in range(7)])
        return r, arr

    @staticmethod
    def _mask(active, n):
        if active is None:
            return None, None
        a = np.ascontiguousarray(active, dtype=np.uint8)
        assert a.shape == (n,)
        return a, a.ctypes.data_as(_u8p)

    def ray_intersect_preliminary(self, rays, active=None, naive=False, nthreads=0, band=False):
        """naive: brute force over every cell; band: brute force over the cells within +/-2 cells of the ray's xy
        segment (independent of the hierarchical walk: no mips, no margins), affordable at BASELINE grid sizes"""
        naive = 2 if band else int(naive)
        r, arr = self._rays(rays)
        n = r.shape[1]
        a, ap = self._mask(active, n)
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32)
        lib().hfo_intersect_batch(self._h, n, arr, ap, int(naive), nthreads, _p(t), _p(u), _p(v),
                                  prim.ctypes.data_as(_u32p))
        return t, u, v, prim

    def ray_test(self, rays, active=None, naive=False, nthreads=0, band=False):
        naive = 2 if band else int(naive)
        r, arr = self._rays(rays)
        n = r.shape[1]
        a, ap = self._mask(active, n)
        hit = np.empty(n, np.uint8)
        lib().hfo_ray_test_batch(self._h, n, arr, ap, int(naive), nthreads, hit.ctypes.data_as(_u8p))
        return hit.astype(bool)

    def compute_surface_interaction(self, rays, t, u, v, prim, ray_flags=RAY_ALL, active=None,
                                    nthreads=0):
        r, arr = self._rays(rays)
        n = r.shape[1]
        a, ap = self._mask(active, n)
        t, u, v = _f32(t), _f32(u), _f32(v)
        prim = np.ascontiguousarray(prim, dtype=np.uint32)
        out = {name: np.zeros((c, n), np.float32) for name, c in SI_FIELDS}
        ptrs = []
        for name, c in SI_FIELDS:
            ptrs += [_p(out[name][k]) for k in range(c)]
        rc = lib().hfo_compute_si_batch(self._h, n, arr, _p(t), _p(u), _p(v),
                                        prim.ctypes.data_as(_u32p), ap, ray_flags, nthreads,
                                        (_fp * 28)(*ptrs))
        if rc != 0:
            raise RuntimeError("Invalid combination of RayFlags: DetachShape | FollowShape")
        for name, c in SI_FIELDS:
            if c == 1:
                out[name] = out[name][0]
        return out

    def adjoint(self, rays, t, u, v, prim, grads, ray_flags=RAY_ALL, active=None, nthreads=0,
                ray_grads=False):
        """grads: dict field -> array ([c, n] or [n]); missing fields are zero.
        Returns grad_heights [H, W] (and grad_o, grad_d [3, n] if ray_grads)."""
        r, arr = self._rays(rays)
        n = r.shape[1]
        a, ap = self._mask(active, n)
        t, u, v = _f32(t), _f32(u), _f32(v)
        prim = np.ascontiguousarray(prim, dtype=np.uint32)
        keep, ptrs = [], []
        for name, c in GRAD_FIELDS:
            g = grads.get(name)
            if g is None:
                ptrs += [None] * c
            else:
                g = _f32(g).reshape(c, n)
                keep.append(g)
                ptrs += [_p(g[k]) for k in range(c)]
        gh = np.zeros((self.H, self.W), np.float32)
        go = np.zeros((3, n), np.float32); gd = np.zeros((3, n), np.float32)
        goa = (_fp * 3)(*[_p(go[k]) for k in range(3)]) if ray_grads else None
        gda = (_fp * 3)(*[_p(gd[k]) for k in range(3)]) if ray_grads else None
        rc = lib().hfo_adjoint_batch(self._h, n, arr, _p(t), _p(u), _p(v),
                                     prim.ctypes.data_as(_u32p), ap, ray_flags, nthreads,
                                     (_fp * 18)(*ptrs), _p(gh), goa, gda)
        if rc != 0:
            raise RuntimeError("Invalid combination of RayFlags: DetachShape | FollowShape")
        return (gh, go, gd) if ray_grads else gh


def make_sine_heights(W, H, fx, fy):
    out = np.empty((H, W), np.float32)
    lib().hfo_make_sine_heights(W, H, fx, fy, _p(out))
    return out


def sample_tea_32(v0, v1, rounds=4):
    out = (C.c_uint32 * 2)()
    lib().hfo_sample_tea_32(v0, v1, rounds, out)
    return int(out[0]), int(out[1])


def adam_step(h, g, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-8, step=1, mask_updates=False, uniform=False):
    """One Adam step on the height texture, restating mitsuba.ad.Adam.step
    (src/python/python/ad/optimizers.py:263-300) in float32 with one rounding per operation:
    lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)   (scale in double, rounded once, :267-268)
    m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2   (:279-281)
    h = h - lr_t m / (sqrt(v) + eps)                               (:290-295)
    mask_updates: entries with g == 0 keep h, m, v (:282-285, 293-294); uniform: the update divides by
    sqrt(max(v_t)) + eps, the maximum over all entries after the mask (:290-291).  Returns (h, m, v) as new arrays."""
    f = np.float32
    h, g, m, v = (np.asarray(a, f) for a in (h, g, m, v))
    lr_scale = f(np.sqrt(1.0 - float(beta2) ** int(step)) / (1.0 - float(beta1) ** int(step)))
    lr_t = f(f(lr) * lr_scale)
    b1, b2, e = f(beta1), f(beta2), f(eps)
    c1, c2 = f(1.0 - float(beta1)), f(1.0 - float(beta2))   # Python doubles in the reference, rounded once
    mt = (b1 * m + c1 * g).astype(f)
    vt = (b2 * v + c2 * (g * g).astype(f)).astype(f)
    z = (g == 0) if mask_updates else np.zeros(g.shape, bool)
    mt = np.where(z, m, mt); vt = np.where(z, v, vt)
    den = (np.sqrt(f(vt.max())) + e).astype(f) if uniform else (np.sqrt(vt).astype(f) + e).astype(f)
    hn = (h - ((lr_t * mt).astype(f) / den).astype(f)).astype(f)
    hn = np.where(z, h, hn)
    return hn, mt, vt


def direct_lighting(sh_n, d, t, lights, albedo=1.0, spp=1, vis=None):
    """Diffuse direct lighting under directional lights + box-filter film (float64 restatement of
    include/hf.h hf_direct_lighting): emitter-sampling term of direct_reparam.py:149-175 for a delta light
    (MIS weight 1), diffuse BSDF value albedo/pi * cos_o gated by cos_i > 0 and cos_o > 0
    (src/bsdfs/diffuse.cpp:135-140), irradiance of src/emitters/directional.cpp:174, pixel = sample // spp
    (src/render/integrator.cpp:251-268).  lights: [K,4] = (unit direction to the light, irradiance).
    Returns image [K, n // spp]."""
    sh_n = np.asarray(sh_n, np.float64); d = np.asarray(d, np.float64); t = np.asarray(t, np.float64)
    lights = np.asarray(lights, np.float64).reshape(-1, 4)
    n = sh_n.shape[1]
    lit = np.isfinite(t) & (-(sh_n * d).sum(0) > 0)
    out = np.zeros((lights.shape[0], n // spp))
    for k, L in enumerate(lights):
        co = (sh_n * L[:3, None]).sum(0)
        c = np.where(lit & (co > 0), albedo / np.pi * L[3] * co, 0.0)
        if vis is not None:
            c = c * (np.asarray(vis[k]) != 0)
        out[k] = c.reshape(-1, spp).mean(1)
    return out


def direct_lighting_adjoint(sh_n, d, t, lights, grad_image, albedo=1.0, spp=1, vis=None):
    """d(sum(image * grad_image)) / d(sh_n), float64: the cos > 0 masks and the visibility are piecewise constant."""
    sh_n = np.asarray(sh_n, np.float64); d = np.asarray(d, np.float64); t = np.asarray(t, np.float64)
    lights = np.asarray(lights, np.float64).reshape(-1, 4)
    lit = np.isfinite(t) & (-(sh_n * d).sum(0) > 0)
    g = np.zeros_like(sh_n)
    for k, L in enumerate(lights):
        co = (sh_n * L[:3, None]).sum(0)
        w = np.where(lit & (co > 0), albedo / np.pi * L[3] / spp, 0.0) * np.repeat(np.asarray(grad_image, np.float64)[k], spp)
        if vis is not None:
            w = w * (np.asarray(vis[k]) != 0)
        g += w[None, :] * L[:3, None]
    return g


def point_lighting(sh_n, p, d, t, lights, albedo=1.0, spp=1, vis=None):
    """direct_lighting under point lights (float64 restatement of include/hf.h hf_point_lighting): the emitter of
    src/emitters/point.cpp:106-123 -- direction (position - p) / r, radiance intensity / r^2 -- with the same BSDF
    term, masks and film.  lights: [K,4] = (position, intensity).  Returns image [K, n // spp]."""
    sh_n = np.asarray(sh_n, np.float64); p = np.asarray(p, np.float64); d = np.asarray(d, np.float64)
    t = np.asarray(t, np.float64)
    lights = np.asarray(lights, np.float64).reshape(-1, 4)
    n = sh_n.shape[1]
    lit = np.isfinite(t) & (-(sh_n * d).sum(0) > 0)
    out = np.zeros((lights.shape[0], n // spp))
    for k, L in enumerate(lights):
        v = L[:3, None] - p
        r2 = (v * v).sum(0)
        l = v / np.sqrt(r2)
        co = (sh_n * l).sum(0)
        c = np.where(lit & (co > 0), albedo / np.pi * L[3] / r2 * co, 0.0)
        if vis is not None:
            c = c * (np.asarray(vis[k]) != 0)
        out[k] = c.reshape(-1, spp).mean(1)
    return out


def point_lighting_adjoint(sh_n, p, d, t, lights, grad_image, albedo=1.0, spp=1, vis=None):
    """d(sum(image * grad_image)) / d(sh_n) and / d(p), float64, by the closed form f = w <n, v> |v|^-3, v = position - p:
    df/dn = w v |v|^-3,  df/dp = w |v|^-3 (3 <n, l> l - n)."""
    sh_n = np.asarray(sh_n, np.float64); p = np.asarray(p, np.float64); d = np.asarray(d, np.float64)
    t = np.asarray(t, np.float64)
    lights = np.asarray(lights, np.float64).reshape(-1, 4)
    lit = np.isfinite(t) & (-(sh_n * d).sum(0) > 0)
    gn = np.zeros_like(sh_n); gp = np.zeros_like(p)
    for k, L in enumerate(lights):
        v = L[:3, None] - p
        r = np.sqrt((v * v).sum(0))
        l = v / r
        co = (sh_n * l).sum(0)
        w = np.where(lit & (co > 0), albedo / np.pi * L[3] / spp, 0.0) * np.repeat(np.asarray(grad_image, np.float64)[k], spp)
        if vis is not None:
            w = w * (np.asarray(vis[k]) != 0)
        gn += (w / r ** 2)[None, :] * l
        gp += (w / r ** 3)[None, :] * (3.0 * co[None, :] * l - sh_n)
    return gn, gp


def _gauss_w(x, stddev):
    """src/rfilters/gaussian.cpp:48-101 (the exponential form of its CUDA branch): windowed Gaussian, radius 4 stddev"""
    alpha = -1.0 / (2.0 * stddev * stddev)
    r = 4.0 * stddev
    return np.maximum(np.exp(alpha * x * x) - np.exp(alpha * r * r), 0.0)


def film_splat(values, pos, width, height, stddev=0.5):
    """ImageBlock::put with a Gaussian reconstruction filter (src/render/imageblock.cpp:258-330), float64: returns
    (accumulated image [K, H*W], accumulated weight [H*W]); the film is image / weight."""
    values = np.asarray(values, np.float64); pos = np.asarray(pos, np.float64)
    K, n = values.shape
    img = np.zeros((K, height * width)); wgt = np.zeros(height * width)
    r = 4.0 * stddev
    for i in range(n):
        fx, fy = pos[0, i] - 0.5, pos[1, i] - 0.5
        x0, x1 = max(int(np.ceil(fx - r)), 0), min(int(np.floor(fx + r)), width - 1)
        y0, y1 = max(int(np.ceil(fy - r)), 0), min(int(np.floor(fy + r)), height - 1)
        if x0 > x1 or y0 > y1:
            continue
        wx = _gauss_w(np.arange(x0, x1 + 1) - fx, stddev); wy = _gauss_w(np.arange(y0, y1 + 1) - fy, stddev)
        w = wy[:, None] * wx[None, :]
        pix = (np.arange(y0, y1 + 1)[:, None] * width + np.arange(x0, x1 + 1)[None, :])
        np.add.at(wgt, pix, w)
        for k in range(K):
            np.add.at(img[k], pix, w * values[k, i])
    return img, wgt


def film_splat_adjoint(pos, width, height, grad_image, stddev=0.5):
    """d(sum(image * grad_image)) / d(values): grad_values[k, i] = sum_pixels w(i, pixel) grad_image[k, pixel]"""
    pos = np.asarray(pos, np.float64); grad_image = np.asarray(grad_image, np.float64)
    K = grad_image.shape[0]; n = pos.shape[1]
    g = np.zeros((K, n))
    r = 4.0 * stddev
    for i in range(n):
        fx, fy = pos[0, i] - 0.5, pos[1, i] - 0.5
        x0, x1 = max(int(np.ceil(fx - r)), 0), min(int(np.floor(fx + r)), width - 1)
        y0, y1 = max(int(np.ceil(fy - r)), 0), min(int(np.floor(fy + r)), height - 1)
        if x0 > x1 or y0 > y1:
            continue
        wx = _gauss_w(np.arange(x0, x1 + 1) - fx, stddev); wy = _gauss_w(np.arange(y0, y1 + 1) - fy, stddev)
        w = wy[:, None] * wx[None, :]
        pix = (np.arange(y0, y1 + 1)[:, None] * width + np.arange(x0, x1 + 1)[None, :])
        g[:, i] = (grad_image[:, pix] * w[None]).sum((1, 2))
    return g


# ---------------------------------------------------------------------------------------------------
# Warped-area reparameterisation of rays for a scene that is this one shape: float64 restatement of
# src/python/python/ad/reparam.py:10-123 (_sample_warp_field) and :151-333 (forward / backward of
# _ReparameterizeOp).  Random numbers: sample_tea_32(key, ray id) with key = sample_tea_32(seed, pair)[0] -> two 23-bit
# floats, the documented stand-in for the PCG32 of the absent Dr.Jit (see include/hf.h); ray id = the ray's index in
# the launch unless `ray_id` names it (a rank's rays of a partitioned wavefront).
# ---------------------------------------------------------------------------------------------------
def _tea32_np(v0, v1, rounds=4):
    v0 = np.asarray(v0, np.uint64) & 0xFFFFFFFF; v1 = np.asarray(v1, np.uint64) & 0xFFFFFFFF
    M = np.uint64(0xFFFFFFFF); s = np.uint64(0)
    for _ in range(rounds):
        s = (s + np.uint64(0x9E3779B9)) & M
        v0 = (v0 + ((((v1 << np.uint64(4)) & M) + np.uint64(0xA341316C)) ^ ((v1 + s) & M) ^ (((v1 >> np.uint64(5)) + np.uint64(0xC8013EA4)) & M))) & M
        v1 = (v1 + ((((v0 << np.uint64(4)) & M) + np.uint64(0xAD90777D)) ^ ((v0 + s) & M) ^ (((v0 >> np.uint64(5)) + np.uint64(0x7E95761E)) & M))) & M
    return v0, v1


def _coordinate_system_np(n):
    """include/mitsuba/core/vector.h:116-136"""
    sign = np.where(n[2] >= 0, 1.0, -1.0)
    a = -1.0 / (sign + n[2]); b = n[0] * n[1] * a
    ms = lambda x: np.where(n[2] >= 0, x, -x)
    s = np.stack([ms(n[0] * n[0] * a) + 1.0, ms(b), ms(-n[0])])
    t = np.stack([b, n[1] * n[1] * a + sign, -n[1]])
    return s, t


_REPARAM_RAY_ID = [None]   # module-level: the ray ids of the current reparam_* call (set by with_ray_ids)


class with_ray_ids:
    """context manager: the reparam_* functions inside draw the samples of rays `ids` (uint32 [n]) instead of 0..n-1"""
    def __init__(self, ids):
        self.ids = None if ids is None else np.asarray(ids, np.uint64)
    def __enter__(self):
        self.prev = _REPARAM_RAY_ID[0]; _REPARAM_RAY_ID[0] = self.ids
    def __exit__(self, *a):
        _REPARAM_RAY_ID[0] = self.prev


def reparam_aux_sample(d, k, kappa, antithetic=False, seed=0):
    """omega_local, sample.y and the frame of auxiliary ray k of every primary direction d [3,n] (float32 inputs,
    float64 maths): warp.h:557-566, reparam.py:80-90."""
    d = np.asarray(d, np.float32).astype(np.float64)
    n = d.shape[1]
    pair = (k >> 1) if antithetic else k
    key, _ = _tea32_np(np.uint64(seed & 0xFFFFFFFF), np.uint64(pair))
    ids = _REPARAM_RAY_ID[0] if _REPARAM_RAY_ID[0] is not None else np.arange(n, dtype=np.uint64)
    assert ids.shape == (n,)
    r0, r1 = _tea32_np(np.full(n, key, np.uint64), ids)
    f = np.float32
    sx = (np.float32(r0 >> np.uint64(9)) * f(1.0 / 8388608.0)).astype(f); sy = (np.float32(r1 >> np.uint64(9)) * f(1.0 / 8388608.0)).astype(f)
    # warp.h:557-566 in float32, one rounding per operation: 1 - cos_theta^2 cancels, so sin_theta carries the
    # rounding of cos_theta -- the reference's Float does the same
    syc = np.maximum(f(1) - sy, f(1e-6)).astype(f)
    cos_t = (f(1) + (np.log(((f(1) - syc) * np.exp(f(-2.0) * f(kappa)) + syc).astype(f)).astype(f) / f(kappa)).astype(f)).astype(f)
    sin_t = np.sqrt(np.maximum(f(1) - (cos_t * cos_t).astype(f), f(0))).astype(f)
    ang = (f(6.283185307179586) * sx).astype(f)
    om = np.stack([(np.cos(ang).astype(f) * sin_t).astype(f), (np.sin(ang).astype(f) * sin_t).astype(f), cos_t]).astype(np.float64)
    sy = sy.astype(np.float64)
    if antithetic and (k & 1) == 0:
        om[0] = -om[0]; om[1] = -om[1]
    fs, ft = _coordinate_system_np(d)
    return om, sy, fs, ft


def reparam_aux_rays(o, d, k, kappa, antithetic=False, seed=0, active=None):
    """[7,n] float32 auxiliary rays (reparam.py:87-90); inactive lanes get maxt = -1."""
    o = np.asarray(o, np.float32); d = np.asarray(d, np.float32)
    om, sy, fs, ft = reparam_aux_sample(d, k, kappa, antithetic, seed)
    ad = fs * om[0] + ft * om[1] + d.astype(np.float64) * om[2]
    maxt = np.full(o.shape[1], np.inf)
    if active is not None:
        maxt[np.asarray(active) == 0] = -1.0
    return np.concatenate([o, ad, maxt[None]]).astype(np.float32)


def _reparam_weight(d, k, kappa, exponent, antithetic, seed, t, bt):
    """reparam.py:97-121.  The weight is (1 / (D - 1 + B))^exponent * D with D - 1 + B down to 1e-4: it is
    evaluated in float32, one rounding per operation in the reference's order, like Dr.Jit's Float does."""
    f = np.float32
    om, sy, fs, ft = reparam_aux_sample(d, k, kappa, antithetic, seed)
    hit = np.isfinite(t)
    B = np.where(hit, bt, 1.0).astype(f)
    sy32 = sy.astype(f)
    inv_vmf = (f(1) / (sy32 * np.exp(f(-2.0) * f(kappa)) + (f(1) - sy32)).astype(f)).astype(f)
    w_denom = ((inv_vmf - f(1)) + B).astype(f)
    ok = w_denom > f(1e-4)
    w_rcp = np.where(ok, f(1) / np.where(ok, w_denom, f(1)), f(0)).astype(f)
    w = (np.power(w_rcp, f(exponent)).astype(f) * inv_vmf).astype(f)
    tmp1 = np.clip((((inv_vmf * w).astype(f) * w_rcp).astype(f) * f(kappa)).astype(f) * f(exponent), f(-1e10), f(1e10)).astype(f)
    tmp2 = (fs * om[0] + ft * om[1]).astype(f)
    dw = (tmp1 * tmp2).astype(f)
    return hit, w.astype(np.float64), dw.astype(np.float64)


def reparam_backward(field, o, d, grad_direction, grad_divergence, num_rays=4, kappa=1e5, exponent=3.0,
                     antithetic=False, seed=0, active=None, nthreads=0, ray_grads=False):
    """dL/dheight through reparameterize_ray (reparam.py:224-333, backward_symbolic): given the upstream gradients
    of its outputs (direction [3,n], divergence [n]).  With ray_grads also (dL/d ray.o, dL/d ray.d) [3,n] each
    (reparam.py:296-325): per sample L_i = <gVd_i, V_direct_i(o, d)> with everything else detached, where
    V_direct = (p - o) / t, t = |p - o| / |d_aux|, d_aux = Frame3f(d).to_world(omega) for a hit (p is glued to the
    shape: FollowShape) and V_direct = d for a miss; differentiated here by float64 central differences of exactly
    that expression -- independent of the analytic chain the kernels and the host mirror use."""
    o = np.asarray(o, np.float32); d = np.asarray(d, np.float32)
    n = o.shape[1]
    act = np.ones(n, bool) if active is None else (np.asarray(active) != 0)
    flags = RAY_ALL | 0x80 | 0x40
    Z = np.zeros(n); dZ = np.zeros((3, n)); recs = []
    for k in range(num_rays):                                     # first loop: Z, dZ
        r = reparam_aux_rays(o, d, k, kappa, antithetic, seed, active)
        t, u, v, prim = field.ray_intersect_preliminary(r, nthreads=nthreads)
        si = field.compute_surface_interaction(r, t, u, v, prim, flags, nthreads=nthreads)
        hit, w, dw = _reparam_weight(d, k, kappa, exponent, antithetic, seed, si["t"].astype(np.float64), si["boundary_test"].astype(np.float64))
        Z += np.where(act, w, 0.0); dZ += np.where(act, dw, 0.0)
        recs.append((r, t, u, v, prim, si, hit & act, w, dw))
    Z = np.maximum(Z, 1e-8)
    dd = d.astype(np.float64); gd = np.asarray(grad_direction, np.float64); gdiv = np.asarray(grad_divergence, np.float64)
    n2 = (dd * dd).sum(0)
    gV = (gd - dd * ((dd * gd).sum(0) / n2)) / np.sqrt(n2) / Z - gdiv / (Z * Z) * dZ
    gdivV = gdiv / Z
    gh = np.zeros((field.H, field.W), np.float64)
    go_ray = np.zeros((3, n)); gd_ray = np.zeros((3, n))
    for k, (r, t, u, v, prim, si, hit, w, dw) in enumerate(recs):  # third loop: back-propagate every sample
        gVd = w * gV + gdivV * dw
        if ray_grads:
            om = reparam_aux_sample(d, k, kappa, antithetic, seed)[0]
            p64 = si["p"].astype(np.float64)
            def L(o_, d_):
                fs, ft = _coordinate_system_np(d_)
                da = fs * om[0] + ft * om[1] + d_ * om[2]
                po = p64 - o_
                tt = np.sqrt((po * po).sum(0) / (da * da).sum(0))
                Vd = np.where(hit, po / np.where(hit, tt, 1.0), d_)
                return (gVd * Vd).sum(0) * act                     # per ray: rays are independent
            o64 = o.astype(np.float64)
            eps = 1e-6
            for c in range(3):
                e = np.zeros((3, 1)); e[c] = eps
                go_ray[c] += (L(o64 + e, dd) - L(o64 - e, dd)) / (2 * eps)
                gd_ray[c] += (L(o64, dd + e) - L(o64, dd - e)) / (2 * eps)
        tt = np.where(hit, si["t"].astype(np.float64), 1.0)
        po = si["p"].astype(np.float64) - o.astype(np.float64)
        gp = np.where(hit, gVd / tt, 0.0)
        gt = np.where(hit, -(gVd * po).sum(0) / (tt * tt), 0.0)
        gh += field.adjoint(r, t, u, v, prim, {"p": gp.astype(np.float32), "t": gt.astype(np.float32)[None]}, flags, nthreads=nthreads)
    return (gh, go_ray, gd_ray) if ray_grads else gh


def reparam_forward(field, o, d, dheights, num_rays=4, kappa=1e5, exponent=3.0, antithetic=False, seed=0, active=None):
    """Forward mode (reparam.py:151-220) for a height perturbation dheights [H,W]: returns (V_theta [3,n],
    div_V_theta [n]).  dV_direct/dtheta for a FollowShape hit: d(p)/dtheta = sum_k b_k s zhat_world dh_k,
    V_direct = (p - o)/t with t = |p - o| (unit auxiliary direction)."""
    o = np.asarray(o, np.float32); d = np.asarray(d, np.float32)
    n = o.shape[1]
    act = np.ones(n, bool) if active is None else (np.asarray(active) != 0)
    flags = RAY_ALL | 0x80 | 0x40
    Z = np.zeros(n); dZ = np.zeros((3, n)); gradV = np.zeros((3, n)); graddiv = np.zeros(n)
    zw = np.asarray(field.to_world, np.float64).reshape(3, 4)[:, 2] * field.max_height
    dh = np.asarray(dheights, np.float64)
    for k in range(num_rays):
        r = reparam_aux_rays(o, d, k, kappa, antithetic, seed, active)
        t, u, v, prim = field.ray_intersect_preliminary(r)
        si = field.compute_surface_interaction(r, t, u, v, prim, flags)
        hit, w, dw = _reparam_weight(d, k, kappa, exponent, antithetic, seed, si["t"].astype(np.float64), si["boundary_test"].astype(np.float64))
        hit = hit & act
        # vertices of the hit triangles and their barycentric weights
        cell = (prim >> 1).astype(np.int64); tri = (prim & 1).astype(np.int64)
        cy, cx = cell // (field.W - 1), cell % (field.W - 1)
        vi = np.where(tri == 0, np.stack([cy, cy, cy + 1]), np.stack([cy + 1, cy + 1, cy]))
        vj = np.where(tri == 0, np.stack([cx, cx + 1, cx]), np.stack([cx + 1, cx, cx + 1]))
        b1, b2 = u.astype(np.float64), v.astype(np.float64); bw = np.stack([1 - b1 - b2, b1, b2])
        vi = np.where(hit, vi, 0); vj = np.where(hit, vj, 0)
        dz = (bw * dh[vi, vj]).sum(0)                               # d(height of p)/dtheta in height units
        dp = zw[:, None] * dz[None, :]
        po = si["p"].astype(np.float64) - o.astype(np.float64)
        tt = np.where(hit, si["t"].astype(np.float64), 1.0)
        dt = (po * dp).sum(0) / tt                                  # t = |p - o|
        dVd = np.where(hit, dp / tt - po * dt / (tt * tt), 0.0)
        Z += np.where(act, w, 0.0); dZ += np.where(act, dw, 0.0)
        gradV += w * dVd; graddiv += (dw * dVd).sum(0)
    iZ = 1.0 / np.maximum(Z, 1e-8)
    Vt = gradV * iZ
    div = (graddiv - (Vt * dZ).sum(0)) * iZ
    return np.where(act, Vt, 0.0), np.where(act, div, 0.0)
