"""Host-side mirror of the reference's Shape plugin interface for the heightfield
hot path, on top of the C ABI (include/hf.h).  Names, argument meaning and error
behaviour follow include/mitsuba/render/shape.h:137-183 and its Python surface
src/render/python/shape_v.cpp:51-102; the data types follow
include/mitsuba/core/ray.h:24-82 and include/mitsuba/render/interaction.h.

torch is used for device memory, streams and autograd plumbing only; all
arithmetic happens in the HIP kernels of libhf.so.  There is no CPU fallback.
"""
import ctypes as C
import enum
import math

import torch

from . import _capi
from ._capi import check, hf_desc_t, hf_pi_t, hf_rays_t, hf_si_grad_t, hf_si_t


class RayFlags(enum.IntFlag):
    """include/mitsuba/render/interaction.h:19-69"""
    Empty = 0x0
    Minimal = 0x1
    UV = 0x2
    dPdUV = 0x4
    ShadingFrame = 0x8
    dNGdUV = 0x10
    dNSdUV = 0x20
    BoundaryTest = 0x40
    FollowShape = 0x80
    DetachShape = 0x100
    All = 0x2 | 0x4 | 0x8
    AllNonDifferentiable = 0x2 | 0x4 | 0x8 | 0x100
    # libhf extension (include/hf.h HF_RAY_BOUNDARY_ALL_EDGES): boundary_test over all three triangle edges (the
    # reference Mesh's semantics) instead of the silhouette edges only
    BoundaryAllEdges = 0x10000


class ParamFlags(enum.IntFlag):
    """include/mitsuba/render/fwd / object.h ParamFlags"""
    Differentiable = 0x0
    NonDifferentiable = 0x1
    Discontinuous = 0x2


def _as_f32(x, device):
    t = torch.as_tensor(x, dtype=torch.float32, device=device)
    return t.contiguous()


class Ray3f:
    """SoA ray wavefront: o, d as [3, n] float32 tensors, maxt [n] (default +inf,
    i.e. dr::Largest, ray.h:37).  time / wavelengths are carried but unused."""

    def __init__(self, o, d, maxt=None, time=0.0, wavelengths=None):
        dev = o.device if isinstance(o, torch.Tensor) else (d.device if isinstance(d, torch.Tensor) else "cuda")
        self.o = _as_f32(o, dev)
        self.d = _as_f32(d, dev)
        if self.o.dim() == 1:
            self.o = self.o.reshape(3, 1)
        if self.d.dim() == 1:
            self.d = self.d.reshape(3, 1)
        n = max(self.o.shape[1], self.d.shape[1])
        if self.o.shape[1] != n:
            self.o = self.o.expand(3, n).contiguous()
        if self.d.shape[1] != n:
            self.d = self.d.expand(3, n).contiguous()
        assert self.o.shape == (3, n) and self.d.shape == (3, n), "Ray3f expects [3, n] SoA tensors"
        if maxt is None:
            self.maxt = torch.full((n,), math.inf, dtype=torch.float32, device=self.o.device)
        else:
            self.maxt = _as_f32(maxt, self.o.device).reshape(-1)
            if self.maxt.numel() == 1 and n != 1:
                self.maxt = self.maxt.expand(n).contiguous()
        self.time = time
        self.wavelengths = wavelengths

    def __len__(self):
        return self.o.shape[1]

    def __call__(self, t):
        """ray(t) = fmadd(d, t, o), ray.h:57"""
        return torch.addcmul(self.o, self.d, t)

    @property
    def device(self):
        return self.o.device


class Frame3f:
    def __init__(self, s, t, n):
        self.s, self.t, self.n = s, t, n

    def to_local(self, v):
        return torch.stack([(v * self.s).sum(0), (v * self.t).sum(0), (v * self.n).sum(0)])


class PreliminaryIntersection3f:
    """interaction.h:587-691"""

    def __init__(self, t, prim_uv, prim_index, shape):
        self.t = t
        self.prim_uv = prim_uv
        self.prim_index = prim_index
        # (uint32_t) -1 for a non-instanced shape (rectangle.cpp:222)
        self.shape_index = torch.full_like(prim_index, -1)
        self.shape = shape
        self.instance = None

    def is_valid(self):
        return self.t != math.inf

    def compute_surface_interaction(self, ray, ray_flags=RayFlags.All, active=True):
        """interaction.h:658-684: shape->compute_surface_interaction + finalize."""
        return self.shape.compute_surface_interaction(ray, self, ray_flags, 0, active)


class SurfaceInteraction3f:
    """interaction.h:175-507 (fields of DRJIT_STRUCT :504-506 that a static shape fills)"""

    def __init__(self):
        self.t = self.p = self.n = self.uv = None
        self.sh_frame = None
        self.dp_du = self.dp_dv = self.dn_du = self.dn_dv = None
        self.duv_dx = self.duv_dy = None
        self.wi = None
        self.prim_index = None
        self.boundary_test = None
        self.shape = None
        self.instance = None
        self.time = 0.0
        self.wavelengths = None

    def is_valid(self):
        return self.t != math.inf

    def spawn_ray(self, d):
        """Semi-infinite ray from the interaction towards ``d`` ([3] or [3, n]); the origin is offset along
        the detached normal by (1 + max|p|) * RayEpsilon, signed by <n, d> (interaction.h:134-136, 161-165;
        RayEpsilon = 1500 * eps/2, math.h:18-22)."""
        d = torch.as_tensor(d, dtype=torch.float32, device=self.p.device)
        if d.dim() == 1:
            d = d[:, None].expand(3, self.p.shape[1])
        p, n = self.p.detach(), self.n.detach()
        mag = (1.0 + p.abs().max(0).values) * (1500.0 * 5.9604644775390625e-08)
        mag = torch.where((n * d).sum(0) < 0, -mag, mag)
        o = torch.addcmul(p, mag[None, :], n)
        maxt = torch.full((p.shape[1],), math.inf, dtype=torch.float32, device=p.device)
        return Ray3f(o.contiguous(), d.contiguous(), maxt)


# order of the differentiable SI block handed to autograd: 18 rows
_DIFF_ROWS = [("t", 1), ("p", 3), ("n", 3), ("uv", 2), ("sh_n", 3), ("dp_du", 3), ("dp_dv", 3)]
_AUX_ROWS = [("boundary_test", 1), ("sh_s", 3), ("sh_t", 3), ("wi", 3)]


def _rows(buf, n):
    """device addresses of the rows of a contiguous [k, n] float32 tensor"""
    base = buf.data_ptr()
    return [base + 4 * n * k for k in range(buf.shape[0])]


def _fill(struct, layout, addrs):
    k = 0
    for name, c in layout:
        if c == 1:
            setattr(struct, name, addrs[k])
        else:
            arr = getattr(struct, name)
            for j in range(c):
                arr[j] = addrs[k + j]
        k += c
    return struct


class _SurfaceInteractionOp(torch.autograd.Function):
    """Differentiable SI block [18, n]; backward = hf_adjoint (atomic scatter of dL/dheight)."""

    @staticmethod
    def forward(ctx, shape, heights, o, d, maxt, t, uv, prim, flags, active, diff_block):
        ctx.shape, ctx.flags, ctx.active = shape, flags, active
        ctx.save_for_backward(o, d, maxt, t, uv, prim)
        ctx.h_version = shape._heights_version
        return diff_block

    @staticmethod
    def backward(ctx, g):
        shape = ctx.shape
        o, d, maxt, t, uv, prim = ctx.saved_tensors
        if ctx.h_version != shape._heights_version:
            raise RuntimeError("heightfield parameters changed between forward and backward")
        n = o.shape[1]
        g = g.contiguous().to(torch.float32)
        need_h, need_o, need_d = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        grad_h = torch.zeros((shape.height, shape.width), dtype=torch.float32, device=o.device) if need_h else None
        grad_od = torch.empty((6, n), dtype=torch.float32, device=o.device) if (need_o or need_d) else None
        shape._adjoint_raw(o, d, maxt, t, uv, prim, ctx.flags, ctx.active, g, grad_h, grad_od)
        go = grad_od[0:3] if need_o else None
        gd = grad_od[3:6] if need_d else None
        return None, grad_h, go, gd, None, None, None, None, None, None, None


class Heightfield:
    """`heightfield` shape plugin mirror.

    Properties (build decision, SURVEY.md section 8a -- the reference snapshot has no
    heightfield plugin): `heightfield` ([H, W] or [H, W, 1] tensor, row 0 at object
    y = -1; cf. TensorXf(data, 3, {H,W,C}) in src/textures/bitmap.cpp:262),
    `max_height`, `to_world` (3x4 / 4x4 affine), `flip_normals`.
    Object space is Rectangle's: x,y in [-1,1], +Z up (src/shapes/rectangle.cpp:47-48).
    """

    def __init__(self, props=None, **kw):
        props = dict(props or {}, **kw)
        props.pop("type", None)
        hfield = props.pop("heightfield")
        self.max_height = float(props.pop("max_height", 1.0))
        to_world = props.pop("to_world", None)
        self.flip_normals = bool(props.pop("flip_normals", False))
        device = props.pop("device", None)
        if props:
            raise RuntimeError(f"Unreferenced properties: {sorted(props)}")  # Properties semantics
        if not torch.cuda.is_available():
            raise RuntimeError("Heightfield: no HIP device available (libhf has no CPU fallback)")
        if device is None:
            device = hfield.device if isinstance(hfield, torch.Tensor) and hfield.is_cuda else torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        h = torch.as_tensor(hfield, dtype=torch.float32)
        if h.dim() == 3 and h.shape[2] == 1:
            h = h[:, :, 0]
        if h.dim() != 2:
            raise RuntimeError("heightfield: expected a [H, W] or [H, W, 1] tensor")
        self.height, self.width = int(h.shape[0]), int(h.shape[1])
        tw = torch.eye(4, dtype=torch.float64)[:3] if to_world is None else torch.as_tensor(to_world, dtype=torch.float64).cpu()
        tw = tw.reshape(-1)[:12].reshape(3, 4)
        self.to_world = tw.to(torch.float32)
        desc = hf_desc_t()
        desc.width, desc.height = self.width, self.height
        desc.max_height = self.max_height
        for k, v in enumerate(self.to_world.reshape(-1).tolist()):
            desc.to_world[k] = v
        desc.has_to_object = 0
        desc.flip_normals = int(self.flip_normals)
        desc.device = self.device.index if self.device.index is not None else torch.cuda.current_device()
        handle = C.c_void_p()
        check(_capi.lib().hf_create(C.byref(desc), C.byref(handle)))  # HF_EINVAL if H or W < 2
        self._h = handle
        self._dirty = True
        self._heights_version = 0
        # the differentiable parameter (put_parameter("heightfield", ..., Differentiable|Discontinuous))
        self.heightfield = h.to(self.device).contiguous().clone()
        self.parameters_changed(["heightfield"])

    # ---- lifetime ---------------------------------------------------------------------
    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _capi.lib().hf_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- parameter plumbing (shape.cpp:536-570, rectangle.cpp:126-142) ---------------
    def traverse(self, callback):
        callback.put_parameter("heightfield", self.heightfield, ParamFlags.Differentiable | ParamFlags.Discontinuous)
        callback.put_parameter("max_height", self.max_height, ParamFlags.NonDifferentiable)
        callback.put_parameter("to_world", self.to_world, ParamFlags.NonDifferentiable)

    def parameters_changed(self, keys=()):
        keys = list(keys)
        if not keys or "heightfield" in keys:
            h = self.heightfield
            if h.dim() == 3 and h.shape[2] == 1:
                h = h[:, :, 0]
            if tuple(h.shape) != (self.height, self.width):
                # bitmap.cpp:272-286: resolution may not change / must stay >= 2
                raise RuntimeError(f"heightfield: tensor shape {tuple(h.shape)} != ({self.height}, {self.width})")
            hd = h.detach().to(device=self.device, dtype=torch.float32).contiguous()
            stream = torch.cuda.current_stream(self.device).cuda_stream
            check(_capi.lib().hf_set_heights(self._h, hd.data_ptr(), stream))
            self._heights_keepalive = hd
            self._heights_version += 1
        if not keys or "to_world" in keys:
            tw = (C.c_float * 12)(*self.to_world.reshape(-1).tolist())
            check(_capi.lib().hf_set_transform(self._h, tw, None))
        self.mark_dirty()

    def parameters_grad_enabled(self):
        return bool(self.heightfield.requires_grad)

    def mark_dirty(self):
        self._dirty = True

    def dirty(self):
        return self._dirty

    def primitive_count(self):
        """one kd-tree primitive (shape.cpp:526-529); the cells are internal"""
        return 1

    def effective_primitive_count(self):
        return 2 * (self.width - 1) * (self.height - 1)

    def is_mesh(self):
        return False

    def bbox(self):
        out = (C.c_float * 6)()
        check(_capi.lib().hf_bbox(self._h, out))
        return torch.tensor(list(out), dtype=torch.float32).reshape(2, 3)

    def num_levels(self):
        return _capi.lib().hf_num_levels(self._h)

    def mip(self, level):
        w, h = C.c_uint32(), C.c_uint32()
        check(_capi.lib().hf_get_mip(self._h, level, None, C.byref(w), C.byref(h)))
        out = torch.empty((h.value, w.value, 2), dtype=torch.float32)
        check(_capi.lib().hf_get_mip(self._h, level, out.data_ptr(), C.byref(w), C.byref(h)))
        return out

    # ---- helpers -------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _check_ray(self, ray):
        if not isinstance(ray, Ray3f):
            raise TypeError("expected a Ray3f")
        if ray.device != self.device:
            raise RuntimeError(f"ray lives on {ray.device}, shape on {self.device}")

    @staticmethod
    def _rays_struct(o, d, maxt):
        n = o.shape[1]
        r = hf_rays_t()
        for k in range(3):
            r.o[k] = o.data_ptr() + 4 * n * k
            r.d[k] = d.data_ptr() + 4 * n * k
        r.maxt = maxt.data_ptr()
        return r

    def _mask(self, active, n):
        if active is True or active is None:
            return None, None
        if active is False:
            active = torch.zeros(n, dtype=torch.bool, device=self.device)
        a = torch.as_tensor(active, device=self.device).to(torch.uint8).contiguous()
        if a.numel() == 1 and n != 1:
            a = a.expand(n).contiguous()
        assert a.shape == (n,)
        return a, a.data_ptr()

    @staticmethod
    def _pi_struct(t, uv, prim):
        n = t.shape[0]
        p = hf_pi_t()
        p.t = t.data_ptr()
        p.prim_uv[0] = uv.data_ptr()
        p.prim_uv[1] = uv.data_ptr() + 4 * n
        p.prim_index = prim.data_ptr()
        return p

    # ---- the hot path ---------------------------------------------------------------------
    # ---- the `coherent` hint of Scene::ray_intersect / ray_test / ray_intersect_preliminary (scene.h:117-146) ----------
    COHERENCE_AUTO, COHERENCE_INCOHERENT, COHERENCE_COHERENT = 0, 1, 2

    def set_ray_coherence(self, mode):
        """``hf_set_ray_coherence``: which kernels the trace launches that follow take -- ``COHERENCE_AUTO`` (default:
        every 64-ray batch decides for itself), ``COHERENCE_INCOHERENT`` (= ``coherent=False``: bounce rays, auxiliary
        rays; kernels without the beam sweep at 6-7 waves per SIMD) or ``COHERENCE_COHERENT``.  Results never depend on it."""
        check(_capi.lib().hf_set_ray_coherence(self._h, int(mode)))

    def ray_coherence(self):
        return int(_capi.lib().hf_get_ray_coherence(self._h))

    class _Coherence:
        """``with shape._coherent(flag):`` -- the hint for the launches inside (None: the handle's mode as it is)"""
        def __init__(self, shape, flag):
            self.shape, self.flag = shape, flag
        def __enter__(self):
            if self.flag is not None:
                self.old = self.shape.ray_coherence()
                self.shape.set_ray_coherence(Heightfield.COHERENCE_COHERENT if self.flag else Heightfield.COHERENCE_INCOHERENT)
        def __exit__(self, *exc):
            if self.flag is not None:
                self.shape.set_ray_coherence(self.old)
            return False

    def _coherent(self, flag):
        return Heightfield._Coherence(self, flag)

    def ray_intersect_preliminary(self, ray, active=True, coherent=None):
        """shape.h:137-138; ``coherent``: the hint of Scene::ray_intersect_preliminary (scene.h:237-259), None = the handle's mode"""
        self._check_ray(ray)
        n = len(ray)
        t = torch.empty(n, dtype=torch.float32, device=self.device)
        uv = torch.empty((2, n), dtype=torch.float32, device=self.device)
        prim = torch.empty(n, dtype=torch.int32, device=self.device)
        keep, ap = self._mask(active, n)
        rays = self._rays_struct(ray.o, ray.d, ray.maxt)
        pi = self._pi_struct(t, uv, prim)
        with self._coherent(coherent):
            check(_capi.lib().hf_ray_intersect_preliminary(self._h, n, C.byref(rays), ap, C.byref(pi), self._stream()))
        return PreliminaryIntersection3f(t, uv, prim, self)

    def ray_test(self, ray, active=True, coherent=None):
        """shape.h:153; ``coherent``: scene.h:188-207"""
        self._check_ray(ray)
        n = len(ray)
        hit = torch.empty(n, dtype=torch.uint8, device=self.device)
        keep, ap = self._mask(active, n)
        rays = self._rays_struct(ray.o, ray.d, ray.maxt)
        with self._coherent(coherent):
            check(_capi.lib().hf_ray_test(self._h, n, C.byref(rays), ap, hit.data_ptr(), self._stream()))
        return hit.bool()

    # ---- scalar / packet forms (shape.h:220-240): host arrays in, host arrays out -------------------
    def ray_intersect_preliminary_packet(self, o, d, maxt=None, active=None):
        """``ray_intersect_preliminary_packet`` / ``_scalar`` (shape.h:220-240, the per-kd-leaf call of
        kdtree.h:2490-2520): up to 16 rays held in HOST memory (numpy / CPU tensors, [3, n] or [3]).  Returns
        host arrays ``(t, prim_uv [2, n], prim_index)``.  Same kernel and arithmetic as the wavefront entry,
        one synchronous launch per call (``hf_ray_intersect_preliminary_packet``)."""
        import numpy as np
        o, d, maxt, act, n = self._host_packet(o, d, maxt, active)
        t = np.empty(n, np.float32); uv = np.empty((2, n), np.float32); prim = np.empty(n, np.uint32)
        op, dp = self._host_rows(o), self._host_rows(d)
        uvp = (C.c_void_p * 2)(uv[0].ctypes.data, uv[1].ctypes.data)
        check(_capi.lib().hf_ray_intersect_preliminary_packet(self._h, n, C.byref(op), C.byref(dp), maxt.ctypes.data,
                                                              act.ctypes.data if act is not None else None,
                                                              t.ctypes.data, C.byref(uvp), prim.ctypes.data))
        return t, uv, prim

    def ray_intersect_preliminary_scalar(self, o, d, maxt=math.inf):
        t, uv, prim = self.ray_intersect_preliminary_packet(o, d, maxt)
        return float(t[0]), (float(uv[0, 0]), float(uv[1, 0])), int(prim[0])

    def ray_test_packet(self, o, d, maxt=None, active=None):
        import numpy as np
        o, d, maxt, act, n = self._host_packet(o, d, maxt, active)
        hit = np.empty(n, np.uint8)
        op, dp = self._host_rows(o), self._host_rows(d)
        check(_capi.lib().hf_ray_test_packet(self._h, n, C.byref(op), C.byref(dp), maxt.ctypes.data,
                                             act.ctypes.data if act is not None else None, hit.ctypes.data))
        return hit.astype(bool)

    def ray_test_scalar(self, o, d, maxt=math.inf):
        return bool(self.ray_test_packet(o, d, maxt)[0])

    @staticmethod
    def _host_rows(a):
        return (C.c_void_p * 3)(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data)

    @staticmethod
    def _host_packet(o, d, maxt, active):
        import numpy as np
        o = np.ascontiguousarray(np.asarray(o, np.float32).reshape(3, -1))
        d = np.ascontiguousarray(np.asarray(d, np.float32).reshape(3, -1))
        n = max(o.shape[1], d.shape[1])
        o = np.ascontiguousarray(np.broadcast_to(o, (3, n))); d = np.ascontiguousarray(np.broadcast_to(d, (3, n)))
        maxt = np.full(n, math.inf, np.float32) if maxt is None else \
            np.ascontiguousarray(np.broadcast_to(np.asarray(maxt, np.float32).reshape(-1), (n,)))
        act = None if active is None else np.ascontiguousarray(np.broadcast_to(np.asarray(active).astype(np.uint8).reshape(-1), (n,)))
        return o, d, maxt, act, n

    def _package_si(self, ray, pi_t, pi_prim, diff, aux, ray_flags):
        n = len(ray)
        si = SurfaceInteraction3f()
        si.t = diff[0]
        si.p, si.n, si.uv = diff[1:4], diff[4:7], diff[7:9]
        sh_n = diff[9:12]
        si.dp_du, si.dp_dv = diff[12:15], diff[15:18]
        si.boundary_test = aux[0] if (ray_flags & RayFlags.BoundaryTest) else torch.zeros(n, device=self.device)
        sh_s, sh_t, wi = aux[1:4], aux[4:7], aux[7:10]
        if diff.requires_grad and (ray_flags & RayFlags.ShadingFrame):
            # finalize_surface_interaction is AD-attached in the reference (interaction.h:257-267, 476-499):
            # sh_frame.s = normalize(dp_du - n <n, dp_du>), t = cross(n, s), wi = to_local(-d).  The kernel's rows are
            # plain outputs, so when gradients are wanted these three are rebuilt here from the differentiable rows
            # (sh_frame.n, dp_du, ray.d): a BSDF that reads cos_theta = wi.z then back-propagates to the heights.
            dp_du = si.dp_du
            s_un = dp_du - sh_n * (sh_n * dp_du).sum(0, keepdim=True)
            nrm = torch.linalg.norm(s_un, dim=0, keepdim=True)
            ok = (nrm > 0) & torch.isfinite(diff[0])[None, :]
            s_at = s_un / torch.where(ok, nrm, torch.ones_like(nrm))
            sh_s = torch.where(ok, s_at, aux[1:4])           # degenerate dp_du / misses: the kernel's (detached) rows
            sh_t = torch.where(ok, torch.linalg.cross(sh_n, sh_s, dim=0), aux[4:7])
            md = -ray.d
            wi = torch.where(ok, torch.stack([(md * sh_s).sum(0), (md * sh_t).sum(0), (md * sh_n).sum(0)]), aux[7:10])
        si.sh_frame = Frame3f(sh_s, sh_t, sh_n)
        si.wi = wi
        zeros3 = torch.zeros((3, n), dtype=torch.float32, device=self.device)
        si.dn_du, si.dn_dv = zeros3, zeros3          # flat shading
        si.duv_dx = si.duv_dy = torch.zeros((2, n), dtype=torch.float32, device=self.device)
        si.prim_index = pi_prim                       # interaction.h:486
        si.shape = self
        si.time, si.wavelengths = ray.time, ray.wavelengths
        return si

    def _wants_grad(self, ray, ray_flags):
        if not torch.is_grad_enabled():
            return False
        h_live = self.heightfield.requires_grad and not (ray_flags & RayFlags.DetachShape)
        return bool(h_live or ray.o.requires_grad or ray.d.requires_grad)

    def compute_surface_interaction(self, ray, pi, ray_flags=RayFlags.All, recursion_depth=0, active=True):
        """shape.h:179-183 + finalize_surface_interaction (interaction.h:476-499)"""
        self._check_ray(ray)
        ray_flags = int(ray_flags)
        n = len(ray)
        diff = torch.empty((18, n), dtype=torch.float32, device=self.device)
        aux = torch.empty((10, n), dtype=torch.float32, device=self.device)
        if recursion_depth > 0:   # mesh.cpp:680-682: early exit, zero-initialised record
            diff.zero_(); aux.zero_()
            return self._package_si(ray, pi.t, pi.prim_index, diff, aux, ray_flags)
        keep, ap = self._mask(active, n)
        out = _fill(_fill(hf_si_t(), _DIFF_ROWS, _rows(diff, n)), _AUX_ROWS, _rows(aux, n))
        rays = self._rays_struct(ray.o, ray.d, ray.maxt)
        pis = self._pi_struct(pi.t, pi.prim_uv, pi.prim_index)
        check(_capi.lib().hf_compute_surface_interaction(self._h, n, C.byref(rays), C.byref(pis), ray_flags, ap,
                                                         C.byref(out), self._stream()))
        if self._wants_grad(ray, ray_flags):
            diff = _SurfaceInteractionOp.apply(self, self.heightfield, ray.o, ray.d, ray.maxt, pi.t, pi.prim_uv,
                                               pi.prim_index, ray_flags, keep, diff)
        return self._package_si(ray, pi.t, pi.prim_index, diff, aux, ray_flags)

    def ray_intersect(self, ray, ray_flags=RayFlags.All, active=True, coherent=None):
        """shape.cpp:436-446: preliminary intersection + surface interaction, one fused kernel; ``coherent``: scene.h:117-146"""
        self._check_ray(ray)
        ray_flags = int(ray_flags)
        n = len(ray)
        t = torch.empty(n, dtype=torch.float32, device=self.device)
        uv = torch.empty((2, n), dtype=torch.float32, device=self.device)
        prim = torch.empty(n, dtype=torch.int32, device=self.device)
        diff = torch.empty((18, n), dtype=torch.float32, device=self.device)
        aux = torch.empty((10, n), dtype=torch.float32, device=self.device)
        keep, ap = self._mask(active, n)
        out = _fill(_fill(hf_si_t(), _DIFF_ROWS, _rows(diff, n)), _AUX_ROWS, _rows(aux, n))
        rays = self._rays_struct(ray.o, ray.d, ray.maxt)
        pis = self._pi_struct(t, uv, prim)
        with self._coherent(coherent):
            check(_capi.lib().hf_ray_intersect(self._h, n, C.byref(rays), ray_flags, ap, C.byref(pis), C.byref(out),
                                               self._stream()))
        if self._wants_grad(ray, ray_flags):
            diff = _SurfaceInteractionOp.apply(self, self.heightfield, ray.o, ray.d, ray.maxt, t, uv, prim,
                                               ray_flags, keep, diff)
        si = self._package_si(ray, t, prim, diff, aux, ray_flags)
        si.prim_uv = uv
        return si

    # ---- adjoint ------------------------------------------------------------------------------
    def _adjoint_raw(self, o, d, maxt, t, uv, prim, ray_flags, active_u8, g, grad_h, grad_od, row_band=None):
        n = o.shape[1]
        rays = self._rays_struct(o, d, maxt)
        pis = self._pi_struct(t, uv, prim)
        gs = _fill(hf_si_grad_t(), _DIFF_ROWS, _rows(g, n))
        go = gd = None
        if grad_od is not None:
            rows = _rows(grad_od, n)
            go = (C.c_void_p * 3)(*rows[0:3])
            gd = (C.c_void_p * 3)(*rows[3:6])
        check(_capi.lib().hf_adjoint_rows(self._h, n, C.byref(rays), C.byref(pis), int(ray_flags),
                                          active_u8.data_ptr() if active_u8 is not None else None, C.byref(gs),
                                          grad_h.data_ptr() if grad_h is not None else None,
                                          C.byref(go) if go is not None else None,
                                          C.byref(gd) if gd is not None else None,
                                          row_band.data_ptr() if row_band is not None else None, self._stream()))

    def new_row_band(self):
        """{height, 0} as int32[2] on the device: the initial value of hf_adjoint_rows' row band."""
        return torch.tensor([self.height, 0], dtype=torch.int32, device=self.device)

    def adjoint(self, ray, pi, grad_si, ray_flags=RayFlags.All, active=True, grad_heightfield=None,
                ray_grads=False, row_band=None):
        """Explicit adjoint: accumulate dL/dheight for upstream gradients `grad_si`
        ([18, n]: t, p, n, uv, sh_frame.n, dp_du, dp_dv) into `grad_heightfield` ([H, W]).
        row_band (int32[2] device tensor from new_row_band(), optional): updated to {lowest texture row that
        received a contribution, highest + 1} (hf_adjoint_rows): what a multi-GPU host needs to all-reduce."""
        self._check_ray(ray)
        n = len(ray)
        g = _as_f32(grad_si, self.device)
        assert g.shape == (18, n)
        if grad_heightfield is None:
            grad_heightfield = torch.zeros((self.height, self.width), dtype=torch.float32, device=self.device)
        keep, _ = self._mask(active, n)
        grad_od = torch.empty((6, n), dtype=torch.float32, device=self.device) if ray_grads else None
        self._adjoint_raw(ray.o, ray.d, ray.maxt, pi.t, pi.prim_uv, pi.prim_index, ray_flags, keep, g,
                          grad_heightfield, grad_od, row_band)
        if ray_grads:
            return grad_heightfield, grad_od[0:3], grad_od[3:6]
        return grad_heightfield


def allreduce_gradient(grad, group=None):
    """Sum the per-GPU dL/dheight textures (one RCCL all-reduce over xGMI; rays are
    sharded over ranks, heights are replicated -- SURVEY.md section 8e)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    return grad


class Adam:
    """Adam on the heightfield parameter of one shape, the mirror of ``mitsuba.ad.Adam``
    (src/python/python/ad/optimizers.py:204-310) + ``params.update()`` (util.py:185-232):
    ``step()`` runs ``hf_adam_step`` -- the update of the parameter tensor in place and the rebuild
    of the shape's acceleration data -- in one call on the current stream.  State (m, v, t) lives here,
    like ``Optimizer.state`` / ``Adam.t``.  ``uniform``: the 'UniformAdam' variant (optimizers.py:259, 290-291)."""

    def __init__(self, shape, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-8, mask_updates=False, uniform=False):
        assert 0 <= beta_1 < 1 and 0 <= beta_2 < 1 and lr > 0 and epsilon > 0  # optimizers.py:248-249
        self.shape, self.lr, self.beta_1, self.beta_2, self.epsilon = shape, lr, beta_1, beta_2, epsilon
        self.mask_updates, self.uniform = mask_updates, uniform
        self.reset()

    def reset(self):
        """zero-initialise the optimiser state (optimizers.py:303-309)"""
        h = self.shape.heightfield
        self.state = (torch.zeros_like(h, requires_grad=False), torch.zeros_like(h, requires_grad=False))
        self.t = 0

    def set_learning_rate(self, lr):
        self.lr = lr

    def zero_grad(self):
        self.shape.heightfield.grad = None

    def step(self):
        h = self.shape.heightfield
        g = h.grad
        if g is None:  # optimizers.py:274-275: nothing to do without a gradient
            return
        if not (h.is_cuda and h.dtype == torch.float32 and h.is_contiguous()):
            raise RuntimeError("Adam: the heightfield parameter must be a contiguous float32 device tensor")
        g = g.to(dtype=torch.float32).contiguous()
        self.t += 1
        m, v = self.state
        stream = torch.cuda.current_stream(h.device).cuda_stream
        check(_capi.lib().hf_adam_step(self.shape._h, h.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                       self.lr, self.beta_1, self.beta_2, self.epsilon, self.t,
                                       (1 if self.mask_updates else 0) | (2 if self.uniform else 0), stream))
        self.shape._heights_keepalive = h
        self.shape._heights_version += 1
        self.shape.mark_dirty()


def _f3(x):
    """[3, n] float32 device tensor -> (keepalive, ctypes array of 3 row pointers)"""
    x = x.to(dtype=torch.float32).contiguous()
    return x, (C.c_void_p * 3)(x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr())


class _DirectLightingOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sh_n, d, t, lights, albedo, spp, vis, weight):
        n = sh_n.shape[1]
        K = lights.shape[0]
        sn, sn_p = _f3(sh_n)
        dd, dd_p = _f3(d)
        tt = t.to(dtype=torch.float32).contiguous()
        ww = weight.detach().to(dtype=torch.float32).contiguous() if weight is not None else None
        L = (_capi.hf_dir_light_t * K)()
        lh = lights.detach().cpu().tolist()
        for k in range(K):
            L[k].to_light[0], L[k].to_light[1], L[k].to_light[2], L[k].irradiance = lh[k]
        vis_p = None
        if vis is not None:
            vis = vis.to(dtype=torch.uint8).contiguous()
            vis_p = (C.c_void_p * K)(*[vis[k].data_ptr() for k in range(K)])
        image = torch.empty((K, n // spp), dtype=torch.float32, device=sh_n.device)
        stream = torch.cuda.current_stream(sh_n.device).cuda_stream
        check(_capi.lib().hf_direct_lighting_weighted(n, spp, C.byref(sn_p), C.byref(dd_p), tt.data_ptr(),
                                                      ww.data_ptr() if ww is not None else None, K, L, albedo,
                                                      vis_p, image.data_ptr(), stream))
        ctx.save_for_backward(sn, dd, tt, *([ww] if ww is not None else []))
        ctx.misc = (L, K, albedo, spp, vis, vis_p, ww is not None)
        return image

    @staticmethod
    def backward(ctx, grad_image):
        L, K, albedo, spp, vis, vis_p, weighted = ctx.misc
        sn, dd, tt = ctx.saved_tensors[:3]
        ww = ctx.saved_tensors[3] if weighted else None
        n = sn.shape[1]
        _, sn_p = _f3(sn)
        _, dd_p = _f3(dd)
        gi = grad_image.to(dtype=torch.float32).contiguous()
        gn = torch.empty_like(sn)
        gw = torch.empty(n, dtype=torch.float32, device=sn.device) if weighted else None
        gn_p = (C.c_void_p * 3)(gn[0].data_ptr(), gn[1].data_ptr(), gn[2].data_ptr())
        stream = torch.cuda.current_stream(sn.device).cuda_stream
        check(_capi.lib().hf_direct_lighting_weighted_adjoint(
            n, spp, C.byref(sn_p), C.byref(dd_p), tt.data_ptr(), ww.data_ptr() if weighted else None, K, L, albedo,
            vis_p, gi.data_ptr(), C.byref(gn_p), gw.data_ptr() if weighted else None, stream))
        return gn, None, None, None, None, None, None, gw


def direct_lighting(si, ray, lights, albedo=1.0, spp=1, vis=None, weight=None):
    """Diffuse direct lighting under directional lights + box-filter film, on the wavefront
    (``hf_direct_lighting``; the emitter-sampling term of direct_reparam.py:149-175 with diffuse.cpp:135-140).
    ``lights``: [K, 4] tensor of (unit direction towards the light, irradiance); ``vis``: optional [K, n] uint8,
    0 = shadowed (``~shape.ray_test(shadow ray)``); ``weight``: optional [n] per-sample factor -- the determinant of a
    reparameterised camera ray (direct_reparam.py:164-180), differentiable.  Returns the [K, n // spp] images;
    differentiable with respect to ``si.sh_frame.n`` (``hf_direct_lighting_adjoint``), which carries the gradient on
    to ``hf_adjoint`` and the heights, and to ``weight``."""
    lights = torch.as_tensor(lights, dtype=torch.float32)
    return _DirectLightingOp.apply(si.sh_frame.n, ray.d, si.t, lights, float(albedo), int(spp), vis, weight)


class _PointLightingOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sh_n, p, d, t, lights, albedo, spp, vis):
        n = sh_n.shape[1]
        K = lights.shape[0]
        sn, sn_p = _f3(sh_n)
        pp, pp_p = _f3(p)
        dd, dd_p = _f3(d)
        tt = t.to(dtype=torch.float32).contiguous()
        L = (_capi.hf_dir_light_t * K)()   # hf_point_light_t: the same packing (position, intensity)
        lh = lights.detach().cpu().tolist()
        for k in range(K):
            L[k].to_light[0], L[k].to_light[1], L[k].to_light[2], L[k].irradiance = lh[k]
        vis_p = None
        if vis is not None:
            vis = vis.to(dtype=torch.uint8).contiguous()
            vis_p = (C.c_void_p * K)(*[vis[k].data_ptr() for k in range(K)])
        image = torch.empty((K, n // spp), dtype=torch.float32, device=sh_n.device)
        stream = torch.cuda.current_stream(sh_n.device).cuda_stream
        check(_capi.lib().hf_point_lighting(n, spp, C.byref(sn_p), C.byref(dd_p), tt.data_ptr(), C.byref(pp_p), K, L,
                                            albedo, vis_p, image.data_ptr(), stream))
        ctx.save_for_backward(sn, pp, dd, tt)
        ctx.misc = (L, K, albedo, spp, vis, vis_p)
        return image

    @staticmethod
    def backward(ctx, grad_image):
        sn, pp, dd, tt = ctx.saved_tensors
        L, K, albedo, spp, vis, vis_p = ctx.misc
        n = sn.shape[1]
        _, sn_p = _f3(sn)
        _, pp_p = _f3(pp)
        _, dd_p = _f3(dd)
        gi = grad_image.to(dtype=torch.float32).contiguous()
        gn = torch.empty_like(sn); gp = torch.empty_like(pp)
        gn_p = (C.c_void_p * 3)(gn[0].data_ptr(), gn[1].data_ptr(), gn[2].data_ptr())
        gp_p = (C.c_void_p * 3)(gp[0].data_ptr(), gp[1].data_ptr(), gp[2].data_ptr())
        stream = torch.cuda.current_stream(sn.device).cuda_stream
        check(_capi.lib().hf_point_lighting_adjoint(n, spp, C.byref(sn_p), C.byref(dd_p), tt.data_ptr(), C.byref(pp_p), K,
                                                    L, albedo, vis_p, gi.data_ptr(), C.byref(gn_p), C.byref(gp_p), stream))
        return gn, gp, None, None, None, None, None, None


def point_lighting(si, ray, lights, albedo=1.0, spp=1, vis=None):
    """``direct_lighting`` under POINT lights (``hf_point_lighting``; src/emitters/point.cpp): ``lights`` is a [K, 4]
    tensor of (position, radiant intensity).  Differentiable with respect to ``si.sh_frame.n`` and ``si.p``."""
    lights = torch.as_tensor(lights, dtype=torch.float32)
    return _PointLightingOp.apply(si.sh_frame.n, si.p, ray.d, si.t, lights, float(albedo), int(spp), vis)


class _FilmGaussianOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, pos, width, height, stddev):
        K, n = values.shape
        v = values.detach().to(dtype=torch.float32).contiguous()
        ps = pos.detach().to(dtype=torch.float32).contiguous()
        image = torch.zeros((K, height * width), dtype=torch.float32, device=v.device)
        weight = torch.zeros(height * width, dtype=torch.float32, device=v.device)
        vp = (C.c_void_p * K)(*[v[k].data_ptr() for k in range(K)])
        stream = torch.cuda.current_stream(v.device).cuda_stream
        check(_capi.lib().hf_film_splat(n, K, vp, ps[0].data_ptr(), ps[1].data_ptr(), width, height, stddev,
                                        image.data_ptr(), weight.data_ptr(), stream))
        ctx.save_for_backward(ps, weight)
        ctx.misc = (K, n, width, height, stddev)
        covered = weight > 0
        return torch.where(covered[None], image / torch.where(covered, weight, torch.ones_like(weight))[None],
                           torch.zeros_like(image))

    @staticmethod
    def backward(ctx, grad_film):
        ps, weight = ctx.saved_tensors
        K, n, width, height, stddev = ctx.misc
        covered = weight > 0
        ga = torch.where(covered[None], grad_film.to(torch.float32) / torch.where(covered, weight, torch.ones_like(weight))[None],
                         torch.zeros_like(grad_film, dtype=torch.float32)).contiguous()
        gv = torch.empty((K, n), dtype=torch.float32, device=ps.device)
        gp = (C.c_void_p * K)(*[gv[k].data_ptr() for k in range(K)])
        stream = torch.cuda.current_stream(ps.device).cuda_stream
        check(_capi.lib().hf_film_splat_adjoint(n, K, ps[0].data_ptr(), ps[1].data_ptr(), width, height, stddev,
                                                ga.data_ptr(), gp, stream))
        return gv, None, None, None, None


def film_gaussian(values, pos, width, height, stddev=0.5):
    """Film with the reference's default reconstruction filter (Gaussian, stddev 0.5 pixel; ``hf_film_splat``;
    src/rfilters/gaussian.cpp, src/render/imageblock.cpp:258-330): ``values`` [K, n] per-sample values (e.g.
    ``direct_lighting(..., spp=1)``), ``pos`` [2, n] film positions in pixels (``workload.film_positions``).  Returns the
    normalised film [K, height * width] (accumulated value / accumulated weight); differentiable w.r.t. ``values``."""
    return _FilmGaussianOp.apply(values, pos, int(width), int(height), float(stddev))


def _p3(x):
    return (C.c_void_p * 3)(x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr())


def _coordinate_system(n):
    """coordinate_system(), include/mitsuba/core/vector.h:116-136, on [3, k] tensors (differentiable)"""
    sign = torch.where(n[2] >= 0, torch.ones_like(n[2]), -torch.ones_like(n[2]))
    a = -1.0 / (sign + n[2])
    b = n[0] * n[1] * a
    s = torch.stack([torch.where(n[2] >= 0, n[0] * n[0] * a, -(n[0] * n[0] * a)) + 1.0,
                     torch.where(n[2] >= 0, b, -b),
                     torch.where(n[2] >= 0, -n[0], n[0])])
    t = torch.stack([b, n[1] * (n[1] * a) + sign, -n[1]])
    return s, t


REPARAM_FUSED = True   # tests switch this off to compare with the per-sample kernels
REPARAM_ONE_LAUNCH = True   # ... and this one to compare hf_reparam_trace_all with num_rays x hf_reparam_trace


class _ReparameterizeOp(torch.autograd.Function):
    """reparam.py:126-333 for a scene that is one heightfield: identity in primal mode; in backward mode the
    warped-area gradient of (direction, determinant) with respect to the heights AND the ray (reparam.py:296-325
    accumulates grad(ray.o), grad(ray.d) over the auxiliary samples)."""

    @staticmethod
    def forward(ctx, heightfield, ray_o, ray_d, shape, num_rays, kappa, exponent, antithetic, seed, active, ray_index=None):
        ctx.shape = shape
        ctx.save_for_backward(ray_o, ray_d)
        ctx.cfg = (int(num_rays), float(kappa), float(exponent), bool(antithetic), int(seed), active, ray_index)
        n = ray_o.shape[1]
        # (an alias of ray.d, not a copy -- 0.8 GB and 0.27 ms for the bench wavefront: the values are ray.d's, reparam.py:139-155)
        return ray_d.detach(), torch.ones(n, dtype=torch.float32, device=ray_o.device)

    @staticmethod
    def backward(ctx, grad_direction, grad_divergence):
        shape = ctx.shape
        ray_o, ray_d = ctx.saved_tensors
        num_rays, kappa, exponent, antithetic, seed, active, ray_index = ctx.cfg
        rid_p = ray_index.data_ptr() if ray_index is not None else None
        need_h, need_o, need_d = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        ray_grads = need_o or need_d
        L = _capi.lib()
        dev = ray_o.device
        n = ray_o.shape[1]
        o = ray_o.detach().to(torch.float32).contiguous(); d = ray_d.detach().to(torch.float32).contiguous()
        gd = grad_direction.to(torch.float32).contiguous(); gdiv = grad_divergence.to(torch.float32).contiguous()
        act = None if active is None else active.to(torch.uint8).contiguous()
        act_p = None if act is None else act.data_ptr()
        stream = torch.cuda.current_stream(dev).cuda_stream
        flags = int(RayFlags.All | RayFlags.FollowShape | RayFlags.BoundaryTest)
        # The auxiliary hits of the first loop (36 B per ray and sample: pi + si.t, si.p, si.boundary_test) are kept
        # for the second one when they fit; the reference re-traces (reparam.py:296-325), which is the fallback.
        keep = 36 * n * num_rays <= (64 << 30)   # (288 GB of HBM per GPU: 16 samples of a 67 M-ray wavefront are 39 GB)
        # heights only and the hits kept: after the traces ONE kernel does the rest (hf_reparam_backward: the weights of
        # all samples, their sums, and the adjoint of every auxiliary hit); the traces then only write pi and
        # si.boundary_test
        fused = REPARAM_FUSED and keep and not ray_grads and num_rays <= 32
        # (the per-sample kernels' intermediates: auxiliary rays, weight sums, upstream gradients of the auxiliary hits --
        # none of them exists on the fused path, where 1.3 GB of zero fills for a 67 M-ray wavefront would be 0.2 ms)
        m = 0 if fused else n
        aux_d = torch.empty((3, m), dtype=torch.float32, device=dev); aux_maxt = torch.empty(m, dtype=torch.float32, device=dev)
        Z = torch.zeros(m, dtype=torch.float32, device=dev); dZ = torch.zeros((3, m), dtype=torch.float32, device=dev)
        g_p = torch.empty((3, m), dtype=torch.float32, device=dev); g_t = torch.empty(m, dtype=torch.float32, device=dev)
        grad_h = torch.zeros((shape.height, shape.width), dtype=torch.float32, device=dev)
        o_p, d_p, ad_p, dZ_p, gd_p, gp_p = _p3(o), _p3(d), _p3(aux_d), _p3(dZ), _p3(gd), _p3(g_p)
        r_s = shape._rays_struct(o, aux_d, aux_maxt)
        g_s = _capi.hf_si_grad_t()
        g_s.t = g_t.data_ptr()
        for c in range(3):
            g_s.p[c] = g_p[c].data_ptr()
        if ray_grads:   # per-sample ray gradients of the auxiliary hit + the gradient w.r.t. V_direct itself
            g_vd = torch.empty((3, n), dtype=torch.float32, device=dev); gvd_p = _p3(g_vd)
            go_adj = torch.empty((3, n), dtype=torch.float32, device=dev); gd_adj = torch.empty((3, n), dtype=torch.float32, device=dev)
            go_adj_p, gd_adj_p = _p3(go_adj), _p3(gd_adj)
            grad_o = torch.zeros((3, n), dtype=torch.float32, device=dev); grad_d = torch.zeros((3, n), dtype=torch.float32, device=dev)
        store = torch.empty((num_rays if keep else 1, 9, n), dtype=torch.float32, device=dev)
        bufs = [store[k] for k in range(store.shape[0])]

        def structs(buf):
            rows = _rows(buf, n)   # si.t, si.p[3], boundary_test | pi.t, u, v, prim_index
            si_s = _capi.hf_si_t()
            if not fused:
                si_s.t = rows[0]
                for c in range(3):
                    si_s.p[c] = rows[1 + c]
            si_s.boundary_test = rows[4]
            pi_s = _capi.hf_pi_t()
            pi_s.t, pi_s.prim_uv[0], pi_s.prim_uv[1], pi_s.prim_index = rows[5], rows[6], rows[7], rows[8]
            return rows, si_s, pi_s

        def aux(k):
            check(L.hf_reparam_aux_rays(n, C.byref(o_p), C.byref(d_p), act_p, k, kappa, int(antithetic), seed, rid_p,
                                        C.byref(ad_p), aux_maxt.data_ptr(), stream))

        def trace(k, buf):
            rows, si_s, pi_s = structs(buf)
            if fused:   # auxiliary ray generated inside the trace kernel
                check(L.hf_reparam_trace(shape._h, n, C.byref(o_p), C.byref(d_p), act_p, k, kappa, int(antithetic), seed, rid_p,
                                         C.byref(pi_s), C.byref(si_s), stream))
                return
            aux(k)
            check(L.hf_ray_intersect(shape._h, n, C.byref(r_s), flags, None, C.byref(pi_s), C.byref(si_s), stream))

        def weights(mode, k, buf):
            rows = _rows(buf, n)
            sp_p = (C.c_void_p * 3)(*rows[1:4])
            check(L.hf_reparam_weights(mode, n, C.byref(o_p), C.byref(d_p), act_p, k, kappa, exponent, int(antithetic),
                                       seed, rid_p, rows[0], C.byref(sp_p), rows[4], Z.data_ptr(), C.byref(dZ_p), C.byref(gd_p),
                                       gdiv.data_ptr(), C.byref(gp_p), g_t.data_ptr(),
                                       C.byref(gvd_p) if (ray_grads and mode == 1) else None, stream))

        if fused and REPARAM_ONE_LAUNCH:    # all samples in one launch (a ray is fetched once; batches whose cones miss are culled)
            rows, si_s, pi_s = structs(bufs[0])
            check(L.hf_reparam_trace_all(shape._h, n, C.byref(o_p), C.byref(d_p), act_p, num_rays, kappa, int(antithetic), seed,
                                         rid_p, C.byref(pi_s), C.byref(si_s), 9 * n, stream))
        for k in range(0 if (fused and REPARAM_ONE_LAUNCH) else num_rays):   # weight normalisation (reparam.py:236-256)
            buf = bufs[k if keep else 0]
            trace(k, buf)
            if not fused:
                weights(0, k, buf)
        if fused and need_h:
            rows, si_s, pi_s = structs(bufs[0])   # sample k: the same rows of store[k], 9 n floats further on
            check(L.hf_reparam_backward(shape._h, n, C.byref(o_p), C.byref(d_p), act_p, num_rays, kappa, exponent,
                                        int(antithetic), seed, rid_p, C.byref(pi_s), rows[4], 9 * n, C.byref(gd_p),
                                        gdiv.data_ptr(), grad_h.data_ptr(), stream))
        for k in range(0 if fused else num_rays):   # the same, per-sample kernels (ray gradients wanted / hits not kept)
            buf = bufs[k if keep else 0]
            if keep:
                aux(k)                      # hf_adjoint needs the auxiliary ray again, not its trace
            else:
                trace(k, buf)
            weights(1, k, buf)
            rows, si_s, pi_s = structs(buf)
            check(L.hf_adjoint(shape._h, n, C.byref(r_s), C.byref(pi_s), flags, act_p, C.byref(g_s),
                               grad_h.data_ptr() if need_h else None,
                               C.byref(go_adj_p) if ray_grads else None, C.byref(gd_adj_p) if ray_grads else None, stream))
            if ray_grads:
                hit = torch.isfinite(buf[0])
                if act is not None:
                    hit = hit & (act != 0)
                # hit: V_direct = (p - o) / t  ->  dL/do = [t's dependence on o: hf_adjoint] - gVd / t  (= g_p);
                #      t's dependence on the auxiliary direction d_aux = Frame3f(d).to_world(omega) goes on to d
                # miss: V_direct = ray.d (reparam.py:93-95)  ->  dL/dd = gVd
                grad_o += torch.where(hit, go_adj - g_p, torch.zeros_like(g_p))
                if need_d:
                    with torch.enable_grad():
                        dq = d.detach().clone().requires_grad_(True)
                        s_, t_ = _coordinate_system(dq)
                        sd, td = s_.detach(), t_.detach()
                        om = torch.stack([(sd * aux_d).sum(0), (td * aux_d).sum(0), (d * aux_d).sum(0)])  # omega_local
                        d_aux = s_ * om[0] + t_ * om[1] + dq * om[2]
                        (gd_through,) = torch.autograd.grad(d_aux, dq, torch.where(hit, gd_adj, torch.zeros_like(gd_adj)))
                    grad_d += gd_through + torch.where(hit, torch.zeros_like(g_vd), g_vd)
        gh = grad_h
        if gh.shape != shape.heightfield.shape:
            gh = gh.reshape(shape.heightfield.shape)
        return ((gh if need_h else None), (grad_o if need_o else None), (grad_d if need_d else None),
                None, None, None, None, None, None, None, None)


def reparameterize_ray(shape, ray, num_rays=4, kappa=1e5, exponent=3.0, antithetic=False, seed=0, active=None,
                       ray_index=None):
    """``mitsuba.ad.reparameterize_ray`` (reparam.py:336-420) for a scene made of this heightfield: returns
    ``(direction, det)`` = ``(ray.d, 1)`` -- the reparameterisation is the identity in primal mode, exactly as in
    the reference (reparam.py:139-155) -- whose gradients flow into ``shape.heightfield`` and into ``ray.o`` /
    ``ray.d`` (when those require grad) through ``num_rays`` auxiliary rays per ray (von Mises-Fisher around
    ``ray.d``, harmonic weights from ``si.boundary_test``, hits followed with ``RayFlags.FollowShape``).
    ``ray.d`` must be unit length.  PCG32 is replaced by sample_tea_32 keyed on (seed, pair, ray id) (include/hf.h);
    ``ray_index`` (int32/uint32 device tensor, one id per ray, e.g. the global pixel index) makes the samples of a
    ray independent of its position in the batch, so a partitioned render draws the same auxiliary rays as the
    unpartitioned one.  Without it the id is the position in the batch."""
    if ray_index is not None:
        if ray_index.dtype not in (torch.int32, torch.uint32) or ray_index.numel() != ray.o.shape[1]:
            raise ValueError("ray_index must be an int32/uint32 tensor with one id per ray")
        if ray_index.device != ray.o.device:
            raise ValueError("ray_index must live on the rays' device")
        ray_index = ray_index.contiguous()
    return _ReparameterizeOp.apply(shape.heightfield, ray.o, ray.d, shape, num_rays, kappa, exponent, antithetic, seed,
                                   active, ray_index)
