"""ctypes binding of include/hf.h (the C ABI of libhf.so).  No torch types cross
this boundary: device pointers are passed as integers (tensor.data_ptr())."""
import ctypes as C
import os

from . import build as _build

HF_OK, HF_EINVAL, HF_EDEVICE, HF_ENOMEM, HF_EFLAGS = 0, 1, 2, 3, 4

_fp = C.c_void_p  # device pointers


class hf_desc_t(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_height", C.c_float),
                ("to_world", C.c_float * 12), ("to_object", C.c_float * 12),
                ("has_to_object", C.c_int32), ("flip_normals", C.c_int32), ("device", C.c_int32)]


class hf_rays_t(C.Structure):
    _fields_ = [("o", _fp * 3), ("d", _fp * 3), ("maxt", _fp)]


class hf_pi_t(C.Structure):
    _fields_ = [("t", _fp), ("prim_uv", _fp * 2), ("prim_index", _fp)]


class hf_si_t(C.Structure):
    _fields_ = [("t", _fp), ("p", _fp * 3), ("n", _fp * 3), ("uv", _fp * 2), ("sh_n", _fp * 3),
                ("dp_du", _fp * 3), ("dp_dv", _fp * 3), ("boundary_test", _fp),
                ("sh_s", _fp * 3), ("sh_t", _fp * 3), ("wi", _fp * 3)]


class hf_si_grad_t(C.Structure):
    _fields_ = [("t", _fp), ("p", _fp * 3), ("n", _fp * 3), ("uv", _fp * 2), ("sh_n", _fp * 3),
                ("dp_du", _fp * 3), ("dp_dv", _fp * 3)]


# every symbol include/hf.h declares: name -> (restype, argtypes)
HF_MAX_LIGHTS = 8


class hf_dir_light_t(C.Structure):
    _fields_ = [("to_light", C.c_float * 3), ("irradiance", C.c_float)]


SYMBOLS = {
    "hf_create": (C.c_int, [C.POINTER(hf_desc_t), C.POINTER(C.c_void_p)]),
    "hf_destroy": (C.c_int, [C.c_void_p]),
    "hf_capture_reset": (C.c_int, [C.c_void_p]),
    "hf_set_ray_coherence": (C.c_int, [C.c_void_p, C.c_int]),
    "hf_get_ray_coherence": (C.c_int, [C.c_void_p]),
    "hf_set_heights": (C.c_int, [C.c_void_p, _fp, C.c_void_p]),
    "hf_set_heights_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hf_adam_step": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, C.c_double, C.c_double, C.c_double, C.c_double,
                               C.c_uint32, C.c_int, C.c_void_p]),
    "hf_adam_lr_t": (C.c_float, [C.c_double, C.c_double, C.c_double, C.c_uint32]),
    "hf_adam_step_scheduled": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, _fp, _fp, C.c_double, C.c_double, C.c_double, C.c_int,
                                         C.c_void_p]),
    "hf_set_transform": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "hf_bbox": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "hf_heights_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "hf_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "hf_ray_intersect_preliminary": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(hf_rays_t), _fp,
                                               C.POINTER(hf_pi_t), C.c_void_p]),
    "hf_ray_test": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(hf_rays_t), _fp, _fp, C.c_void_p]),
    "hf_compute_surface_interaction": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(hf_rays_t),
                                                 C.POINTER(hf_pi_t), C.c_uint32, _fp,
                                                 C.POINTER(hf_si_t), C.c_void_p]),
    "hf_ray_intersect": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(hf_rays_t), C.c_uint32, _fp,
                                   C.POINTER(hf_pi_t), C.POINTER(hf_si_t), C.c_void_p]),
    "hf_adjoint": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(hf_rays_t), C.POINTER(hf_pi_t),
                             C.c_uint32, _fp, C.POINTER(hf_si_grad_t), _fp,
                             C.POINTER(_fp * 3), C.POINTER(_fp * 3), C.c_void_p]),
    "hf_adjoint_rows": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(hf_rays_t), C.POINTER(hf_pi_t),
                             C.c_uint32, _fp, C.POINTER(hf_si_grad_t), _fp,
                             C.POINTER(_fp * 3), C.POINTER(_fp * 3), C.c_void_p, C.c_void_p]),
    "hf_direct_lighting": (C.c_int, [C.c_size_t, C.c_uint32, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp, C.c_uint32,
                                     C.POINTER(hf_dir_light_t), C.c_float, C.POINTER(_fp), _fp, C.c_void_p]),
    "hf_direct_lighting_adjoint": (C.c_int, [C.c_size_t, C.c_uint32, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp,
                                             C.c_uint32, C.POINTER(hf_dir_light_t), C.c_float, C.POINTER(_fp), _fp,
                                             C.POINTER(_fp * 3), C.c_void_p]),
    "hf_direct_lighting_weighted": (C.c_int, [C.c_size_t, C.c_uint32, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp, _fp,
                                              C.c_uint32, C.POINTER(hf_dir_light_t), C.c_float, C.POINTER(_fp), _fp,
                                              C.c_void_p]),
    "hf_direct_lighting_weighted_adjoint": (C.c_int, [C.c_size_t, C.c_uint32, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp,
                                                      _fp, C.c_uint32, C.POINTER(hf_dir_light_t), C.c_float,
                                                      C.POINTER(_fp), _fp, C.POINTER(_fp * 3), _fp, C.c_void_p]),
    "hf_point_lighting": (C.c_int, [C.c_size_t, C.c_uint32, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp, C.POINTER(_fp * 3),
                                    C.c_uint32, C.POINTER(hf_dir_light_t), C.c_float, C.POINTER(_fp), _fp, C.c_void_p]),
    "hf_point_lighting_adjoint": (C.c_int, [C.c_size_t, C.c_uint32, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp,
                                            C.POINTER(_fp * 3), C.c_uint32, C.POINTER(hf_dir_light_t), C.c_float,
                                            C.POINTER(_fp), _fp, C.POINTER(_fp * 3), C.POINTER(_fp * 3), C.c_void_p]),
    "hf_film_splat": (C.c_int, [C.c_size_t, C.c_uint32, C.POINTER(_fp), _fp, _fp, C.c_uint32, C.c_uint32, C.c_float, _fp, _fp,
                                C.c_void_p]),
    "hf_film_splat_adjoint": (C.c_int, [C.c_size_t, C.c_uint32, _fp, _fp, C.c_uint32, C.c_uint32, C.c_float, _fp,
                                        C.POINTER(_fp), C.c_void_p]),
    "hf_reparam_aux_rays": (C.c_int, [C.c_size_t, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp, C.c_uint32, C.c_float,
                                      C.c_int, C.c_uint32, C.c_void_p, C.POINTER(_fp * 3), _fp, C.c_void_p]),
    "hf_reparam_weights": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp, C.c_uint32,
                                     C.c_float, C.c_float, C.c_int, C.c_uint32, C.c_void_p, _fp, C.POINTER(_fp * 3), _fp, _fp,
                                     C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp, C.POINTER(_fp * 3), _fp,
                                     C.POINTER(_fp * 3), C.c_void_p]),
    "hf_reparam_trace": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp, C.c_uint32, C.c_float,
                                   C.c_int, C.c_uint32, C.c_void_p, C.POINTER(hf_pi_t), C.POINTER(hf_si_t), C.c_void_p]),
    "hf_reparam_trace_all": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp, C.c_uint32, C.c_float,
                                       C.c_int, C.c_uint32, C.c_void_p, C.POINTER(hf_pi_t), C.POINTER(hf_si_t), C.c_size_t, C.c_void_p]),
    "hf_reparam_backward": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(_fp * 3), C.POINTER(_fp * 3), _fp, C.c_uint32,
                                      C.c_float, C.c_float, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, _fp, C.c_size_t,
                                      C.POINTER(_fp * 3), _fp, _fp, C.c_void_p]),
    "hf_ray_intersect_preliminary_packet": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(_fp * 3), C.POINTER(_fp * 3),
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_fp * 2), C.c_void_p]),
    "hf_ray_test_packet": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(_fp * 3), C.POINTER(_fp * 3), C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "hf_allreduce_grad": (C.c_int, [_fp, C.c_size_t, C.c_void_p, C.c_void_p]),
    "hf_num_levels": (C.c_int, [C.c_void_p]),
    "hf_get_mip": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint32),
                             C.POINTER(C.c_uint32)]),
    "hf_invert_affine": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "hf_last_error_string": (C.c_char_p, []),
    "hf_version": (C.c_int, []),
}

_lib = None


class HfError(RuntimeError):
    """Raised for a non-zero status from libhf (the adapter's Throw(...))."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def lib():
    """Load libhf.so; fails loudly if it is missing -- there is no fallback path."""
    global _lib
    if _lib is None:
        # HF_LIB: another build of the SAME library (scripts/vb.sh variants for A/B measurements), never a fallback
        path = os.environ.get("HF_LIB") or _build.LIB_PATH
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
                "This package has no CPU/eager fallback.")
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != HF_OK:
        raise HfError(rc, lib().hf_last_error_string().decode())
