"""CPU-side checks of the boundary: libhf.so loads and exports every symbol include/hf.h
declares (no compute without a GPU), the host mirror's enums/types, the error path without a
device, and the synthetic workload generators."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "hf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hf_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import hf_amd
    from hf_amd import _capi
    assert os.path.exists(hf_amd.build.LIB_PATH), "libhf.so not built (run python __graft_entry__.py)"
    lib = C.CDLL(hf_amd.build.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/hf.h but not exported by libhf.so"
    assert sorted(_capi.SYMBOLS) == declared, "python binding table out of sync with include/hf.h"
    assert _capi.lib().hf_version() == 4   # HF_VERSION: 4 since hf_set_ray_coherence (3: hf_adam_step_scheduled; 2: hf_reparam_* take ray_id)


def test_rayflags_match_reference_values():
    """include/mitsuba/render/interaction.h:19-69"""
    import hf_amd
    F = hf_amd.RayFlags
    assert (F.Minimal, F.UV, F.dPdUV, F.ShadingFrame, F.dNGdUV, F.dNSdUV) == (1, 2, 4, 8, 16, 32)
    assert (F.BoundaryTest, F.FollowShape, F.DetachShape) == (0x40, 0x80, 0x100)
    assert F.All == F.UV | F.dPdUV | F.ShadingFrame and F.AllNonDifferentiable == F.All | F.DetachShape
    hdr = open(os.path.join(ROOT, "include", "hf.h")).read()
    for name, val in (("HF_RAY_BOUNDARYTEST", 0x40), ("HF_RAY_FOLLOWSHAPE", 0x80), ("HF_RAY_DETACHSHAPE", 0x100)):
        assert re.search(rf"{name}\s*=\s*{hex(val)}", hdr)


def test_invert_affine_host_function():
    from hf_amd import _capi
    import common
    m = common.affine(7)
    out = (C.c_float * 12)()
    src = (C.c_float * 12)(*m.reshape(-1).tolist())
    assert _capi.lib().hf_invert_affine(src, out) == 0
    inv = np.array(list(out), np.float64).reshape(3, 4)
    A = np.eye(4); A[:3] = m
    B = np.eye(4); B[:3] = inv
    assert np.allclose(A @ B, np.eye(4), atol=1e-6)
    sing = (C.c_float * 12)(*([0.0] * 12))
    assert _capi.lib().hf_invert_affine(sing, out) == _capi.HF_EINVAL
    assert b"singular" in _capi.lib().hf_last_error_string()


def test_ray_container_semantics():
    import hf_amd
    r = hf_amd.Ray3f(torch.zeros(3, 5), torch.ones(3, 5))
    assert len(r) == 5 and torch.isinf(r.maxt).all()          # maxt defaults to +inf (ray.h:37)
    assert torch.equal(r(torch.full((5,), 2.0)), torch.full((3, 5), 2.0))
    r1 = hf_amd.Ray3f(torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.0, 0.0, -1.0]), 3.0)
    assert len(r1) == 1 and float(r1.maxt[0]) == 3.0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device error path")
def test_shape_fails_loudly_without_device():
    import hf_amd
    with pytest.raises(RuntimeError, match="no HIP device"):
        hf_amd.Heightfield(heightfield=torch.zeros(4, 4))


def test_workload_generators():
    import hf_amd
    from oracle import hf_oracle as O
    h = hf_amd.workload.sine_heights(64, 64).numpy()
    assert np.allclose(h, O.make_sine_heights(64, 64, 4.0, 4.0), atol=1e-6)
    assert np.allclose(hf_amd.workload.sine_heights(4096, 16)[0, :8].numpy(),
                       O.make_sine_heights(4096, 16, 32.0, 32.0)[0, :8], atol=1e-6)
    r = hf_amd.workload.ortho_rays(128, 128, 1, "cpu").numpy()
    assert r.shape == (7, 128 * 128)
    d = r[3:6]
    assert np.allclose(np.linalg.norm(d, axis=0), 1, atol=1e-6) and np.allclose(d, d[:, :1])   # orthographic
    want = np.array([0, 0, 0.125]) - np.array([1.5, 1.5, 1.5]); want /= np.linalg.norm(want)
    assert np.allclose(d[:, 0], want, atol=1e-6)
    assert np.allclose(r[6], 1e4 - 1e-2)                                                        # far - near
    # chunked generation == one shot, and sub-ranges are consistent (rank sharding)
    a = hf_amd.workload.ortho_rays(64, 64, 4, "cpu", chunk=1000)
    b = hf_amd.workload.ortho_rays(64, 64, 4, "cpu")
    assert torch.equal(a, b)
    c = hf_amd.workload.ortho_rays(64, 64, 4, "cpu", start=5000, count=300)
    assert torch.equal(c, b[:, 5000:5300])
    # the 4 samples of a pixel stay inside that pixel's footprint: origins differ by < one pixel step
    o = b[0:3].reshape(3, -1, 4)
    pixel = 3.2 / 64          # film covers 2 * scale(1.6) world units over 64 pixels
    assert ((o - o[:, :, :1]).norm(dim=0) < 1.4143 * pixel).all()
