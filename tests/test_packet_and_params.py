"""Rows a3 and a7 of SURVEY.md section 8.

a3  Shape::ray_intersect_preliminary_scalar / _packet(4/8/16) and ray_test_* (include/mitsuba/render/shape.h:
    220-240; called per kd-tree leaf, kdtree.h:2490-2520): the host-pointer packet entry of the C ABI
    (hf_ray_intersect_preliminary_packet / hf_ray_test_packet) at n in {1, 4, 8, 16} against the oracle's brute
    force, bit for bit, plus the wavefront entry at the same sizes.
a7  traverse / parameters_changed / parameters_grad_enabled / bbox / mark_dirty (src/render/shape.cpp:536-570,
    src/shapes/rectangle.cpp:114-142): the host mirror's plumbing, each followed by a trace against a FRESH
    oracle built from the new parameters.
"""
import numpy as np
import pytest
import torch

import common

pytestmark = pytest.mark.gpu


def _scene(hf, oracle, W=37, H=23, seed=5, to_world=None, max_height=0.6):
    rng = np.random.default_rng(seed)
    h = common.heights("sine", W, H, rng)
    f_o = oracle.OracleField(h, max_height=max_height, to_world=to_world)
    props = dict(heightfield=torch.from_numpy(h), max_height=max_height)
    if to_world is not None:
        props["to_world"] = torch.from_numpy(np.asarray(to_world))
    return rng, h, f_o, hf.Heightfield(props)


@pytest.mark.parametrize("n", [1, 4, 8, 16])
def test_packet_entry_matches_brute_force(hf, oracle, n):
    tw = common.affine(11)
    rng, h, f_o, f_g = _scene(hf, oracle, to_world=tw)
    hits = 0
    for rep in range(12):
        r = common.to_world_rays(np.concatenate([common.random_rays(n - n // 2, rng, 0.6), common.inside_rays(n // 2, rng, 0.6)], 1), tw)
        t, u, v, prim = f_o.ray_intersect_preliminary(r, naive=True)
        tg, uvg, pg = f_g.ray_intersect_preliminary_packet(r[0:3], r[3:6], r[6])
        assert np.array_equal(t, tg) and np.array_equal(prim, pg), (n, rep)
        assert np.array_equal(u, uvg[0]) and np.array_equal(v, uvg[1])
        assert np.array_equal(f_g.ray_test_packet(r[0:3], r[3:6], r[6]), np.isfinite(t))
        # the wavefront entry at the same small n
        rt = torch.from_numpy(r).cuda()
        pi = f_g.ray_intersect_preliminary(hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous()))
        assert np.array_equal(pi.t.cpu().numpy(), t) and np.array_equal(pi.prim_index.cpu().numpy().view(np.uint32), prim)
        hits += int(np.isfinite(t).sum())
        # active mask: inactive lanes are misses
        act = rng.uniform(size=n) < 0.5
        ta, _, pa = f_g.ray_intersect_preliminary_packet(r[0:3], r[3:6], r[6], active=act)
        assert np.array_equal(ta[act], t[act]) and np.all(np.isinf(ta[~act])) and np.all(pa[~act] == 0)
    assert hits > 0


def test_scalar_entry_and_errors(hf, oracle):
    rng, h, f_o, f_g = _scene(hf, oracle)
    r = np.array([[0.1], [-0.2], [2.0], [0.0], [0.0], [-1.0], [np.inf]], np.float32)
    t, u, v, prim = f_o.ray_intersect_preliminary(r, naive=True)
    ts, uvs, ps = f_g.ray_intersect_preliminary_scalar(r[0:3, 0], r[3:6, 0])
    assert ts == float(t[0]) and ps == int(prim[0]) and uvs == (float(u[0]), float(v[0]))
    assert f_g.ray_test_scalar(r[0:3, 0], r[3:6, 0]) is True
    assert f_g.ray_test_scalar([0.1, -0.2, 2.0], [0, 0, 1.0]) is False          # pointing away
    assert f_g.ray_test_scalar(r[0:3, 0], r[3:6, 0], maxt=0.5) is False         # maxt in front of the surface
    with pytest.raises(hf.HfError, match="at most 16"):
        f_g.ray_test_packet(np.zeros((3, 17), np.float32), np.ones((3, 17), np.float32))
    # many host threads at once (the kd-tree calls from all render workers, integrator.cpp:161-200)
    import threading
    out = [None] * 8
    def work(k):
        out[k] = f_g.ray_intersect_preliminary_packet(r[0:3], r[3:6], r[6])[0][0]
    th = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    [x.start() for x in th]; [x.join() for x in th]
    assert all(o == t[0] for o in out)


class _Collect:
    def __init__(self):
        self.params = {}

    def put_parameter(self, name, value, flags):
        self.params[name] = (value, flags)


def test_traverse_and_grad_enabled(hf, oracle):
    rng, h, f_o, f_g = _scene(hf, oracle)
    cb = _Collect()
    f_g.traverse(cb)
    assert set(cb.params) == {"heightfield", "max_height", "to_world"}
    val, flags = cb.params["heightfield"]
    assert val is f_g.heightfield and tuple(val.shape) == h.shape
    # heights are Differentiable | Discontinuous like every shape parameter that moves silhouettes (rectangle.cpp:126-129)
    assert flags == hf.ParamFlags.Differentiable | hf.ParamFlags.Discontinuous
    assert cb.params["to_world"][1] == hf.ParamFlags.NonDifferentiable
    assert f_g.parameters_grad_enabled() is False
    f_g.heightfield.requires_grad_(True)
    assert f_g.parameters_grad_enabled() is True


def _check_against(hf, f_o, f_g, r):
    t, u, v, prim = f_o.ray_intersect_preliminary(r)
    rt = torch.from_numpy(r).cuda()
    ray = hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous())
    pi = f_g.ray_intersect_preliminary(ray)
    assert np.array_equal(pi.t.cpu().numpy(), t) and np.array_equal(pi.prim_index.cpu().numpy().view(np.uint32), prim)
    si = f_g.compute_surface_interaction(ray, pi)
    so = f_o.compute_surface_interaction(r, t, u, v, prim)
    hit = np.isfinite(t)
    assert hit.mean() > 0.2
    assert np.allclose(si.p.cpu().numpy()[:, hit], so["p"][:, hit], rtol=1e-5, atol=1e-6)
    assert np.allclose(si.n.cpu().numpy()[:, hit], so["n"][:, hit], rtol=1e-5, atol=1e-6)


def test_parameters_changed_to_world_then_trace(hf, oracle):
    """params['to_world'] = T; params.update() (rectangle.cpp:131-142) -> hf_set_transform: a trace afterwards
    sees the moved shape; checked against an oracle constructed with the new transform."""
    rng, h, f_o, f_g = _scene(hf, oracle)
    assert f_g.dirty()
    r_obj = np.concatenate([common.random_rays(3000, rng, 0.6), common.inside_rays(1000, rng, 0.6)], 1)
    _check_against(hf, f_o, f_g, r_obj.astype(np.float32))
    tw = common.affine(23)
    f_g.to_world = torch.from_numpy(tw)
    f_g._dirty = False
    f_g.parameters_changed(["to_world"])
    assert f_g.dirty()                                                   # mark_dirty (shape.h:540)
    f_new = oracle.OracleField(h, max_height=0.6, to_world=tw)
    _check_against(hf, f_new, f_g, common.to_world_rays(r_obj, tw))
    assert np.allclose(f_g.bbox().reshape(-1).numpy(), f_new.bbox(), rtol=1e-6, atol=1e-6)
    # and the old world-space rays no longer see the old surface
    t_old = f_o.ray_intersect_preliminary(r_obj.astype(np.float32))[0]
    rt = torch.from_numpy(r_obj.astype(np.float32)).cuda()
    t_now = f_g.ray_intersect_preliminary(hf.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous())).t.cpu().numpy()
    assert not np.array_equal(t_old, t_now)


def test_parameters_changed_heights_then_trace(hf, oracle):
    rng, h, f_o, f_g = _scene(hf, oracle)
    h2 = common.heights("rand", h.shape[1], h.shape[0], rng)
    with torch.no_grad():
        f_g.heightfield.copy_(torch.from_numpy(h2))
    f_g.parameters_changed(["heightfield"])
    f_new = oracle.OracleField(h2, max_height=0.6)
    r = np.concatenate([common.random_rays(3000, rng, 0.6), common.inside_rays(1000, rng, 0.6)], 1).astype(np.float32)
    _check_against(hf, f_new, f_g, r)
    assert np.allclose(f_g.bbox().reshape(-1).numpy(), f_new.bbox(), rtol=1e-6, atol=1e-6)
    # resolution may not change (bitmap.cpp:272-286)
    f_g.heightfield = torch.zeros(5, 5).cuda()
    with pytest.raises(RuntimeError, match="tensor shape"):
        f_g.parameters_changed(["heightfield"])


def test_allreduce_grad_through_rccl_single_rank(hf):
    """hf_allreduce_grad driven the way a C++ host would: an ncclComm_t from RCCL's own
    ncclCommInitRank (one rank: the sum is the identity), the texture summed in place on a stream."""
    import ctypes as C
    from hf_amd import _capi
    try:
        rccl = C.CDLL("librccl.so", mode=C.RTLD_GLOBAL)
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so", mode=C.RTLD_GLOBAL)
    class ncclUniqueId(C.Structure):            # passed BY VALUE to ncclCommInitRank (rccl.h)
        _fields_ = [("internal", C.c_byte * 128)]
    uid = ncclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, ncclUniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    g = torch.randn(257, 129, device="cuda")
    want = g.clone()
    st = torch.cuda.current_stream().cuda_stream
    _capi.check(_capi.lib().hf_allreduce_grad(g.data_ptr(), g.numel(), comm, st))
    torch.cuda.synchronize()
    assert torch.equal(g, want)
    assert _capi.lib().hf_allreduce_grad(None, 4, comm, st) == _capi.HF_EINVAL
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
