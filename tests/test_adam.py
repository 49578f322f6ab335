"""Adam step on the height texture (optimiser + params.update on the device).
CPU: the oracle restatement against the closed forms that follow from optimizers.py:263-300.
GPU: hf_adam_step (through the C ABI / the host mirror hf_amd.Adam) bit for bit against the oracle,
and the rebuilt acceleration data is what the next trace sees."""
import numpy as np
import pytest


def test_oracle_first_step_is_lr_sign_g(oracle):
    rng = np.random.default_rng(0)
    h = rng.uniform(0, 1, (7, 9)).astype(np.float32)
    g = rng.normal(size=(7, 9)).astype(np.float32)
    hn, m, v = oracle.adam_step(h, g, np.zeros_like(h), np.zeros_like(h), lr=0.01, step=1)
    # m = (1-b1) g, v = (1-b2) g^2, lr_t = lr sqrt(1-b2)/(1-b1)  =>  step = lr g / (|g| + eps/sqrt(1-b2))
    assert np.allclose(m, np.float32(0.1) * g, rtol=1e-6) and np.allclose(v, np.float32(0.001) * (g * g), rtol=1e-6)
    assert np.allclose(h - hn, 0.01 * np.sign(g), atol=1e-6)


def test_oracle_constant_gradient_walks_lr_per_step(oracle):
    h = np.full((3, 3), 0.5, np.float32); g = np.full((3, 3), -0.25, np.float32)
    m = np.zeros_like(h); v = np.zeros_like(h)
    for t in range(1, 21):
        h, m, v = oracle.adam_step(h, g, m, v, lr=0.02, step=t)
    assert np.allclose(h, 0.5 + 20 * 0.02, atol=1e-5)


def test_oracle_mask_updates_freezes_unobserved_texels(oracle):
    h = np.full((2, 4), 0.5, np.float32); g = np.array([[0, 1, 0, -1], [2, 0, 0, 0]], np.float32)
    m0 = np.full_like(h, 0.3); v0 = np.full_like(h, 0.2)
    hn, m, v = oracle.adam_step(h, g, m0, v0, lr=0.1, step=3, mask_updates=True)
    z = g == 0
    assert np.array_equal(hn[z], h[z]) and np.array_equal(m[z], m0[z]) and np.array_equal(v[z], v0[z])
    assert np.all(hn[~z] != h[~z])
    hn2, m2, v2 = oracle.adam_step(h, g, m0, v0, lr=0.1, step=3, mask_updates=False)
    assert np.all(hn2[z] != h[z])  # without the mask the stale momentum keeps moving them


def test_oracle_uniform_adam_divides_by_the_largest_second_moment(oracle):
    """optimizers.py:259, 290-291: step = lr_t m_t / (sqrt(max(v_t)) + eps): on the first step every entry moves by
    lr g / max|g| (up to eps), the largest by lr"""
    rng = np.random.default_rng(2)
    h = rng.uniform(0, 1, (5, 6)).astype(np.float32)
    g = rng.normal(size=(5, 6)).astype(np.float32)
    hn, m, v = oracle.adam_step(h, g, np.zeros_like(h), np.zeros_like(h), lr=0.01, step=1, uniform=True)
    assert np.allclose(h - hn, 0.01 * g / np.abs(g).max(), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("mask_updates,uniform", [(False, False), (True, False), (False, True), (True, True)])
def test_gpu_adam_matches_oracle_bit_for_bit(hf, oracle, mask_updates, uniform):
    import torch
    rng = np.random.default_rng(5)
    H, W = 37, 53
    h0 = rng.uniform(0.2, 0.8, (H, W)).astype(np.float32)
    shape = hf.Heightfield(heightfield=torch.from_numpy(h0).cuda(), max_height=0.5)
    opt = hf.Adam(shape, lr=0.03, beta_1=0.9, beta_2=0.99, epsilon=1e-8, mask_updates=mask_updates, uniform=uniform)
    h, m, v = h0.copy(), np.zeros_like(h0), np.zeros_like(h0)
    o = torch.tensor([[0.1], [-0.2], [2.0]], device="cuda"); d = torch.tensor([[0.0], [0.0], [-1.0]], device="cuda")
    t_prev = None
    for step in range(1, 6):
        g = rng.normal(size=(H, W)).astype(np.float32)
        g[rng.uniform(size=(H, W)) < 0.3] = 0.0
        shape.heightfield.grad = torch.from_numpy(g).cuda()
        opt.step()
        h, m, v = oracle.adam_step(h, g, m, v, 0.03, 0.9, 0.99, 1e-8, step, mask_updates, uniform)
        assert np.array_equal(shape.heightfield.detach().cpu().numpy(), h), f"heights differ at step {step}"
        assert np.array_equal(opt.state[0].cpu().numpy(), m) and np.array_equal(opt.state[1].cpu().numpy(), v)
        # the step also rebuilt the acceleration data: the trace sees the new surface
        f_o = oracle.OracleField(h, max_height=0.5)
        r = np.array([[0.1], [-0.2], [2.0], [0.0], [0.0], [-1.0], [np.inf]], np.float32)
        t_o = f_o.ray_intersect_preliminary(r)[0]
        t_g = shape.ray_intersect_preliminary(hf.Ray3f(o, d)).t.cpu().numpy()
        assert np.array_equal(t_o, t_g)
        assert t_prev is None or t_g[0] != t_prev
        t_prev = t_g[0]
        bb = shape.bbox()
        assert np.isclose(float(bb[1, 2]), 0.5 * h.max(), rtol=1e-6) and np.isclose(float(bb[0, 2]), 0.5 * h.min(), rtol=1e-6)


@pytest.mark.gpu
def test_gpu_adam_argument_errors(hf):
    import torch
    shape = hf.Heightfield(heightfield=torch.full((4, 4), 0.5).cuda(), max_height=1.0)
    with pytest.raises(AssertionError):
        hf.Adam(shape, lr=0.1, beta_1=1.0)
    opt = hf.Adam(shape, lr=0.1)
    opt.step()  # no gradient yet: a no-op like optimizers.py:274-275
    assert opt.t == 0
    shape.heightfield.grad = torch.ones(4, 4).cuda()
    opt.beta_2 = 1.5
    with pytest.raises(hf.HfError):
        opt.step()
