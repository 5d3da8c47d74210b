"""CPU: the bf16-emulation hooks of the oracle (test infrastructure for the reduced-precision GPU tests) and the ray
generators of the non-lego BASELINE configs."""
import numpy as np
import pytest

from oracle import oracle_np as O


def test_bf16_round_is_round_to_nearest_even():
    torch = pytest.importorskip("torch")
    r = np.random.RandomState(0)
    a = np.concatenate([r.standard_normal(50000).astype(np.float32) * 10.0 ** r.randint(-20, 20, 50000),
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.3895314e38, np.inf, -np.inf], np.float32)])
    ref = torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(O.bf16_round(a), ref)
    assert np.isnan(O.bf16_round(np.array([np.nan], np.float32))).all()


def test_bf16_emulated_forward_and_backward_close_to_fp32():
    p = O.init_params(0, True)
    r = np.random.RandomState(1)
    pts = r.uniform(-2, 2, (512, 3)).astype(np.float32)
    x = np.concatenate([O.embedding(pts, 10), O.embedding(r.standard_normal((512, 3)).astype(np.float32), 4)], 1)
    c32, c16 = {}, {}
    o32 = O.nerf_forward(p, x, cache=c32)
    with O.bf16_operands():
        o16 = O.nerf_forward(p, x, cache=c16)
    assert not np.array_equal(o32, o16)
    assert np.abs(o32 - o16).max() <= 3e-2 * np.abs(o32).max()
    assert np.array_equal(O.nerf_forward(p, x), o32)                      # the hook is restored on exit
    g = r.standard_normal((512, 4)).astype(np.float32)
    g32 = O.nerf_backward(p, c32, g)
    g16 = O.nerf_backward(p, c32, g, operand_round=O.bf16_round)
    for k, v in g32.items():
        e = np.linalg.norm(g16[k] - v) / np.linalg.norm(v)
        assert 0 < e < 2e-2, (k, e)


def test_bf16x3_split_emulation_is_fp32_level():
    """the 3-term split (hi.hi + lo.hi + hi.lo, fp32 accumulate) of csrc/sn_mlp_fwd_bf16x3.hip restated in the oracle: norm-wise
    error of the whole MLP ~1e-5 against the fp32 arithmetic -- three orders below one bf16 product, inside the 2e-4 MLP bar"""
    p = O.init_params(0, True)
    r = np.random.RandomState(2)
    pts = r.uniform(-3, 3, (1024, 3)).astype(np.float32)
    x = np.concatenate([O.embedding(pts, 10), O.embedding(r.standard_normal((1024, 3)).astype(np.float32), 4)], 1)
    o32 = O.nerf_forward(p, x)
    with O.bf16x3_operands():
        o3 = O.nerf_forward(p, x)
    with O.bf16_operands():
        o1 = O.nerf_forward(p, x)
    assert np.array_equal(O.nerf_forward(p, x), o32)                      # hooks restored
    n3 = np.linalg.norm(o3.astype(np.float64) - o32) / np.linalg.norm(o32)
    n1 = np.linalg.norm(o1.astype(np.float64) - o32) / np.linalg.norm(o32)
    assert 0 < n3 < 3e-5 and n1 > 100 * n3, (n3, n1)
    assert (np.abs(o3 - o32) / (np.abs(o32) + 1e-3)).max() <= 2e-4
    hi, lo = O.bf16_split(np.array([1.2345678, -3.3e-5, 0.0], np.float32))
    assert np.array_equal(O.bf16_round(hi), hi) and np.array_equal(O.bf16_round(lo), lo)
    assert np.abs((hi + lo) - np.array([1.2345678, -3.3e-5, 0.0], np.float32)).max() <= 2.0 ** -16 * 1.3


def test_bf16x3_backward_emulation_is_fp32_level_and_distinct():
    """``nerf_backward(operand_round="bf16x3")``: the split backward of csrc/sn_mlp_bwd_bf16x3.hip + sn_dw.hip modes 5-7 restated
    (every operand as its (hi, lo) pair, the lo.lo term dropped).  Same cache, same upstream: within 2e-5 norm-wise of the wide
    backward on every tensor (one bf16 product: ~5e-3) and NOT identical to it; stored pre-activation gradients (gy_out) are the
    decoded (hi + lo) values, 2^-16 relative from the wide ones; bias gradients are the column sums of those."""
    p = O.init_params(0, True)
    r = np.random.RandomState(1)
    pts = r.uniform(-2, 2, (384, 3)).astype(np.float32)
    x = np.concatenate([O.embedding(pts, 10), O.embedding(r.standard_normal((384, 3)).astype(np.float32), 4)], 1)
    c = {}
    O.nerf_forward(p, x, cache=c)
    g = r.standard_normal((384, 4)).astype(np.float32)
    gy8, gy3 = {}, {}
    wide = O.nerf_backward(p, c, g, gy_out=gy8)
    # the wide backward with the SAME softplus-derivative form (1 - exp(-d)) the low-precision paths use: isolates the split's error
    x3 = O.nerf_backward(p, c, g, gy_out=gy3, operand_round="bf16x3")
    b16 = O.nerf_backward(p, c, g, operand_round=O.bf16_round)
    for k, v in wide.items():
        e3 = np.linalg.norm(x3[k] - v) / np.linalg.norm(v)
        e16 = np.linalg.norm(b16[k] - v) / np.linalg.norm(v)
        assert 0 < e3 <= 2e-5, (k, e3)
        assert e3 < 0.02 * e16, (k, e3, e16)
    for k in ("l1", "l4", "l5", "l8", "final"):
        a, b = gy3[k], gy8[k]
        hi, lo = O.bf16_split(a.astype(np.float32))
        assert np.array_equal((hi.astype(np.float64) + lo), a), k          # what is stored decodes to itself
        assert np.abs(a - b).max() <= 3e-5 * np.abs(b).max(), k
        assert np.array_equal(a == 0, b == 0), k                             # same ReLU masks: same cache
    assert np.allclose(x3["xyz_encoding_3.0.bias"], gy3["l3"].sum(0), rtol=0, atol=1e-12)


def test_patch_ray_generators_match_config_sizes():
    ll, dt = O.llff_patch_rays(0), O.dtu_patch_rays(0)
    assert ll.shape == (5292, 8) and dt.shape == (3920, 8)                # BASELINE configs 3, 4 (SURVEY §8a sizes)
    assert np.allclose(ll[:, 6], 1.2) and np.allclose(ll[:, 7], 8.0)
    assert np.allclose(dt[:, 6], 2.125) and np.allclose(dt[:, 7], 4.525)
    full = O.get_rays(8, 10, 7.0, np.concatenate([np.eye(3), np.zeros((3, 1))], 1), 1.0, 2.0).reshape(8, 10, 8)
    pr = O.patch_rays(8, 10, 7.0, np.concatenate([np.eye(3), np.zeros((3, 1))], 1), 1.0, 2.0, 1, 2, 3, 2, 3, 2)
    assert np.array_equal(pr.reshape(2, 3, 8), full[2:6:2, 1:10:3])
