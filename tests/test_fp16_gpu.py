"""GPU: compute_dtype='fp16' (SN_DTYPE_F16, round 6) -- the bf16 inference kernels' instruction streams with fp16 operands
(v_cvt_pk_f16_f32 / v_mfma_f32_32x32x16_f16: 11 significand bits instead of 8 at the same matrix rate).  VERDICT r5 #7 asked for the oracle
experiment first: oracle_np.fp16_operands() is 8-17x tighter than bf16_operands() on the trained-weight fixtures (tests/test_trained_weights_cpu.py,
tools/precision_probe.py), which is the bar it set for building the kernels.  Held here to (a) the fp16-EMULATED oracle (same operand roundings),
(b) the PSNR bar of the reduced-precision modes, (c) "at least 4x closer to the reference than bf16" on the trained student.  Inference only."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import oracle_np as O                                                                   # noqa: E402
from tests.helpers import load_case                                                                 # noqa: E402
from tests.test_parity_gpu import dev, embeddings, make_model                                       # noqa: E402
from tests.test_trained_weights_cpu import err_over_bound                                           # noqa: E402
from tests.test_trained_weights_gpu import record, render_case                                      # noqa: E402


def test_fp16_mlp_vs_fp16_emulated_oracle():
    from sinnerf_amd import rendering
    model, p = make_model(0, True, dtype="fp16")
    rays = O.lego_rays(400, 400, seed=0)[::1601][:100]
    n = rays.shape[0]
    z = O.coarse_z_vals(rays, 70, False, 1.0, np.random.RandomState(1).uniform(0, 1, (n, 70)).astype(np.float32))
    with O.fp16_operands():                       # both Linear operands rounded to fp16 (RNE), fp32 heads, as in the kernel
        ref16 = O._run_model(p, rays, z, O.embedding(rays[:, 3:6], 4), False, 1 << 20)
    ref32 = O._run_model(p, rays, z, O.embedding(rays[:, 3:6], 4), False, 1 << 20)
    with O.bf16_operands():
        refb = O._run_model(p, rays, z, O.embedding(rays[:, 3:6], 4), False, 1 << 20)
    for sigma_only in (False, True):
        with torch.no_grad():
            got = rendering._mlp(model, torch.from_numpy(rays).to(dev()), torch.from_numpy(z).to(dev()), sigma_only).cpu().numpy()
        r16 = ref16[..., 3] if sigma_only else ref16
        r32 = ref32[..., 3] if sigma_only else ref32
        rb = refb[..., 3] if sigma_only else refb
        assert got.shape == r16.shape and np.isfinite(got).all()
        scale = np.abs(r32).max()
        e16, e32, eb = np.abs(got - r16).max() / scale, np.abs(got - r32).max() / scale, np.abs(rb - r32).max() / scale
        print("fp16 MLP (sigma_only=%s): vs fp16-emulated oracle %.2e, vs fp32 oracle %.2e (bf16 emulation vs fp32: %.2e)" % (sigma_only, e16, e32, eb))
        assert e16 <= 2.5e-4, e16                 # same arithmetic (the bf16 kernel against ITS emulation: 2e-3)
        assert e32 <= 4e-3 and e32 < eb / 3, (e32, eb)


def test_fp16_nerf_forward_embedded_rows():
    """NeRF.forward(x) / sigma_only (sn_mlp_forward_embedded) under no_grad against the fp16-emulated oracle on the golden rows"""
    from tests.helpers import GOLDEN
    z = np.load(f"{GOLDEN}/nerf_mlp.npz")
    model, p = make_model(int(z["seed"]), bool(z["teacher"]), dtype="fp16")
    x_np = np.concatenate([z["emb_xyz"], z["emb_dir"]], 1)
    x = torch.from_numpy(x_np).to(dev())
    with torch.no_grad():
        full = model(x).cpu().numpy()
        sig = model(x[:, :63].contiguous(), sigma_only=True).cpu().numpy()
    with O.fp16_operands():
        ref = O.nerf_forward(p, x_np)
    assert full.shape == (300, 4) and sig.shape == (300, 1)
    e = (np.abs(full - ref) / (np.abs(ref) + 1e-2)).max()
    print("fp16 NeRF.forward vs fp16-emulated oracle: %.2e" % e)
    assert e <= 1e-3
    assert (np.abs(full - z["out_full"]) / (np.abs(z["out_full"]) + 1e-2)).max() <= 2e-2
    assert np.abs(sig - full[:, 3:]).max() <= 1e-5 * np.abs(sig).max()


@pytest.mark.parametrize("name", ["render_trained_lego_eval", "render_trained_llff_eval_128", "render_trained_lego_train"])
def test_fp16_render_on_trained_weights(name):
    got, ref = render_case(name, "fp16")
    gotb, _ = render_case(name, "bf16")
    gt = ref["rgb_fine"] + np.random.RandomState(0).normal(0, 0.02, ref["rgb_fine"].shape).astype(np.float32)
    d = O.psnr(got["rgb_fine"], gt) - O.psnr(ref["rgb_fine"], gt)
    e, eb = err_over_bound(got, ref), err_over_bound(gotb, ref)
    print(f"{name} [fp16]: dPSNR = {d:+.5f} dB, err / fp32 bound = {e:.4f} (bf16: {eb:.4f}, x{eb / e:.1f})")
    record(f"{name}:fp16:dpsnr_db", d)
    record(f"{name}:fp16:err_over_bound", e)
    assert all(np.isfinite(v).all() for v in got.values())
    assert abs(d) <= 0.05
    assert e < 1.0 and e < eb / 4, (e, eb)        # inside the FP32 bar on this student, and >= 4x closer than bf16 (oracle: 8-17x)


def test_fp16_is_inference_only():
    import sinnerf_amd
    mc, _ = make_model(0, True, dtype="fp16")
    mf, _ = make_model(1, True, dtype="fp16")
    mc.train(); mf.train()
    rays = torch.from_numpy(O.lego_rays(400, 400, seed=0)[::5000]).to(dev())
    with pytest.raises(NotImplementedError, match="INFERENCE"):
        sinnerf_amd.render_rays([mc, mf], embeddings(), rays, 64, False, 1.0, 1.0, 64, 32768, True)
    with pytest.raises(NotImplementedError, match="INFERENCE"):
        mc(torch.randn(8, 90, device=dev()))
    with torch.no_grad():
        r = sinnerf_amd.render_rays([mc, mf], embeddings(), rays, 64, False, 0, 0, 64, 32768, True)
    assert torch.isfinite(r["rgb_fine"]).all()
