"""GPU: ``NeRF(use_new_activation=False)`` -- the constructor's default of models/nerf.py:47-50 (ReLU after dir_encoding,
Sigmoid after rgb, :91-100) -- through every MLP entry point of the C ABI (SN_DTYPE_CLASSIC_HEADS): inference forward
(render path and pre-embedded rows), training forward + backward chain + weight gradients, fp32 and bf16 operands; against
the reference's golden vectors (oracle/gen_golden.py --classic-heads) and the oracle inside ``with O.classic_heads():``."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import oracle_np as O                                              # noqa: E402
from tests.helpers import GOLDEN, check_render                                 # noqa: E402
from tests.test_parity_gpu import dev, embeddings, to_np                       # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


def classic_model(seed, teacher, dtype="fp32"):
    import sinnerf_amd
    m = sinnerf_amd.NeRF(compute_dtype=dtype)                                  # use_new_activation=False: nerf.py:50
    assert m.use_new_activation is False
    p = O.init_params(seed, teacher)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    return m.to(dev()).eval(), p


def _golden_grad_errors(z, got):
    errs = {}
    for k, g in got.items():
        ref_norm = float(z["gnorm." + k])
        if g.ndim == 1:
            ref, mine = z["gfull." + k].astype(np.float64), g
        else:
            idx = z["gidx." + k]
            ref, mine = z["gval." + k].astype(np.float64), g.reshape(-1)[idx]
        errs[k] = (float(np.abs(mine - ref).max() / (np.abs(ref).max() + 1e-12)), abs(np.linalg.norm(g) - ref_norm) / ref_norm)
    return errs


def test_classic_heads_forward_and_autograd_vs_reference_golden():
    """fp32: model(x) and model(x).backward(g) against the reference module's own outputs / autograd gradients."""
    z = np.load(f"{GOLDEN}/nerf_mlp_classic_heads.npz")
    model, p = classic_model(int(z["seed"]), bool(z["teacher"]))
    x = torch.from_numpy(z["x"]).to(dev())
    with torch.no_grad():
        out = model(x)
        sig = model(x[:, :63].contiguous(), sigma_only=True)
    assert np.abs(npy(out) - z["out"]).max() <= 2e-5 * np.abs(z["out"]).max()
    assert np.abs(npy(sig) - z["sigma_only"]).max() <= 2e-5 * np.abs(z["sigma_only"]).max()
    assert (npy(out)[:, :3] > 0).all() and (npy(out)[:, :3] < 1).all()    # plain sigmoid: inside (0, 1)
    model.train()
    o = model(x)
    assert torch.equal(o.detach(), out)                                       # training forward == inference forward
    o.backward(torch.from_numpy(z["g"]).to(dev()))
    got = {k: q.grad.detach().cpu().numpy().astype(np.float64) for k, q in model.named_parameters()}
    errs = _golden_grad_errors(z, got)
    bad = {k: e for k, e in errs.items() if e[0] > 2e-4 or e[1] > 5e-5}      # reference = fp32 autograd
    assert not bad, bad
    # ... and the switch is real: the same weights under the SinNeRF heads give different colours and gradients
    import sinnerf_amd
    new = sinnerf_amd.NeRF(use_new_activation=True)
    new.load_state_dict(model.state_dict())
    with torch.no_grad():
        assert (new.to(dev())(x)[:, :3] - out[:, :3]).abs().max() > 1e-2


def test_classic_heads_render_rays_golden_and_bf16_psnr():
    import sinnerf_amd
    r = np.load(f"{GOLDEN}/render_lego_eval_classic_heads.npz")
    ref = {k: r[k] for k in r.files if k not in ("rays", "seeds", "teacher")}
    rays = torch.from_numpy(r["rays"]).to(dev())
    models = [classic_model(int(s), bool(r["teacher"]))[0] for s in r["seeds"]]
    with torch.no_grad():
        res = sinnerf_amd.render_rays(models, embeddings(), rays, 64, False, 0, 0, 64, 32768, True)
    check_render(to_np(res), ref, tag="classic_heads")
    # bf16 operands (hand-scheduled inference kernel, classic pass): PSNR of the fine image against the fp32 reference
    m16 = [classic_model(int(s), bool(r["teacher"]), dtype="bf16")[0] for s in r["seeds"]]
    with torch.no_grad():
        res16 = sinnerf_amd.render_rays(m16, embeddings(), rays, 64, False, 0, 0, 64, 32768, True)
    mse = float(np.mean((to_np(res16)["rgb_fine"].astype(np.float64) - ref["rgb_fine"]) ** 2))
    assert mse < 10 ** (-40 / 10), mse                                        # > 40 dB against the reference image


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_classic_heads_training_path_vs_oracle(dtype):
    """render-path training forward (points from rays, z) + backward chain + weight gradients against the oracle (fp32:
    wide accumulation; bf16: the bf16-emulated backward with the stored activations, as test_bf16_mlp_backward_...)."""
    from sinnerf_amd.autograd import _MLPFn
    model, p = classic_model(3, True, dtype=dtype)
    model.train()
    rays = O.lego_rays(400, 400, seed=0)[::2503][:60]
    n, S = rays.shape[0], 37                                                  # 2220 points: ragged last tile
    zv = O.coarse_z_vals(rays, S, False, 1.0, np.random.RandomState(1).uniform(0, 1, (n, S)).astype(np.float32))
    g = np.random.RandomState(2).standard_normal((n, S, 4)).astype(np.float32)
    rays_t, z_t = torch.from_numpy(rays).to(dev()), torch.from_numpy(zv).to(dev())
    out = _MLPFn.apply(model, rays_t, z_t, *model.raw_tensors())
    (out * torch.from_numpy(g).to(dev())).sum().backward(retain_graph=True)
    got = {k: q.grad.detach().cpu().numpy().astype(np.float64) for k, q in model.named_parameters()}
    xin = np.concatenate([O.embedding(O._points(rays, zv).reshape(-1, 3), 10),
                          np.repeat(O.embedding(rays[:, 3:6], 4), S, 0)], 1)
    cache = {}
    outc = out.detach().cpu().numpy().reshape(-1, 4)
    with O.classic_heads():
        if dtype == "bf16":
            with O.bf16_operands():
                ref_out = O.nerf_forward(p, xin, cache=cache)
        else:
            ref_out = O.nerf_forward(p, xin, cache=cache)
        assert np.abs(outc - ref_out).max() <= (6e-3 if dtype == "bf16" else 2e-5) * np.abs(ref_out).max()
        acts = out.grad_fn.saved_tensors[0].float().cpu().numpy()[:, :n * S]
        for i in range(8):                                                    # masks from the stored activations
            cache[f"h{i+1}"] = acts[i]
        cache["final"], cache["d"] = acts[8], acts[9][:, :128]
        assert (cache["d"] >= 0).all() and (cache["d"] == 0).mean() > 0.05    # a ReLU output, not a softplus one
        # Sigmoid' from the kernel's own output, as the chain does: y3 = logit(rgb)
        rgb = np.clip(outc[:, :3].astype(np.float64), 1e-12, 1 - 1e-12)
        cache["y3"] = np.log(rgb / (1.0 - rgb))
        ref = O.nerf_backward(p, cache, g.reshape(-1, 4), operand_round=O.bf16_round if dtype == "bf16" else None)
    errs = {k: np.linalg.norm(got[k] - v) / max(np.linalg.norm(v), 1e-12) for k, v in ref.items()}
    bad = {k: e for k, e in errs.items() if e > (4e-3 if dtype == "bf16" else 5e-5)}
    assert not bad, bad
