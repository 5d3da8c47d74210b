"""GPU: the two generations of the fp32 MLP kernels hold the SAME BITS.  Round 6 replaced the LDS-ring kernels of rounds 1-5
(csrc/sn_mlp_fwd.hip: weights through an LDS ring filled by LDS-DMA, one barrier per slab, VALU epilogues) by csrc/sn_mlp_fwd_f32g.hip
(A fragments straight from L2 into a register ring, no VALU instruction in the trunk, ReLU by LDS integer max, no barrier) for the
inference entries AND the training forward; the old kernels stay reachable for A/B (sn_mlp_forward flag SN_FLAG_F32_LDS_RING,
SN_DTYPE_COMPILER_SCHEDULED on sn_mlp_forward_train).  Same MFMA order, same VALU heads -> torch.equal, on ragged shapes too.  Parity of
either against the oracle / the reference's goldens: tests/test_parity_gpu.py, tests/test_trained_weights_gpu.py."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import oracle_np as O                                          # noqa: E402
from tests.test_parity_gpu import dev, make_model                          # noqa: E402


def _mlp(model, rays_t, z_t, sigma_only, flags):
    from sinnerf_amd import rendering
    with torch.no_grad():
        return rendering._mlp(model, rays_t, z_t, sigma_only, flags)


@pytest.mark.parametrize("n,S", [(1, 1), (3, 37), (129, 64), (700, 192), (4096, 128)])
@pytest.mark.parametrize("new_act", [True, False])
def test_inference_kernel_generations_are_bit_identical(n, S, new_act):
    import sinnerf_amd
    from sinnerf_amd import _lib
    m = sinnerf_amd.NeRF(use_new_activation=new_act)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.trained_params("fine").items()})
    m = m.to(dev()).eval()
    rays = O.lego_rays(400, 400, seed=3)
    rays = np.ascontiguousarray(rays[np.random.RandomState(n).choice(rays.shape[0], n, replace=False)])
    z = np.sort(np.random.RandomState(S).uniform(2, 6, (n, S)).astype(np.float32), -1)
    rays_t, z_t = torch.from_numpy(rays).to(dev()), torch.from_numpy(z).to(dev())
    for sigma_only in (False, True):
        a = _mlp(m, rays_t, z_t, sigma_only, 0)
        b = _mlp(m, rays_t, z_t, sigma_only, _lib.SN_FLAG_F32_LDS_RING)
        assert torch.isfinite(a).all() and torch.equal(a, b), (n, S, sigma_only)
    # NeRF.forward's entry (pre-embedded rows, leading dimension 90 / 63)
    x = torch.randn(n * S, 90, device=dev())
    outs = []
    for flags in (0, _lib.SN_FLAG_F32_LDS_RING):
        o = torch.empty(n * S, 4, device=dev())
        _lib.check(_lib.lib.sn_mlp_forward_embedded(_lib.ptr(m.packed()), m.kernel_dtype(), _lib.ptr(x), n * S, 90, 0, flags, _lib.ptr(o),
                                                    _lib.stream_ptr()), "sn_mlp_forward_embedded")
        outs.append(o)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])


def _train_forward(m, rays_t, z_t, flag):
    from sinnerf_amd import _lib
    n, s = z_t.shape
    P = n * s
    rows = -(-P // 128) * 128
    d = rays_t.device
    out = torch.zeros((n, s, 4), dtype=torch.float32, device=d)
    acts = torch.full((10, rows, 256), float("nan"), dtype=torch.float32, device=d)
    emb = torch.zeros((rows, 128), dtype=torch.float32, device=d)
    _lib.check(_lib.lib.sn_mlp_forward_train(_lib.ptr(m.packed()), m.kernel_dtype(_lib.SN_DTYPE_F32) | flag, _lib.ptr(rays_t), _lib.ptr(z_t), n, s,
                                             _lib.ptr(out), _lib.ptr(acts), _lib.ptr(emb), rows, _lib.stream_ptr()), "sn_mlp_forward_train")
    torch.cuda.synchronize()
    return out, acts, emb


@pytest.mark.parametrize("n,S", [(60, 37), (4096, 128), (5, 3)])
@pytest.mark.parametrize("new_act", [True, False])
def test_training_forward_generations_write_the_same_state(n, S, new_act):
    """sn_mlp_forward_train(SN_DTYPE_F32): output, all ten activation slots (the rows of real points; slot 9 = 128 softplus / ReLU
    columns) and the embedded inputs, new kernel against the LDS-ring one, bit for bit -- and the output equals the inference kernel's."""
    import sinnerf_amd
    from sinnerf_amd import _lib
    m = sinnerf_amd.NeRF(use_new_activation=new_act)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.trained_params("coarse").items()})
    m = m.to(dev()).train()
    rays = O.lego_rays(400, 400, seed=0)
    rays = np.ascontiguousarray(rays[np.random.RandomState(7 + n).choice(rays.shape[0], n, replace=False)])
    z = np.sort(np.random.RandomState(S).uniform(2, 6, (n, S)).astype(np.float32), -1)
    rays_t, z_t = torch.from_numpy(rays).to(dev()), torch.from_numpy(z).to(dev())
    out_n, acts_n, emb_n = _train_forward(m, rays_t, z_t, 0)
    out_o, acts_o, emb_o = _train_forward(m, rays_t, z_t, _lib.SN_DTYPE_COMPILER_SCHEDULED)
    P = n * S
    assert torch.isfinite(out_n).all() and torch.equal(out_n, out_o)
    assert torch.equal(out_n, _mlp(m, rays_t, z_t, False, 0))
    for slot in range(10):
        w = 128 if slot == 9 else 256
        assert torch.isfinite(acts_n[slot, :P, :w]).all(), slot
        assert torch.equal(acts_n[slot, :P, :w], acts_o[slot, :P, :w]), slot
    assert torch.equal(emb_n[:P, :63], emb_o[:P, :63]) and torch.equal(emb_n[:P, 64:91], emb_o[:P, 64:91])


@pytest.mark.parametrize("n,S", [(70, 37), (4096, 128), (3, 5)])
@pytest.mark.parametrize("new_act", [True, False])
def test_backward_chain_generations_write_the_same_gradients(n, S, new_act):
    """sn_mlp_backward_chain(SN_DTYPE_F32): the round-6 chain (csrc/sn_mlp_bwd_f32g.hip: fragments from L2, the derivative mask as ONE VALU
    gap per slab, LDS round trip into the AGPRs, no barrier) against the round-2 LDS-ring chain (SN_DTYPE_COMPILER_SCHEDULED) on the same
    stored state: all ten slots of G (rows of real points; pad rows stay zero), the head block, g_out -- bit for bit."""
    import sinnerf_amd
    from sinnerf_amd import _lib
    d = dev()
    m = sinnerf_amd.NeRF(use_new_activation=new_act)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.trained_params("fine").items()})
    m = m.to(d).train()
    rays = O.lego_rays(400, 400, seed=0)
    rays = np.ascontiguousarray(rays[np.random.RandomState(11 + n).choice(rays.shape[0], n, replace=False)])
    z = np.sort(np.random.RandomState(S).uniform(2, 6, (n, S)).astype(np.float32), -1)
    rays_t, z_t = torch.from_numpy(rays).to(d), torch.from_numpy(z).to(d)
    out, acts, emb = _train_forward(m, rays_t, z_t, 0)
    P = n * S
    rows = acts.shape[1]
    acts[:, P:] = 0                                                          # pad rows: whatever the forward left there is not read back
    g = torch.from_numpy(np.random.RandomState(2).standard_normal((n, S, 4)).astype(np.float32)).to(d)
    res = []
    for flag in (0, _lib.SN_DTYPE_COMPILER_SCHEDULED):
        G = torch.full((10, rows, 256), float("nan"), device=d); G[:, P:] = 0
        g_o = torch.full((P, 4), float("nan"), device=d)
        _lib.check(_lib.lib.sn_mlp_backward_chain(_lib.ptr(m.packed_bwd("fp32")), m.kernel_dtype(_lib.SN_DTYPE_F32) | flag, _lib.ptr(acts),
                                                  _lib.ptr(out), _lib.ptr(g), P, rows, _lib.ptr(G), _lib.ptr(g_o), _lib.stream_ptr()), "chain")
        torch.cuda.synchronize()
        res.append((G, g_o))
    (Gn, gon), (Go, goo) = res
    assert torch.isfinite(gon).all() and torch.equal(gon, goo)
    for slot in range(10):
        w = 128 if slot == 9 else 256
        assert torch.isfinite(Gn[slot, :P, :w]).all(), slot
        assert torch.equal(Gn[slot, :P, :w], Go[slot, :P, :w]), slot
    assert torch.equal(Gn[9, :P, 128:160], Go[9, :P, 128:160])               # the rgb / sigma block of the weight-gradient kernels
