"""The GENERATED instruction streams of the bf16 MLP kernels, executed on the CPU by tools/gcn_sim.py (a functional simulator of
the instruction subset they use that also checks every asynchronous hand-off: counted s_waitcnt coverage of LDS reads and LDS-DMA
pieces, barrier visibility across the workgroup's waves, LDS overwrite races) and compared with the bf16-emulated numpy oracle.
No GPU: the blob comes from the library's host-side pack table, the LDS / register inputs are what the kernel sources set up
(tests/stream_harness.py)."""
import numpy as np
import pytest

from tests import stream_harness as H

O = H.O


@pytest.fixture(scope="module")
def setup():
    params = O.init_params(0, teacher=True)
    pts = np.random.RandomState(0).uniform(-1.5, 1.5, (256, 3)).astype(np.float32)
    return params, pts


def _close_bf16(got, want, frac=5e-3):
    """equal up to single bf16-ulp flips of values on a rounding boundary (fp32 accumulation order differs)"""
    bad = got != want
    assert bad.mean() <= frac, bad.mean()
    assert np.abs(got - want).max() <= 2.0 ** -7 * max(1.0, np.abs(want).max())


def test_inference_trunk_stream_matches_oracle_and_every_wait_covers_its_data(setup):
    params, pts = setup
    gen = H.load_tool("gen_bf16_trunk")
    g = gen.gen(dict(gen.KNOBS))
    run = H.TrunkRun(params, pts).run(g.out)
    cache, sig = H.oracle_trunk(params, run.x_emb)
    _close_bf16(run.activation_set(0), O.bf16_round(cache["final"]))
    assert np.abs(run.sigma() - sig).max() <= 2e-4
    n, cyc, mn = run.wg.lds.stats["ds_read_b128"]
    assert cyc == 4 * n                                       # A-fragment / bias reads: conflict-free (4 LDS cycles each)
    # the checker is not vacuous: dropping one counted wait / one vmcnt wait / one barrier is caught
    lines = g.out
    for what, pick in (("lgkmcnt", lambda l: l.startswith("s_waitcnt lgkmcnt")), ("vmcnt", lambda l: l.startswith("s_waitcnt vmcnt")),
                       ("barrier", lambda l: l.startswith("s_barrier"))):
        idx = [i for i, l in enumerate(lines) if pick(l)][30]
        with pytest.raises(H.G.SimError):
            H.TrunkRun(params, pts).run(lines[:idx] + lines[idx + 1:])


def test_training_trunk_stream_stores_state_and_sign_words(setup):
    """store mode (csrc/sn_mlp_fwd_bf16_t.hip): besides the activation hand-over in AGPRs every layer output reaches acts[] as
    bf16 in whole 128-byte rows, the ReLU sign words reach the unused half of slot 9, the staging tile is conflict-free."""
    params, pts = setup
    gen = H.load_tool("gen_bf16_trunk")
    g = gen.gen(dict(gen.KNOBS, **gen.STORE_KNOBS))
    run = H.TrainTrunkRun(params, pts).run(g.out)
    cache, sig = H.oracle_trunk(params, run.x_emb)
    _close_bf16(run.activation_set(0), O.bf16_round(cache["final"]))
    assert np.abs(run.sigma() - sig).max() <= 2e-4
    raw = run.acts.view(np.uint16).reshape(10, H.SLOT_ROWS, 256)
    for l in range(9):
        assert not (raw[l] == 0xEEEE).any(), l                # every element of the slot written
        _close_bf16(run.stored(l), O.bf16_round(cache["h%d" % (l + 1)] if l < 8 else cache["final"]))
    # the AGPR hand-over and the stored state are the SAME values (what the backward relies on)
    assert np.array_equal(run.stored(8), run.activation_set(0))
    assert run.wg.n_store_bytes == 4 * (9 * 64 * 512 + 64 * 256)          # per wave: 9 layers x 64 rows x 512 B + 64 sign rows
    for kind, per in (("ds_read_b128", 4), ("ds_write_b128", 8)):
        n, cyc, mn = run.wg.lds.stats[kind]
        assert cyc == per * n, (kind, cyc / n)                # conflict-free: 4 LDS cycles per b128 read (16 lanes over 64 banks), 8 per write (8 lanes over 32)
    # sign words: bit `step` = sign of the LOW half of packed word `step`, bit 16 + step = its HIGH half; word order of a tile:
    # point tile outermost in layers 1..7, quad outermost in layer 8 (the chain's convention, csrc/sn_mlp_bf16.h)
    sw = run.sign_words()
    lane = np.arange(64)
    j, h = lane & 31, lane >> 5
    for l in range(8):
        act = run.stored(l)
        for t in range(8):
            for w in range(4):
                word = sw[w, 8 * l + t]
                for pt in range(2):
                    for i in range(4):
                        for e in range(2):
                            step = (2 * (2 * i + pt) + e) if l == 7 else (8 * pt + 2 * i + e)
                            for half in range(2):
                                feat = 32 * t + 8 * i + 4 * h + 2 * e + half
                                bit = (word >> (step + 16 * half)) & 1
                                a = act[64 * w + 32 * pt + j, feat]
                                assert not (bit & (a > 0)).any()        # a stored positive activation never carries a sign bit
                                assert ((bit == 0) & (a == 0)).mean() <= 0.02   # zero without sign bit: only an exact +0 pre-activation


def test_backward_chain_stream_matches_oracle(setup):
    """tools/gen_bf16_chain.py (csrc/sn_mlp_bwd_bf16_t.hip): the 72-slab statement consumes what the simulated training forward
    stored (sign words, activations) and writes G[slot] = bf16 pre-activation gradients; against ``nerf_backward`` with both
    operands of every contraction rounded to bf16 (masks from the forward's stored values, as on the GPU)."""
    params, pts = setup
    gen = H.load_tool("gen_bf16_trunk"); genc = H.load_tool("gen_bf16_chain")
    fwd = H.TrainTrunkRun(params, pts).run(gen.gen(dict(gen.KNOBS, **gen.STORE_KNOBS)).out)
    cache, _ = H.oracle_trunk(params, fwd.x_emb)
    for l in range(8):
        cache["h%d" % (l + 1)] = fwd.stored(l)
    cache["final"] = fwd.stored(8)
    g_out = np.random.RandomState(1).standard_normal((256, 4)).astype(np.float32)
    gy = {}
    O.nerf_backward(params, cache, g_out, gy_out=gy, operand_round=O.bf16_round)
    run = H.ChainRun(params, fwd.acts, gy["dir"].astype(np.float32), g_out[:, 3]).run(genc.gen(dict(genc.KNOBS)).out)
    raw = run.G.view(np.uint16).reshape(10, H.SLOT_ROWS, 256)
    names = {8: "final", **{l: "l%d" % (l + 1) for l in range(8)}}
    for slot in range(9):
        assert not (raw[slot] == 0xEEEE).any(), slot
        got, want = run.stored(slot), O.bf16_round(gy[names[slot]].astype(np.float32))
        bad = got != want
        assert bad.mean() <= 1e-2, (slot, bad.mean())                         # bf16-ulp flips + a few exact-zero mask cases
        assert np.abs(got - want).max() <= 2.0 ** -6 * np.abs(want).max(), slot
    for kind, per in (("ds_read_b128", 4), ("ds_write_b128", 8)):
        n, cyc, mn = run.wg.lds.stats[kind]
        assert cyc == per * n, kind
    assert run.wg.n_store_bytes == 4 * 9 * 64 * 512
