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


# ---- bf16x3 (3-term split) training kernels: tools/gen_x3_trunk.py / csrc/sn_mlp_fwd_bf16x3_t.hip -------------------------------
@pytest.fixture(scope="module")
def setup_x3():
    params = O.init_params(0, teacher=True)
    pts = np.random.RandomState(0).uniform(-1.5, 1.5, (128, 3)).astype(np.float32)
    return params, pts


def test_x3_training_trunk_stream_stores_the_split_state(setup_x3):
    """The generated bf16x3 training trunk EXECUTED (not model-checked: tests/test_x3_protocol_cpu.py walks the compiler-scheduled
    kernel's queue; this runs the real stream): every layer output reaches acts[] as (hi, lo) pairs in whole 128-byte point segments,
    decoded values within 1e-5 of each slot's range of the oracle under ``bf16x3_operands()``; the AGPR hand-over holds the same pairs;
    sign words per lane and tile pair; sigma head; the staging tile conflict-free; every counted wait / both barriers per slab needed.
    The ring rotation (76 slabs = 1 mod 3) is exercised with the tile starting in slot 1."""
    from tests.helpers import x3_state_decode
    params, pts = setup_x3
    gen = H.load_tool("gen_x3_trunk")
    g = gen.gen(dict(gen.KNOBS))
    run = H.X3TrunkRun(params, pts, rot=1).run(g.out)
    cache = {}
    x = np.concatenate([run.x_emb, np.zeros((128, 27), np.float32)], 1)
    with O.bf16x3_operands():
        O.nerf_forward(params, x, cache=cache)
    st = run.state()
    dec = x3_state_decode(st[:9])
    for l in range(9):
        ref = cache["h%d" % (l + 1)] if l < 8 else cache["final"]
        assert not (st[l].view(np.uint32) == 0xEEEEEEEE).any(), l          # every element of the slot written
        assert np.abs(dec[l] - ref).max() <= 1e-5 * np.abs(ref).max(), l
        if l < 8:
            assert (dec[l] >= 0).all() and ((dec[l] > 0) != (ref > 0)).mean() <= 1e-3
        # what is stored is a split: the lo part is a remainder, at most half an ulp of the hi part
        u = np.ascontiguousarray(st[l]).view(np.uint16).reshape(H.X3_ROWS, 32, 2, 8)
        hi, lo = H.G.bf16_to_f32(u[:, :, 0].astype(np.uint32)), H.G.bf16_to_f32(u[:, :, 1].astype(np.uint32))
        assert (np.abs(lo) <= 2.0 ** -8 * np.abs(hi) + 1e-38).all()
    assert np.array_equal(run.agpr_set(0), dec[8])                         # AGPR hand-over == stored state (what the backward relies on)
    sig = cache["h8"].astype(np.float64) @ params["sigma.weight"].astype(np.float64).T + params["sigma.bias"]
    assert np.abs(run.sigma() - sig[:, 0]).max() <= 2e-5
    assert run.wg.n_store_bytes == 4 * (9 * 8 * 4096 + 8 * 1024)           # per wave: 9 layers x 8 tiles x 4 KB + 8 sign rows of 1 KB
    for kind, per in (("ds_read_b128", 4), ("ds_write_b128", 8)):
        n, cyc, mn = run.wg.lds.stats[kind]
        assert cyc == per * n, (kind, cyc / n)                             # conflict-free in the guide's bank model
    for w_, v0 in zip(run.wg.waves, run.vo0):
        assert np.array_equal(w_.v[45], v0)                                # the running store offset is handed back unchanged
    # sign words (include/sinnerf_hip.h): for the 32 points of a wave, layer l in rows 4 l + (lane >> 4), bytes [512 + 16 (lane & 15), +16):
    # word t >> 1, bit d + 8 (t & 1) + 16 e  <->  [h_l[point j][32 t + feature(2 d + e, h)] > 0]
    words = st[9].view(np.uint32)[:, 128:192]
    lane = np.arange(64)
    j, h = lane & 31, lane >> 5
    for w in range(4):
        for l in range(8):
            wl = words[32 * w + 4 * l: 32 * w + 4 * l + 4].reshape(64, 4)          # [lane][tile pair]
            for t in range(8):
                for d in range(8):
                    for e in range(2):
                        r = 2 * d + e
                        feat = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h
                        bit = (wl[lane, t >> 1] >> (d + 8 * (t & 1) + 16 * e)) & 1
                        assert np.array_equal(bit, (dec[l][32 * w + j, feat] > 0).astype(np.uint32)), (w, l, t, d, e)
    # the checker is not vacuous: dropping a counted LDS wait / a vmcnt wait / a barrier is caught (a given wait can be redundant --
    # covered by the next one -- so each class is probed at three places and must be caught at least once; barriers at two parities:
    # B1 "slot free" and B2 "next slab visible")
    _dropping_is_caught(g.out, lambda lines: H.X3TrunkRun(params, pts, rot=2).run(lines))


def _dropping_is_caught(lines, run, probes=(40, 41, 58)):
    for what, pick in (("lgkmcnt", lambda l_: l_.startswith("s_waitcnt lgkmcnt")), ("vmcnt", lambda l_: l_.startswith("s_waitcnt vmcnt")),
                       ("barrier", lambda l_: l_.startswith("s_barrier"))):
        idxs = [i for i, l_ in enumerate(lines) if pick(l_)]
        caught = 0
        for which in probes:
            try:
                run(lines[:idxs[which]] + lines[idxs[which] + 1:])
            except H.G.SimError:
                caught += 1
        assert caught >= 1, what


def test_x3_backward_chain_stream_matches_the_split_emulated_oracle(setup_x3):
    """tools/gen_x3_chain.py (csrc/sn_mlp_bwd_bf16x3_t.hip): the 72-slab statement consumes what the simulated bf16x3 training forward
    stored (its sign words) and writes G[slot] as (hi, lo) pairs; decoded, within 2e-5 of each slot's range (two 16-bit roundings) of
    ``nerf_backward(operand_round="bf16x3")`` on the forward's stored state (the same masks); staging conflict-free; the weight stream
    wraps to the next point tile's slabs 0..2 (running offset back at its start value + 3 slabs)."""
    from tests.helpers import x3_state_decode
    params, pts = setup_x3
    gen, genc = H.load_tool("gen_x3_trunk"), H.load_tool("gen_x3_chain")
    fwd = H.X3TrunkRun(params, pts, rot=0).run(gen.gen(dict(gen.KNOBS)).out)
    st = fwd.state()
    dec = x3_state_decode(st[:9])
    x = np.concatenate([fwd.x_emb, np.random.RandomState(3).uniform(-1, 1, (128, 27)).astype(np.float32)], 1)
    cache = {}
    with O.bf16x3_operands():
        O.nerf_forward(params, x, cache=cache)
    for l in range(8):
        cache["h%d" % (l + 1)] = dec[l]
    cache["final"] = dec[8]
    g_out = np.random.RandomState(1).standard_normal((128, 4)).astype(np.float32)
    gy = {}
    O.nerf_backward(params, cache, g_out, gy_out=gy, operand_round="bf16x3")
    g = genc.gen(dict(genc.KNOBS))
    run = H.X3ChainRun(params, st, gy["dir"].astype(np.float32), g_out[:, 3]).run(g.out)
    Gs = run.state()
    Gd = x3_state_decode(Gs[:9])
    names = {8: "final", **{l: "l%d" % (l + 1) for l in range(8)}}
    for slot in range(9):
        assert not (Gs[slot].view(np.uint32) == 0xEEEEEEEE).any(), slot
        want = gy[names[slot]]
        assert np.abs(Gd[slot] - want).max() <= 2e-5 * np.abs(want).max(), (slot, np.abs(Gd[slot] - want).max() / np.abs(want).max())
        assert np.array_equal(Gd[slot] == 0, want == 0) or ((Gd[slot] == 0) != (want == 0)).mean() < 1e-4, slot     # the same masks
    for kind, per in (("ds_read_b128", 4), ("ds_write_b128", 8)):
        n, cyc, mn = run.wg.lds.stats[kind]
        assert cyc == per * n, kind
    assert run.wg.n_store_bytes == 4 * 9 * 8 * 4096
    for w_, g0 in zip(run.wg.waves, run.goff0):
        assert np.array_equal(w_.v[35], g0)                                   # requested: slabs 3..71, then the wrap and slabs 0..2 again
    _dropping_is_caught(g.out, lambda lines: H.X3ChainRun(params, st, gy["dir"].astype(np.float32), g_out[:, 3]).run(lines))
