"""CPU: the packed-weight layout + kernel schedule, checked through a register-level emulation of the MFMA
dataflow against the oracle; and the C ABI surface (library loads, every declared symbol resolves)."""
import ctypes
import os
import re

import numpy as np

from oracle import oracle_np as O
from tests.mfma_emulator import emulate_bwd_tile, emulate_tile, pack_blob, pack_blob_bwd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from sinnerf_amd import _lib
    return _lib


def test_abi_exports_every_declared_symbol():
    L = _lib()
    hdr = open(os.path.join(REPO, "include", "sinnerf_hip.h")).read()
    declared = set(re.findall(r"\b(sn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 14
    raw = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/sinnerf_hip.h but not exported"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    assert L.lib.sn_abi_version() == L.ABI_VERSION == 5
    assert L.lib.sn_packed_weights_bytes(0) == 4 * (32 * (8 * 64 + 24 * 256 + 8 * 320 + 32 * 256 + 4 * 288) + 76 * 32 + 648)
    assert L.lib.sn_error_string(-3).decode().startswith("perturb")


def test_pack_table_is_a_bijection_onto_the_raw_weights():
    L = _lib()
    p = O.init_params(0, True)
    blob, table = pack_blob(L.lib, p)
    src = table[:, 1]
    used = src[src >= 0]
    assert len(np.unique(used)) == len(used) == 595844            # every parameter exactly once
    assert np.isclose(float(np.sum(blob, dtype=np.float64)),
                      sum(float(np.sum(v, dtype=np.float64)) for v in p.values()), rtol=0, atol=1e-3)


def test_emulated_kernel_matches_oracle():
    L = _lib()
    p = O.init_params(3, True)
    blob, _ = pack_blob(L.lib, p)
    r = np.random.RandomState(0)
    x = O.embedding(r.uniform(-3, 3, (32, 3)).astype(np.float32), 10)
    d = O.embedding(r.uniform(-1, 1, (32, 3)).astype(np.float32), 4)
    xin = np.concatenate([x, d], 1)
    got = emulate_tile(L.lib, blob, xin)
    ref = O.nerf_forward(p, xin)
    assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    got_s = emulate_tile(L.lib, blob, xin, sigma_only=True)
    assert np.abs(got_s - ref[:, 3]).max() <= 5e-5 * max(1.0, np.abs(ref[:, 3]).max())


def test_emulated_backward_chain_matches_oracle():
    L = _lib()
    p = O.init_params(4, True)
    bblob = pack_blob_bwd(L.lib, p)
    r = np.random.RandomState(1)
    x = O.embedding(r.uniform(-3, 3, (32, 3)).astype(np.float32), 10)
    d = O.embedding(r.uniform(-1, 1, (32, 3)).astype(np.float32), 4)
    xin = np.concatenate([x, d], 1)
    cache = {}
    out = O.nerf_forward(p, xin, cache=cache)
    g_raw = r.standard_normal((32, 4)).astype(np.float32)
    gy = {}
    O.nerf_backward(p, cache, g_raw, gy_out=gy)
    acts = {i: cache[f"h{i+1}"] for i in range(8)}
    acts[8] = cache["final"]
    acts[9] = np.concatenate([cache["d"], np.zeros((32, 128), np.float32)], 1)
    G, g_out = emulate_bwd_tile(bblob, acts, out, g_raw)
    ref = {i: gy[f"l{i+1}"] for i in range(8)}
    ref[8] = gy["final"]
    ref[9] = np.concatenate([gy["dir"], np.zeros((32, 128))], 1)
    assert np.abs(g_out - np.concatenate([gy["rgb"], gy["sigma"]], 1)).max() <= 1e-6
    for slot in range(10):
        scale = max(np.abs(ref[slot]).max(), 1e-9)
        assert np.abs(G[slot] - ref[slot]).max() <= 2e-5 * scale, (slot, np.abs(G[slot] - ref[slot]).max(), scale)


def test_bf16_backward_blob_table_covers_the_transposed_weights_once():
    """sn_build_pack_table_bwd_bf16: every element of dir_encoding[:, :256], xyz_encoding_final and the hidden blocks of
    xyz_encoding_2..8 appears exactly once among the bf16 slabs; the fp32 tail carries 72 x 32 zeros, rgb.0.weight and
    sigma.weight once each."""
    import ctypes
    from sinnerf_amd import _lib
    n = _lib.lib.sn_pack_table_entries_bwd_bf16()
    tab = np.zeros((n, 2), np.int32)
    assert _lib.lib.sn_build_pack_table_bwd_bf16(ctypes.c_void_p(tab.ctypes.data)) == 0
    dst, src = tab[:, 0].astype(np.int64), tab[:, 1].astype(np.int64)
    n_w = 8 * 32 * 128 + 64 * 32 * 256
    assert n == n_w + 72 * 32 + 640 and _lib.lib.sn_packed_weights_bytes_bwd_bf16() == 2 * n_w + 4 * (72 * 32 + 640)
    assert np.array_equal(np.sort(dst[:n_w]), 2 * np.arange(n_w))                    # bf16 slabs: dense, no overlap
    assert np.array_equal(dst[n_w:], 2 * n_w + 4 * np.arange(72 * 32 + 640))         # fp32 tail
    w = src[:n_w]
    assert (w >= 0).all() and ((w >> 30) & 1 == 0).all()
    tid, off = (w >> 20) & 0x3ff, w & 0xfffff
    def cols(t, ncol, c0, c1, rows):
        m = tid == t
        r, c = off[m] // ncol, off[m] % ncol
        assert m.sum() == rows * (c1 - c0) and len(set(zip(r.tolist(), c.tolist()))) == m.sum(), t
        assert r.min() == 0 and r.max() == rows - 1 and c.min() == c0 and c.max() == c1 - 1, t
    cols(18, 283, 0, 256, 128)                                                        # dir_encoding.0.weight[:, :256]
    cols(16, 256, 0, 256, 256)                                                        # xyz_encoding_final.weight
    for li in range(1, 8):
        cols(2 * li, 319 if li == 4 else 256, 63 if li == 4 else 0, 319 if li == 4 else 256, 256)
    tail = src[n_w:]
    assert (tail[:72 * 32] == -2).all()
    aux = tail[72 * 32:]
    assert ((aux >> 30) & 1 == 1).all()
    at, ao = (aux >> 20) & 0x3ff, aux & 0xfffff
    assert sorted(ao[at == 22].tolist()) == list(range(384)) and sorted(ao[at == 20].tolist()) == list(range(256))


def test_bf16x3_pack_table_pairs_every_weight_with_its_remainder():
    """SN_DTYPE_BF16X3 blob (csrc/sn_layout.h DT_BF16X3): per slab and k-step the hi fragment (64 lanes x 8 bf16) then the lo
    fragment of the SAME weights (SRC_LO_FLAG), dense over [0, 4 x weights); the fp32 tail (biases, head table) is the fp32 blob's"""
    from sinnerf_amd import _lib
    n3, n32 = _lib.lib.sn_pack_table_entries_dtype(3), _lib.lib.sn_pack_table_entries_dtype(0)
    t3, t32 = np.zeros((n3, 2), np.int32), np.zeros((n32, 2), np.int32)
    assert _lib.lib.sn_build_pack_table(3, ctypes.c_void_p(t3.ctypes.data)) == 0
    assert _lib.lib.sn_build_pack_table(0, ctypes.c_void_p(t32.ctypes.data)) == 0
    n_w = 593920
    assert n32 == n_w + 3080 == _lib.lib.sn_pack_table_entries() and n3 == 2 * n_w + 3080
    assert _lib.lib.sn_packed_weights_bytes(3) == _lib.lib.sn_packed_weights_bytes(0) == 4 * n_w + 4 * 3080
    assert np.array_equal(t3[2 * n_w:], t32[n_w:])                                     # same tail, same place
    dst, src = t3[:2 * n_w, 0].astype(np.int64), t3[:2 * n_w, 1].astype(np.int64)
    assert np.array_equal(np.sort(dst), 2 * np.arange(2 * n_w))                        # dense, no overlap
    LO = 1 << 29
    frag = (dst // 1024) % 2                                                            # 1 KB fragments alternate hi, lo
    live = src >= 0
    assert ((src[live] & LO != 0) == (frag[live] == 1)).all()
    hi, lo = (frag == 0), (frag == 1)
    order_hi, order_lo = np.argsort(dst[hi]), np.argsort(dst[lo])
    assert np.array_equal(dst[hi][order_hi] + 1024, dst[lo][order_lo])                # lo fragment right behind its hi fragment
    sh, sl = src[hi][order_hi], src[lo][order_lo]
    assert np.array_equal(np.where(sh >= 0, sh | LO, sh), sl)                          # ... of the same raw element
    used = sh[sh >= 0]
    # the hi halves alone reproduce the bf16 blob's table (same K-slot order, element size 2 -> 4)
    tb = np.zeros((n32, 2), np.int32)
    assert _lib.lib.sn_build_pack_table(1, ctypes.c_void_p(tb.ctypes.data)) == 0
    bdst, bsrc = tb[:n_w, 0].astype(np.int64), tb[:n_w, 1].astype(np.int64)
    ob = np.argsort(bdst)
    assert np.array_equal(bsrc[ob], sh)
    assert np.array_equal((bdst[ob] // 1024) * 2048 + bdst[ob] % 1024, dst[hi][order_hi])


def test_generated_instruction_streams_are_well_formed(tmp_path):
    """tools/gen_dw_f32.py / gen_dw_bf16.py / gen_bf16_trunk.py (run by csrc/Makefile): deterministic output, the MFMA count
    the kernels' tilings imply, every accumulator block touched once per k-step, and the DMA destination set one
    instruction ahead of its use (m0 -> LDS-DMA needs a wait state hipcc cannot insert into inline asm)."""
    import importlib.util
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    def check_m0(body):
        for i, l in enumerate(body):
            if l.startswith("global_load_lds"):
                prev = [k for k in range(i) if body[k].startswith("s_add_u32 m0")]
                assert prev and i - prev[-1] >= 2, (i, l)                       # at least one instruction in between

    f32 = load("gen_dw_f32").gen()
    assert f32 == load("gen_dw_f32").gen()
    mf = [l for l in f32 if l.startswith("v_mfma_f32_32x32x2_f32")]
    assert len(mf) == 128 and len({re.match(r"v_mfma\S+ a\[(\d+):", l).group(1) for l in mf}) == 16   # 8 k-step pairs x 16 blocks
    assert sum(l.startswith("global_load_lds") for l in f32) == 8               # A + B tile of a 16-point chunk: 8 x 4 KB
    check_m0(f32)

    g16 = load("gen_dw_bf16")
    pair, tail = g16.gen(), g16.gen_tail()
    assert pair == g16.gen()
    for body, n_mfma, n_reads, n_dma in ((pair, 32, 32, 8), (tail, 16, 16, 0)):
        mf = [l for l in body if l.startswith("v_mfma_f32_32x32x16_bf16")]
        assert len(mf) == n_mfma and len({re.match(r"v_mfma\S+ a\[(\d+):", l).group(1) for l in mf}) == 16
        assert sum(l.startswith("ds_read_b64_tr_b16") for l in body) == n_reads
        assert sum(l.startswith("global_load_lds") for l in body) == n_dma
        check_m0(body)
        # a fragment set is complete (lgkmcnt(0)) before the first MFMA that reads it
        first_mfma = next(i for i, l in enumerate(body) if l.startswith("v_mfma"))
        assert any(l == "s_waitcnt lgkmcnt(0)" for l in body[:first_mfma])
    assert pair[0] == "s_waitcnt vmcnt(16)" and pair[1] == "s_barrier"          # two chunks landed, four (x 4 pieces) in flight

    out = tmp_path / "trunk.inc"
    trunk = load("gen_bf16_trunk")
    import sys
    argv = sys.argv
    try:
        sys.argv = ["gen_bf16_trunk.py", str(out)]
        trunk.main()
    finally:
        sys.argv = argv
    text = out.read_text()
    assert text.count("v_mfma_f32_32x32x16_bf16") == 2176                       # 72 slabs of a wave's two point tiles


def test_training_streams_are_deterministic_and_complete():
    """tools/gen_bf16_trunk.py store=1 and tools/gen_bf16_chain.py (the training forward / backward chain statements): deterministic
    output, all 2176 MFMAs of a wave's two point tiles, every row of state leaves through exactly one global_store_dwordx4, every
    LDS-DMA destination is set one instruction ahead of its use, counted waits stay inside their fields, and the inference trunk's
    text did not change under the store-mode additions."""
    import importlib.util
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        import sys
        sys.path.insert(0, os.path.join(root, "tools"))
        spec.loader.exec_module(mod)
        return mod

    T, C = load("gen_bf16_trunk"), load("gen_bf16_chain")
    fwd = T.gen(dict(T.KNOBS, **T.STORE_KNOBS)).out
    assert fwd == T.gen(dict(T.KNOBS, **T.STORE_KNOBS)).out
    chain = C.gen(dict(C.KNOBS)).out
    assert chain == C.gen(dict(C.KNOBS)).out
    for body, n_sign_st, n_sign_ld in ((fwd, 64, 0), (chain, 0, 64)):
        assert sum(l.startswith("v_mfma_f32_32x32x16_bf16") for l in body) == 2176
        assert sum(l.startswith("global_store_dwordx4") for l in body) == 9 * 4 * 8          # 9 layers x 4 tile pairs x 8 row groups
        assert sum(l.startswith("global_store_dword ") for l in body) == n_sign_st
        assert sum(l.startswith("global_load_dword ") for l in body) == n_sign_ld
        assert sum(l.startswith("v_permlane32_swap_b32") for l in body) == 72 * 8
        assert sum(l.startswith("ds_write_b128") for l in body) == 72 * 4
        for i, l in enumerate(body):
            if l.startswith("global_load_lds"):
                prev = [k for k in range(i) if body[k].startswith("s_add_u32 m0")]
                assert prev and i - prev[-1] >= 2, (i, l)
            m = re.match(r"s_waitcnt (lgkmcnt|vmcnt)\((\d+)\)", l)
            if m:
                assert int(m.group(2)) <= (15 if m.group(1) == "lgkmcnt" else 63), l
    # the statements stay inside the registers they declare: v128.. (forward), v64.. (chain: the sign-word file)
    for body, lo in ((fwd, 128), (chain, 64)):
        regs = set()
        for l in body:
            regs |= {int(x) for x in re.findall(r"\bv(\d+)\b", l)}
            for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", l):
                regs |= set(range(int(a), int(b) + 1))
        assert min(regs) >= lo, min(regs)


def test_bf16_emb_positions_invert_the_layout_slot_maps():
    """SN_DTYPE_EMB_BF16 stores the embedded inputs at their K-slot positions (32 h + e / 16 h + e); sn_dw.hip's emb_xyz_pos / emb_dir_pos
    (restated here, and in tests/test_training_kernels_system_gpu.py where the device code is checked) must be the inverse of csrc/sn_layout.h's
    slot -> column maps the packed weights are built from."""
    from sinnerf_amd import _lib as L

    def xyz_pos(c):
        if c < 3:
            return (30, 31, 62)[c]
        k, h = (c - 3) % 30, (c - 3) // 30
        return 32 * h + 2 * (3 * (k // 6) + k % 3) + (k % 6) // 3

    def dir_pos(c):
        if c < 3:
            return (12, 13, 28)[c]
        k, h = (c - 3) % 12, (c - 3) // 12
        return 16 * h + 2 * (3 * (k // 6) + k % 3) + (k % 6) // 3

    for c in range(63):
        p = xyz_pos(c)
        assert L.lib.sn_layout_xyz_slot_col(p // 32, p % 32) == c
    for c in range(27):
        p = dir_pos(c)
        assert L.lib.sn_layout_dir_slot_col(p // 16, p % 16) == c
    # the positions no column maps to are exactly the pad slots of the layout
    assert {32 * h + e for h in (0, 1) for e in range(32) if L.lib.sn_layout_xyz_slot_col(h, e) < 0} == set(range(64)) - {xyz_pos(c) for c in range(63)}
    assert {16 * h + e for h in (0, 1) for e in range(16) if L.lib.sn_layout_dir_slot_col(h, e) < 0} == set(range(32)) - {dir_pos(c) for c in range(27)}


def test_x3_state_layout_helpers_round_trip():
    """tests/helpers.py x3_state_encode / _decode restate the bf16x3 training-state layout (csrc/sn_layout.h "x3 state",
    include/sinnerf_hip.h sn_mlp_forward_train): per 8 consecutive features 16 B of hi parts (RNE bf16 of x), then 16 B of lo parts
    (RNE bf16 of x - hi); hi + lo = x to 2^-16 relative; a row keeps its 1 KB."""
    from tests.helpers import x3_state_decode, x3_state_encode
    x = np.random.RandomState(0).standard_normal((3, 7, 256)).astype(np.float32)
    x[0, 0, :8] = [0.0, -0.0, 1.0, -1.0, 3.0e-39, 65504.0, 1.0 + 2.0 ** -9, 1.0 - 2.0 ** -10]      # zeros, a subnormal, ties of the hi rounding
    e = x3_state_encode(x)
    assert e.shape == x.shape and e.dtype == np.float32
    u = e.view(np.uint16).reshape(3, 7, 32, 2, 8)
    hi = (u[..., 0, :].astype(np.uint32) << 16).view(np.float32).reshape(3, 7, 256)
    lo = (u[..., 1, :].astype(np.uint32) << 16).view(np.float32).reshape(3, 7, 256)
    assert np.array_equal(hi, O.bf16_round(x))                                      # the oracle's RNE bf16
    assert np.array_equal(lo, O.bf16_round(x - hi))
    d = x3_state_decode(e)
    assert np.array_equal(d, hi + lo)
    nz = np.abs(x) > 1e-30
    assert (np.abs(d - x)[nz] / np.abs(x)[nz]).max() <= 2.0 ** -16


def test_x3_split_tile_swizzle_is_consistent_and_bank_conflict_free():
    """The staged image of a SPLIT tile in the bf16x3 weight-gradient kernel (csrc/sn_dw.hip RowStager<.., SPLIT>), restated: LDS piece
    (row, lp) receives the global 16-byte piece lp ^ swz(row), swz(row) = (row & 1) | ((row & 2) << 2); lane (q, G) of a transpose read
    addresses the 8-byte group of features f .. f+3 of row 8 (G >> 1) + (q >> 2) (+ 4 for the second read), hi part, lo part at ^ 16.
    (a) every lane's address holds exactly the global bytes it wants; (b) the 32 lanes a ds_read_b64_tr_b16 serves together
    (MI355X_MICROARCH.md LDS table: 2 x 32 lanes, bank = (a / 4) mod 64) cover 64 distinct banks -- for hi and lo, every tile column
    block, both reads, every tile width."""
    swz = lambda row: (row & 1) | ((row & 2) << 2)
    for W in (64, 128, 256):                                               # features per row of the tile (row = W * 4 bytes)
        per_row = W // 4                                                   # 16-byte pieces per row
        lds = {}                                                           # LDS byte address of a piece -> (row, global piece)
        for row in range(16):
            for lp in range(per_row):
                lds[row * W * 4 + lp * 16] = (row, lp ^ swz(row))
        assert len(lds) == 16 * per_row
        for f0 in range(0, W, 32):                                         # accumulator tile column block
            for second in (0, 1):
                for part in (0, 1):
                    addrs = []
                    for lane in range(64):
                        q, G = lane & 15, lane >> 4
                        row = 8 * (G >> 1) + (q >> 2) + 4 * second
                        f = f0 + 16 * (G & 1) + 4 * (q & 3)
                        off = row * W * 4 + ((2 * (f >> 3)) ^ swz(row)) * 16 + 8 * ((f >> 2) & 1)
                        off ^= 16 * part
                        r, gp = lds[off & ~15]
                        assert r == row and gp == 2 * (f >> 3) + part, (W, lane, part)      # (a) hi piece 2 (f / 8), lo piece + 1
                        assert (off & 15) == 8 * ((f >> 2) & 1)
                        addrs.append(off)
                    for half in (addrs[:32], addrs[32:]):                  # (b)
                        banks = [(a // 4 + d) % 64 for a in half for d in (0, 1)]
                        assert len(set(banks)) == 64, (W, f0, second, part)


def test_relu_as_signed_integer_max_is_exact():
    """csrc/sn_mlp_pipe.h epi32_relu_lds / lds_relu_word (round 6): the fp32 kernels apply ReLU in LDS as `ds_max_i32(word, 0)` -- no VALU
    instruction next to the f32-input MFMA.  As signed integers every negative float (and -0.0) is < 0 and every positive float is its own
    bit pattern, monotone in value: max_i32(bits(x), 0) must be bits(max(x, +0.0)) for every non-NaN x, incl. denormals and infinities."""
    r = np.random.RandomState(0)
    x = np.concatenate([r.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** r.randint(-38, 38, 200000).astype(np.float32),
                        np.array([0.0, -0.0, 1e-45, -1e-45, 1.1754944e-38, -1.1754944e-38, np.inf, -np.inf, 3.4028235e38, -3.4028235e38,
                                  1.0, -1.0], np.float32)])
    x = x[np.isfinite(x) | np.isinf(x)]
    got = np.maximum(x.view(np.int32), np.int32(0)).view(np.float32)
    want = np.where(x > 0, x, np.float32(0.0)).astype(np.float32)            # max(x, +0.0) with -0.0 -> +0.0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
