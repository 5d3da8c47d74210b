"""The GENERATED chunk-pair statements of the bf16-state weight-gradient kernels (tools/gen_dw_narrow.py -> csrc/sn_dw_narrow_bf16.hip;
tools/gen_dw_bf16.py -> csrc/sn_dw_bf16.hip, shape 0 below), executed on the CPU by tools/gcn_sim.py: one workgroup = one K-range of
one problem, the 256 x 256 shape and every one of the seven narrow ones, even and odd chunk counts, against the fp64 contraction of
the same bf16 operands.  The simulator also checks what
the statements promise the hardware: every counted s_waitcnt covers the LDS reads / LDS-DMA pieces it is meant to cover, a chunk is
only read behind the barrier that follows its landing, a ring slot is only restaged after every wave has left it -- for BOTH forms of
the statement (waves that stage both tiles / the others).  The C++ glue of the kernel (offsets, ring bookkeeping, prologue) is
restated here from csrc/sn_dw_narrow_bf16.hip."""
import numpy as np
import pytest

from tests import stream_harness as H

G = H.G
LANE = np.arange(64)
KB = 16
GA_BASE, GB_BASE = 0x10000000, 0x30000000
# operand registers of the harness (everything below the statements' own v48..v127)
BIND = dict(oa0="v0", oa1="v1", ob0="v2", ob1="v3", ta0="v4", ta1="v5", ta2="v6", ta3="v7", tb0="v8", tb1="v9", tb2="v10", tb3="v11",
            tbf="v12", one="v13", bs0="v14", bs1="v15", bs2="v16", bs3="v17",
            sl0="s20", sl1="s21", ga0="s[22:23]", gb0="s[24:25]", ga1="s[26:27]", gb1="s[28:29]", md0="s30", md1="s31")


def tr_offset(W, row, col):
    """byte offset of the 8-byte group (row, columns col..col+3) in a staged W-wide bf16 tile (dwn_tr_offset)"""
    return row * W * 2 + swz(W * 2 // 16, row, col // 8) * 16 + (col * 2) % 16


def swz(per_row, row, p):
    """the swizzled DMA image (dwn_swz): LDS piece (row, p) holds global piece (row, swz)"""
    return p ^ (4 * (row & 3)) if per_row >= 16 else p ^ (4 * ((row >> 1) & 1)) if per_row == 8 else p


class BigShape:
    """the 256 x 256 kernel of tools/gen_dw_bf16.py (csrc/sn_dw_bf16.hip) in the terms of gen_dw_narrow.Shape"""
    WA = WB = 256
    R, BUF, A_BYTES = 8, 16384, 8192
    nA = nB = 2
    a_waves = b_waves = 4


def run_task(v, n_chunks, seed=0, lda=256, a_col0=0, b_col0=0, pairs_only=False):
    if v == 0:
        gen = H.load_tool("gen_dw_bf16")
        sh = BigShape
        MT, NT, WM, WN, EB = 4, 4, 2, 2, 2
        gen_pair, gen_tail = (lambda sh_, full: gen.gen()), (lambda sh_: gen.gen_tail())
    else:
        gen = H.load_tool("gen_dw_narrow")
        sh = gen.Shape(v)
        MT, NT, WM, WN, EB = gen.VARIANTS[v]
        gen_pair, gen_tail = gen.gen_pair, gen.gen_tail
    WA, WB, R, BUF, A_BYTES = sh.WA, sh.WB, sh.R, sh.BUF, sh.A_BYTES
    ldb = 256 if WB > 128 else 128
    rs = np.random.RandomState(seed)
    K = n_chunks * KB
    ga = G.bf16_rne(rs.standard_normal((K, lda)).astype(np.float32)).astype(np.uint16)
    if EB == 2:
        gb = G.bf16_rne(rs.standard_normal((K, ldb)).astype(np.float32)).astype(np.uint16)
        b_val = G.bf16_to_f32(gb.astype(np.uint32))
    else:
        gb = rs.standard_normal((K, ldb)).astype(np.float32)
        b_val = G.bf16_to_f32(G.bf16_rne(gb))
    a_val = G.bf16_to_f32(ga.astype(np.uint32))
    wg = G.Workgroup(4, lds_bytes=81920 if v else 131072)
    wg.mem.add("a", GA_BASE, data=ga.tobytes(), writable=False)
    wg.mem.add("b", GB_BASE, data=gb.tobytes(), writable=False)
    a_base, b_base = GA_BASE + a_col0 * 2, GB_BASE + b_col0 * EB
    chunk_a = lambda c: a_base + min(c, n_chunks - 1) * KB * lda * 2
    chunk_b = lambda c: b_base + min(c, n_chunks - 1) * KB * ldb * EB
    pair_full, pair_other, tail = gen_pair(sh, True), gen_pair(sh, False), gen_tail(sh)
    zero = ["v_accvgpr_write_b32 a%d, 0" % i for i in range(16 * MT * NT)]
    n_full = min(sh.a_waves, sh.b_waves)
    nch = getattr(gen, "QUAD", {}).get(v, 2) if (v and not pairs_only) else 2      # chunks per sync point (csrc: SN_DWN_TASK_Q)
    if nch != 2:
        pair_full, pair_other = gen.gen_group(sh, True, nch), gen.gen_group(sh, False, nch)
    programs = []
    for w, wave in enumerate(wg.waves):
        tid = 64 * w + LANE
        wr, wc = w // WN, w % WN
        m0, n0 = wr * MT * 32, wc * NT * 32
        q, Gq, i, h = LANE & 15, LANE >> 4, LANE & 31, LANE >> 5
        # DMA offsets of the thread's pieces
        for it in range(2):
            c = it * 256 + tid
            pr = WA * 2 // 16
            row, lp = c // pr, c % pr
            wave.v[it] = row * lda * 2 + swz(pr, row, lp) * 16
            pr = WB * EB // 16
            row, lp = c // pr, c % pr
            wave.v[2 + it] = row * ldb * EB + (swz(pr, row, lp) if EB == 2 else lp) * 16
        for a in range(MT):
            wave.v[4 + a] = [tr_offset(WA, 8 * (int(Gq[l]) >> 1) + (int(q[l]) >> 2), m0 + 32 * a + 16 * (int(Gq[l]) & 1) + 4 * (int(q[l]) & 3)) for l in range(64)]
        if EB == 2:
            for b in range(NT):
                wave.v[8 + b] = [tr_offset(WB, 8 * (int(Gq[l]) >> 1) + (int(q[l]) >> 2), n0 + 32 * b + 16 * (int(Gq[l]) & 1) + 4 * (int(q[l]) & 3)) + A_BYTES for l in range(64)]
        else:
            wave.v[12] = (8 * h * WB + n0 + i) * 4 + A_BYTES
        wave.v[13] = 0x3F803F80
        prog = []
        def set64(reg, val):
            prog.append("s_mov_b32 s%d, %d" % (reg, val & 0xFFFFFFFF)); prog.append("s_mov_b32 s%d, %d" % (reg + 1, val >> 32))
        # prologue: R - 2 chunks in flight
        for c in range(R - nch):
            set64(40, chunk_a(c)); set64(42, chunk_b(c))
            if w < sh.a_waves:
                for it in range(sh.nA):
                    prog += ["s_mov_b32 m0, %d" % (c * BUF + it * 4096 + w * 1024), "global_load_lds_dwordx4 v%d, s[40:41]" % it]
            if w < sh.b_waves:
                for it in range(sh.nB):
                    prog += ["s_mov_b32 m0, %d" % (c * BUF + A_BYTES + it * 4096 + w * 1024), "global_load_lds_dwordx4 v%d, s[42:43]" % (2 + it)]
        prog += zero
        s0 = 0
        if nch == 2:
            for p in range(n_chunks // 2):
                c = 2 * p
                sn0 = s0 - 2 if s0 + R - 2 >= R else s0 + R - 2
                prog += ["s_mov_b32 s20, %d" % (s0 * BUF), "s_mov_b32 s21, %d" % ((s0 + 1) * BUF),
                         "s_mov_b32 s30, %d" % (sn0 * BUF + w * 1024), "s_mov_b32 s31, %d" % ((sn0 + 1) * BUF + w * 1024)]
                set64(22, chunk_a(c + R - 2)); set64(24, chunk_b(c + R - 2)); set64(26, chunk_a(c + R - 1)); set64(28, chunk_b(c + R - 1))
                prog += G.bind(pair_full if w < n_full else pair_other, BIND)
                s0 = 0 if s0 + 2 == R else s0 + 2
            if n_chunks & 1:
                prog += ["s_mov_b32 s20, %d" % (s0 * BUF)] + G.bind(tail, BIND)
        else:
            # SN_DWN_TASK_Q: groups of nch chunks, then the left-over chunks one tail statement at a time
            qbind = dict(BIND, sl2="s32", sl3="s33", md2="s34", md3="s35", ga2="s[44:45]", gb2="s[46:47]", ga3="s[48:49]", gb3="s[50:51]")
            stmt = G.bind(pair_full if w < n_full else pair_other, qbind)
            for g_ in range(n_chunks // nch):
                sn = s0 - nch if s0 >= nch else s0 + R - nch
                for k, (sl, md, ga_, gb_) in enumerate(((20, 30, 22, 24), (21, 31, 26, 28), (32, 34, 44, 46), (33, 35, 48, 50))):
                    prog += ["s_mov_b32 s%d, %d" % (sl, (s0 + k) * BUF), "s_mov_b32 s%d, %d" % (md, (sn + k) * BUF + w * 1024)]
                    set64(ga_, chunk_a(nch * g_ + R - nch + k)); set64(gb_, chunk_b(nch * g_ + R - nch + k))
                prog += stmt
                s0 = 0 if s0 + nch == R else s0 + nch
            for c in range(n_chunks // nch * nch, n_chunks):
                prog += ["s_mov_b32 s20, %d" % (s0 * BUF)] + G.bind(tail, BIND)
                s0 = 0 if s0 + 1 == R else s0 + 1
        prog += ["s_waitcnt vmcnt(0)"]
        programs.append(prog)
    wg.run(programs)
    # results
    M, N = WM * MT * 32, WN * NT * 32
    C = np.zeros((M, N), np.float32)
    bias = np.zeros(M, np.float32)
    for w, wave in enumerate(wg.waves):
        wr, wc = w // WN, w % WN
        m0, n0 = wr * MT * 32, wc * NT * 32
        i, h = LANE & 31, LANE >> 5
        for a in range(MT):
            for b in range(NT):
                for r in range(16):
                    C[m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h, n0 + 32 * b + i] = wave.a[16 * (NT * a + b) + r].view(np.float32)
            if wc == 0:
                bs = wave.v[14 + a].view(np.float32)
                bias[m0 + 32 * a + np.arange(32)] = bs[:32] + bs[32:]
    want = a_val[:, a_col0:a_col0 + M].astype(np.float64).T @ b_val[:, b_col0:b_col0 + N].astype(np.float64)
    want_bias = a_val[:, a_col0:a_col0 + M].astype(np.float64).sum(0)
    return wg, C, want, bias, want_bias


@pytest.mark.parametrize("v,n_chunks", [(0, 11), (0, 16), (1, 9), (2, 8), (3, 13), (4, 11), (5, 21), (5, 32), (6, 10), (7, 15), (7, 26)])
def test_narrow_dw_statements_match_the_contraction(v, n_chunks):
    # A: the columns a 32-wide head block / a full G row starts at; B: the direction half of emb for the 128 x 64 shapes
    a_col0 = 128 if v in (4, 5) else 0
    b_col0 = 64 if v in (3, 7) else 0
    wg, C, want, bias, want_bias = run_task(v, n_chunks, seed=v, a_col0=a_col0, b_col0=b_col0)
    scale = np.abs(want).max()
    assert scale > 1.0
    assert np.abs(C - want).max() <= 2e-6 * scale * np.sqrt(n_chunks * KB), np.abs(C - want).max()
    assert np.abs(bias - want_bias).max() <= 1e-5 * max(1.0, np.abs(want_bias).max())
    # the transpose reads of every bf16 tile shape (256-, 128-, 64-byte rows): conflict-free in the bank model (2 LDS cycles each)
    st = wg.lds.stats.get("ds_read_b64_tr_b16")
    assert st is not None and st[1] == 2 * st[0], st


@pytest.mark.parametrize("v,n_chunks", [(0, 1), (0, 2), (2, 1), (5, 2), (5, 3), (5, 4), (5, 7), (6, 1), (7, 3), (7, 5)])
def test_dw_statements_on_k_ranges_shorter_than_the_ring(v, n_chunks):
    """one to three chunks per K-range (tiny batches: every ring slot but the first is a clamped re-read of the last chunk): the tail
    statement alone, one pair, pair + tail"""
    wg, C, want, bias, want_bias = run_task(v, n_chunks, seed=10 * v + n_chunks, a_col0=128 if v in (4, 5) else 0, b_col0=64 if v in (3, 7) else 0)
    assert np.abs(C - want).max() <= 2e-6 * max(1.0, np.abs(want).max()) * np.sqrt(n_chunks * KB)
    assert np.abs(bias - want_bias).max() <= 1e-5 * max(1.0, np.abs(want_bias).max())


def test_narrow_dw_checker_is_not_vacuous():
    """dropping the counted vmcnt wait of a pair / its barrier / the lgkmcnt wait in front of the second half is caught"""
    gen = H.load_tool("gen_dw_narrow")
    orig = gen.gen_pair
    for pick in (lambda l: l.startswith("s_waitcnt vmcnt"), lambda l: l.startswith("s_barrier"), lambda l: l.startswith("s_waitcnt lgkmcnt")):
        def broken(sh, full, pick=pick):
            lines = orig(sh, full)
            idx = [k for k, l in enumerate(lines) if pick(l)][-1 if pick("s_waitcnt lgkmcnt(0)") else 0]
            return lines[:idx] + lines[idx + 1:]
        import importlib.util, os
        spec = importlib.util.spec_from_file_location("gen_dw_narrow_b", os.path.join(H.ROOT, "tools", "gen_dw_narrow.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.gen_pair = broken
        saved = H.load_tool
        H.load_tool = lambda name, mod=mod: mod
        try:
            with pytest.raises(G.SimError):
                run_task(2, 8, seed=1)
        finally:
            H.load_tool = saved


# ---- the fp32 256 x 256 kernel (tools/gen_dw_f32.py -> csrc/sn_dw_f32.hip): one statement per chunk, 4-slot ring -----------------
def run_task_f32(n_chunks, seed=0, drop=None):
    gen = H.load_tool("gen_dw_f32")
    body = gen.gen()
    if drop is not None:
        idx = [k for k, l in enumerate(body) if drop(l)][0]
        body = body[:idx] + body[idx + 1:]
    A_BYTES, BUF, NBUF = 16384, 32768, 4
    lda = ldb = 256
    rs = np.random.RandomState(seed)
    K = n_chunks * KB
    ga = rs.standard_normal((K, lda)).astype(np.float32)
    gb = rs.standard_normal((K, ldb)).astype(np.float32)
    wg = G.Workgroup(4, lds_bytes=131072)
    wg.mem.add("a", GA_BASE, data=ga.tobytes(), writable=False)
    wg.mem.add("b", GB_BASE, data=gb.tobytes(), writable=False)
    chunk = lambda base, c: base + min(c, n_chunks - 1) * KB * 1024
    bind = dict(bs0="v14", bs1="v15", bs2="v16", bs3="v17", la0="v20", la1="v21", lb0="v22", lb1="v23",
                **{"oa%d" % k: "v%d" % k for k in range(4)}, **{"ob%d" % k: "v%d" % (4 + k) for k in range(4)},
                ga="s[22:23]", gb="s[24:25]", md="s30")
    stmt = G.bind(body, bind)
    programs = []
    for w, wave in enumerate(wg.waves):
        tid = 64 * w + LANE
        i, h = LANE & 31, LANE >> 5
        m0, n0 = (w >> 1) * 128, (w & 1) * 128
        for it in range(4):
            c = it * 256 + tid
            wave.v[it] = (c >> 6) * lda * 4 + (c & 63) * 16
            wave.v[4 + it] = (c >> 6) * ldb * 4 + (c & 63) * 16
        wave.v[24] = h * 1024 + (m0 + i) * 4                    # la
        wave.v[25] = A_BYTES + h * 1024 + (n0 + i) * 4          # lb
        prog = []
        def set64(reg, val):
            prog.append("s_mov_b32 s%d, %d" % (reg, val & 0xFFFFFFFF)); prog.append("s_mov_b32 s%d, %d" % (reg + 1, val >> 32))
        for c in range(NBUF - 1):
            set64(40, chunk(GA_BASE, c)); set64(42, chunk(GB_BASE, c))
            for it in range(4):
                prog += ["s_mov_b32 m0, %d" % (c * BUF + it * 4096 + w * 1024), "global_load_lds_dwordx4 v%d, s[40:41]" % it,
                         "s_mov_b32 m0, %d" % (c * BUF + A_BYTES + it * 4096 + w * 1024), "global_load_lds_dwordx4 v%d, s[42:43]" % (4 + it)]
        prog += ["v_accvgpr_write_b32 a%d, 0" % r for r in range(256)]
        for c in range(n_chunks):
            slot = (c % NBUF) * BUF
            prog += ["v_add_u32 v20, %d, v24" % slot, "v_add_u32 v21, %d, v24" % (slot + 128),
                     "v_add_u32 v22, %d, v25" % slot, "v_add_u32 v23, %d, v25" % (slot + 128),
                     "s_mov_b32 s30, %d" % (((c + NBUF - 1) % NBUF) * BUF + w * 1024)]
            set64(22, chunk(GA_BASE, c + NBUF - 1)); set64(24, chunk(GB_BASE, c + NBUF - 1))
            prog += stmt
        prog += ["s_waitcnt vmcnt(0)"]
        programs.append(prog)
    wg.run(programs)
    C = np.zeros((256, 256), np.float32)
    bias = np.zeros(256, np.float32)
    for w, wave in enumerate(wg.waves):
        i, h = LANE & 31, LANE >> 5
        m0, n0 = (w >> 1) * 128, (w & 1) * 128
        for a in range(4):
            for b in range(4):
                for r in range(16):
                    C[m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h, n0 + 32 * b + i] = wave.a[16 * (4 * a + b) + r].view(np.float32)
            if (w & 1) == 0:
                bs = wave.v[14 + a].view(np.float32)
                bias[m0 + 32 * a + np.arange(32)] = bs[:32] + bs[32:]
    return C, ga.astype(np.float64).T @ gb.astype(np.float64), bias, ga.astype(np.float64).sum(0)


@pytest.mark.parametrize("n_chunks", [3, 6])
def test_fp32_dw_chunk_statement_matches_the_contraction(n_chunks):
    C, want, bias, want_bias = run_task_f32(n_chunks, seed=n_chunks)
    assert np.abs(C - want).max() <= 2e-6 * np.abs(want).max() * np.sqrt(n_chunks * KB)
    assert np.abs(bias - want_bias).max() <= 1e-5 * max(1.0, np.abs(want_bias).max())


def test_fp32_dw_checker_is_not_vacuous():
    for drop in (lambda l: l.startswith("s_waitcnt vmcnt"), lambda l: l.startswith("s_barrier"), lambda l: l.startswith("s_waitcnt lgkmcnt")):
        with pytest.raises(G.SimError):
            run_task_f32(6, seed=1, drop=drop)
