"""The counted-wait protocol of the bf16x3 training kernels, model-checked on the CPU (csrc/sn_mlp_x3.h slab_x3 / x3_store_step).

These two kernels (csrc/sn_mlp_fwd_bf16x3.hip <STORE>, csrc/sn_mlp_bwd_bf16x3.hip) are compiler-scheduled C++ around inline asm, so the
CPU simulator that executes the GENERATED streams (tools/gcn_sim.py) cannot run them; what can be checked without a GPU is the arithmetic
their synchronisation rests on.  A wave's vector-memory operations retire in issue order, and a sync point waits with a COUNTED
``s_waitcnt vmcnt(N)``: every operation but the youngest N has completed.  The weights of slab s+1 are requested (LDS-DMA pieces) during
slab s-1 and must have landed at slab s's visibility sync; N = VMW + (pieces of slab s+2 issued since this slab's first barrier), VMW = 4
= the row stores slab s-1 issued BEHIND its last piece.  Until round 4 the short slabs dealt their stores between their pieces: the wait
then left the tail of slab s+1's weights in flight across the barrier -- a race worth ~1e-5 in a few hundred rays, found on the GPU by a
run-to-run comparison (tools/x3_determinism.py).  This file restates the schedule (the formulas are asserted to be the ones in the
header), walks the vector-memory queue of two consecutive point tiles of both kernels and checks every sync point; the pre-round-4 store
schedule is the negative control."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "sinnerf_amd", "csrc")
B_L0, B_H, B_SKIP, B_DIR, B_D = 64 * 128, 256 * 128, 320 * 128, 288 * 128, 128 * 128


def src(name):
    return open(os.path.join(CSRC, name)).read()


def schedule(nk, gb, nbytes):
    """constants of slab_x3 for one slab (mirrors the constexpr block of csrc/sn_mlp_x3.h)"""
    np_ = nbytes // 4096
    ppk = (np_ + (nk - gb) - 1) // (nk - gb)
    nps = (np_ + ppk - 1) // ppk
    gb2 = nk - 4 if nk >= 8 else gb
    issued = min((gb2 - gb) * ppk, np_)
    lastp = gb + nps - 1 if np_ > 0 else gb
    st0 = max(lastp, nk - 4, gb)
    return dict(np=np_, ppk=ppk, nps=nps, gb2=gb2, issued=issued, lastp=lastp, st0=st0)


def store_steps(nk, st0):
    """x3_store_step: k-step -> row stores issued there (behind the k-step's DMA pieces)"""
    m = nk - st0
    out = {}
    if m == 4 and st0 >= 5:
        for j in range(1, 5):
            out[st0 - 1 + j] = [j - 1]
    else:
        per = (4 + m - 1) // m
        for j in range(m):
            ids = list(range(j * per, min(4, (j + 1) * per)))
            if ids:
                out[st0 + j] = ids
    return out


def store_steps_round3(nk, gb):
    """the schedule these kernels shipped with until round 4 (negative control)"""
    if nk >= 9:
        return {nk - 4 + j: [j] for j in range(4)}
    if nk - gb >= 4:
        return {nk - 4 + j: [j] for j in range(4)}
    return {gb: [0], gb + 1: [1], gb + 2: [2, 3]}


def slab_ops(s, nk, gb, nbytes, vmw, posts, steps_fn):
    """program-order events of slab s: ('wait', N) at a sync point, ('visible', s + 1) where slab s+1's weights must have landed,
    ('piece', target slab), ('store',)"""
    c = schedule(nk, gb, nbytes)
    stores = steps_fn(nk, c["st0"]) if steps_fn is store_steps else steps_fn(nk, gb)
    ev, piece = [], 0
    for ks in range(nk):
        if ks == gb:
            if c["gb2"] == gb:
                ev += [("wait", vmw), ("visible", s + 1)]
        if c["gb2"] != gb and ks == c["gb2"]:
            ev += [("wait", vmw + c["issued"]), ("visible", s + 1)]
        if ks >= gb:
            for _ in range(c["ppk"]):
                if piece < c["np"]:
                    ev.append(("piece", s + 2))
                    piece += 1
            if posts:
                ev += [("store",)] * len(stores.get(ks, []))
    assert piece == c["np"]
    if posts:
        assert sum(len(v) for v in stores.values()) == 4
    return ev


def tile_slabs(kind):
    """(nk, gb, bytes staged = slab two ahead, tile index T, layer-end stores) per slab of one point tile"""
    out = []
    if kind == "forward":
        layers = [(4, 1, B_L0, B_H)]                                                  # SNX_LAYER(4, 0, -1, -1, 1, B_L0, B_H, ..)
        for l in range(1, 8):
            if l == 4:
                layers.append((20, 2, B_SKIP, B_H))
            elif l == 3:
                layers.append((16, 2, B_H, B_SKIP))
            else:
                layers.append((16, 2, B_H, B_H))
        layers.append((16, 2, B_H, B_DIR))                                            # xyz_encoding_final
        for nk, gb, nba, nbb in layers:
            for t in range(8):
                out.append((nk, gb, nba if t < 6 else nbb, t, 4 if t == 7 else 0))
        for t, nb in enumerate((B_DIR, B_DIR, B_L0, B_L0)):                           # dir_encoding: 16 + 2 k-steps
            out.append((18, 2, nb, t, 4 if t == 3 else 0))
    else:
        layers = [(8, 2, B_D, B_H), (16, 2, B_H, B_H)] + [(16, 2, B_H, B_D if li == 1 else B_H) for li in range(7, 0, -1)]
        for nk, gb, nba, nbb in layers:
            for t in range(8):
                out.append((nk, gb, nba if t < 6 else nbb, t, 4 if t == 7 else 0))
    return out


def check(kind, steps_fn):
    """walk two point tiles; return the sync points whose wait does not cover the weights they make visible"""
    slabs = tile_slabs(kind) * 2
    ops, bad = [], []                      # ops: list of ('piece', target) / ('store',) in issue order
    for s, (nk, gb, nb, t, tail) in enumerate(slabs):
        pending_wait = None
        for e in slab_ops(s, nk, gb, nb, 0 if t == 1 else 4, t > 0, steps_fn):
            if e[0] == "wait":
                pending_wait = e[1]
            elif e[0] == "visible":
                need = [i for i, o in enumerate(ops) if o == ("piece", e[1])]
                if need and s >= 2:                                     # (the first two slabs' weights come from the prologue)
                    younger = len(ops) - 1 - need[-1]
                    if younger < pending_wait:
                        bad.append((s, t, nk, pending_wait, younger))
            else:
                ops.append(e)
        ops += [("store",)] * tail                                      # the last tile's rows leave at the layer end
    return bad


def test_the_model_is_the_header():
    h = src("sn_mlp_x3.h")
    for line in ("constexpr int PPK = (NP + (NK - GB) - 1) / (NK - GB);", "constexpr int NPS = (NP + PPK - 1) / PPK;",
                 "constexpr int GB2 = (NK >= 8) ? NK - 4 : GB;", "constexpr int ISSUED = ((GB2 - GB) * PPK < NP) ? (GB2 - GB) * PPK : NP;",
                 "constexpr int LASTP = NP > 0 ? GB + NPS - 1 : GB;",
                 "constexpr int ST0 = (LASTP > NK - 4 ? LASTP : NK - 4) > GB ? (LASTP > NK - 4 ? LASTP : NK - 4) : GB;",
                 "if (m == 4 && st0 >= 5) {", "const int per = (4 + m - 1) / m, j = ks - st0;"):
        assert line in h, line
    f, b = src("sn_mlp_fwd_bf16x3.hip"), src("sn_mlp_bwd_bf16x3.hip")
    for line in ("SNX_LAYER(4, 0, -1, -1, 1, B_L0, B_H, relu_tile, 0);", "SNX_LAYER(4, 16, -1, 1, 2, B_SKIP, B_H, relu_tile, 0);",
                 "SNX_LAYER(16, 0, 0, 0, 2, B_H, B_SKIP, relu_tile, 1);", "SNX_LAYER(16, 0, 1, 1, 2, B_H, B_DIR, copy_tile, 0);",
                 "constexpr int VW_ = (STORE && (T_) != 1) ? 4 : 0;", "SNX_SLAB(3, 16, 2, 0, -1, 2, 2, B_L0, dh, dl, ssp_tile, 0);",
                 "constexpr int B_L0 = 64 * 128, B_H = 256 * 128, B_SKIP = 320 * 128, B_DIR = 288 * 128;"):
        assert line in f, line
    for line in ("SNY_LAYER(8, 0, B_D, B_H, copy_tile, 1);", "SNY_LAYER(16, 1, B_H, B_H, mask_sigma_tile, 0);",
                 "if (li == 1) SNY_LAYER(16, 0, B_H, B_D, mask_tile, 1);", "constexpr int VW_ = ((T_) != 1) ? 4 : 0;",
                 "constexpr int B_D = 128 * 128, B_H = 256 * 128;"):
        assert line in b, line
    assert len(re.findall(r"slab_x3<NK_, 0, SET_, SET_, 2, 0, NB_, VW_>", b)) == 2           # the chain's slabs: GB = 2


def test_every_sync_point_covers_the_weights_it_makes_visible():
    for kind in ("forward", "chain"):
        assert check(kind, store_steps) == [], kind


def test_the_round3_store_schedule_is_caught():
    """stores dealt between the pieces of the 4-k-step (forward, layer 0) and 8-k-step (chain, dir_encoding^T) slabs: the sync points of
    the slabs behind them wait for too little"""
    fwd, chain = check("forward", store_steps_round3), check("chain", store_steps_round3)
    assert fwd and all(nk == 4 for (_, _, nk, _, _) in fwd)
    assert chain and all(nk == 8 for (_, _, nk, _, _) in chain)


def test_row_groups_are_read_after_the_epilogue_filled_the_tile():
    """rd(0) never before k-step 4 (the epilogue blocks of the previous tile land behind k-steps 0..3) in the slabs that read a row group
    one k-step ahead of its store"""
    for nk, gb, nb in ((16, 2, B_H), (16, 2, B_SKIP), (16, 2, B_DIR), (18, 2, B_L0), (18, 2, B_DIR), (20, 2, B_H), (20, 2, B_SKIP)):
        c = schedule(nk, gb, nb)
        assert nk - c["st0"] == 4 and c["st0"] - 1 >= 4, (nk, c)
        assert c["st0"] >= c["lastp"]
    for nk, gb, nb in ((4, 1, B_L0), (4, 1, B_H), (8, 2, B_D), (8, 2, B_H)):
        c = schedule(nk, gb, nb)
        assert c["st0"] >= c["lastp"] and c["st0"] >= gb and c["st0"] < nk


def test_weight_gradient_loop_computes_every_chunk_once():
    """The bf16x3 chunk loop of csrc/sn_dw.hip (run_task, X3 branch) restated: the gathers of chunk c go out in front of the MFMAs of chunk
    c-1 (two register sets, unrolled by two, an odd step count evened out by one peeled step whose set is copied): every chunk 0 .. n-1 is
    multiplied exactly once, from the set that holds it, for every chunk count."""
    h = src("sn_dw.hip")
    for line in ("step(1, ra1, rb1, ra0, rb0);", "step(c, ra1, rb1, ra0, rb0);", "step(c + 1, ra0, rb0, ra1, rb1);", "for (; c < n_chunks; c += 2) {",
                 "if (n_chunks & 1) {", "gather_a(smem, ra0);"):
        assert line in h, line
    for n in range(1, 40):
        sets, done = {0: 0, 1: None}, []                      # register set -> chunk it holds (after the prologue: chunk 0 in set 0)

        def step(c, g, k):
            sets[g] = c                                       # gather(chunk c) -- chunk n is the staged clamp copy nobody multiplies
            done.append(sets[k])                              # compute(chunk c - 1)
            assert sets[k] == c - 1, (n, c)
        c = 1
        if n & 1:
            step(1, 1, 0)
            sets[0] = sets[1]
            c = 2
        while c < n:
            step(c, 1, 0)
            step(c + 1, 0, 1)
            c += 2
        assert done == list(range(n)), (n, done)
