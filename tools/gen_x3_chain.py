#!/usr/bin/env python3
"""Generator of the hand-scheduled slab loop of the bf16x3 backward chain (sinnerf_amd/csrc/sn_mlp_bwd_bf16x3_t.hip).

The chain = input-gradient propagation g_x = W^T g_y, g_y = g_h (.) act'(.) through dir_encoding^T, xyz_encoding_final^T and
xyz_encoding_8..2^T -- what torch autograd derives from models/nerf.py:122-148 -- for one wave's ONE 32-point tile in the 3-term split
arithmetic of csrc/sn_mlp_x3.h (W^T g ~= Wh^T gh + Wl^T gh + Wh^T gl on v_mfma_f32_32x32x16_bf16, two accumulator chains in strict
alternation A B A | B A B, both started from the literal 0): the 72 transposed (hi, lo) weight slabs of csrc/sn_layout.h
("Backward-chain blob, bf16x3"), 3 264 MFMAs.  Same arithmetic and accumulation order, same stored gradient state (G slots 0..8 as
(hi, lo) pairs) as the compiler-scheduled mlp_bwd_chain_bf16x3_kernel it replaces (bit-identical on the device), laid out by the list
scheduler of tools/gen_bf16_trunk.py (class Gen) like the training forward (tools/gen_x3_trunk.py); executed on the CPU by
tools/gcn_sim.py (tests/test_streams_cpu.py).

Per output tile (slab s = tile t of chain layer L) the deferred epilogue, run inside slab s + 1, per block of four accumulator registers:
  x = A + B (v_add_f32); [xyz_encoding_final^T: the sigma head's term x += sigma.weight[f] g_sigma (nerf.py:136), v_fmac_f32];
  [ReLU mask from the SIGN WORD the training forward left for this (layer, tile pair): v_bfe_i32 + v_and_b32 per value]; hi = cvt_pk(v)
  -> AGPRs of the other activation set; lo = cvt_pk(v - float(hi)) -> AGPRs; two v_permlane32_swap_b32 + ONE ds_write_b128 into the
  wave's staging tile (32 rows x 128 B, chunk ^= row & 7: conflict-free both ways); then the tile's row stores: 4 x (ds_read_b128 +
  global_store_dwordx4 nt of 8 whole 128-byte point segments of G[slot]).
Sign words: the 8 x 4 words of a point tile are loaded by the kernel in front of the statement (eight 16-byte loads per lane) and
handed in as operands -- no load inside the slab loop (the compiler-scheduled kernel requested each layer's words a layer ahead and
took them over with a hand-counted vmcnt).
Weight ring: FOUR slots of 32 KB (the widest transposed slab; the chain has no bias table in front of the ring to make room for),
slab s in slot s % 4 (72 = 0 mod 4: static), staged 3 slabs ahead behind ONE sync point per slab (counted vmcnt for slab s + 1 derived
from the emitted order, then the barrier, then the pieces of slab s + 3); slabs 69..71 stage the next point tile's slabs 0..2 (the
stream wraps).

Register plan inside the statement (v[128:255] named as clobbers; the kernel keeps v0..v127, where the sign words live):
  v[128:191] accumulators [set][chain][16]     v[192:223] A-fragment ring: 4 entries x (hi 4 + lo 4)
  v[224:227] sigma^T weights of the block in flight    v[228:231], v[232:235] row buffers    v[236:237] float(hi) temporaries
  s[84:85] running pointer into G[slot] (- slot_rows * 1024 B per layer)

usage: gen_x3_chain.py out.inc [knob=value ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_bf16_trunk as T                      # noqa: E402  (Gen, Filler, write_inc)

KNOBS = dict(prefetch=3, cap=5.0, dma_cost=2.0, valu_cost=1.0, lds_cost=1.0, salu_cost=0.5,
             bar_gap=3,         # the sync point sits behind this MFMA of a slab
             epi_from=1,
             pk=0)              # 1: v_pk_add_f32 (see tools/gen_x3_trunk.py: an anti-lever beside MFMAs)

N_SLABS = 72
N_SLOTS, SLOT_BYTES, DMA_DIST = 4, 32768, 3
ACC = lambda st, ch: 128 + st * 32 + ch * 16
RING0, RING_N = 192, 4
SIGT = 224
ROW_A, ROW_B = 228, 232
TMP0 = 236
SGPR_G = 84


def nk_of(s): return 8 if s < 8 else 16
def layer_of(s): return s // 8                # 0 dir_encoding^T, 1 xyz_encoding_final^T, L >= 2: xyz_encoding_{10 - L}^T
def slab_bytes(s): return nk_of(s) * 2048
def read_set(L): return L & 1
def write_set(L): return 1 - (L & 1)
def out_slot(L): return 8 if L == 0 else 8 - L            # G slot the layer writes = acts slot of its ReLU mask (L >= 1)
def x3_reg(st, part, ks): return st * 128 + part * 64 + ks * 4
TOTAL_BYTES = sum(slab_bytes(s) for s in range(N_SLABS))


def gen(knobs):
    K = knobs
    g = T.Gen(dict(T.KNOBS, store=1, cap=K["cap"]))
    g.vm = [1] * 4 + [2] * 4         # entry: at most the pieces of slabs 1, 2 in flight (older operations only make the first waits stricter)
    D = K["prefetch"]
    assert 1 <= D <= RING_N - 1
    ring = lambda kidx: RING0 + 8 * (kidx % RING_N)

    mf, gk, first, kstep_list = [], {}, {}, []
    for s in range(N_SLABS):
        first[s] = len(mf)
        for ks in range(nk_of(s)):
            gk[(s, ks)] = len(kstep_list)
            kstep_list.append((s, ks))
            for term in range(3):
                mf.append((s, ks, term))
    first[N_SLABS] = len(mf)
    idx_of = {m: i for i, m in enumerate(mf)}

    fillers = []
    seq = [0]
    def add(f):
        f.seq = seq[0]; seq[0] += 1
        fillers.append(f)
        return f

    bar = {s: first[s] + K["bar_gap"] - 1 for s in range(N_SLABS)}

    def frag_addr(s, ks, part):
        slot = s % N_SLOTS
        return ("%[vaA]", "%[vaB]")[slot >> 1], (slot & 1) * SLOT_BYTES + ks * 2048 + part * 1024

    # ---- A fragments (the first D k-steps are loaded in the preamble: slab 0 is resident)
    for kidx, (s, ks) in enumerate(kstep_list):
        if kidx < D:
            continue
        use = idx_of[(s, ks, 0)]
        ps, pk = kstep_list[kidx - RING_N] if kidx >= RING_N else (None, None)
        prev_user = idx_of[(ps, pk, 2)] if ps is not None else -1
        ws, wk = kstep_list[kidx - D]
        rel = max(prev_user, idx_of[(ws, wk, 0)])
        if s > 0:
            rel = max(rel, bar[s - 1] + 1)                           # the sync point of slab s - 1 makes slab s visible
        for part in range(2):
            r = ring(kidx) + 4 * part
            base, off = frag_addr(s, ks, part)
            add(T.Filler("ds_read_b128 v[%d:%d], %s offset:%d" % (r, r + 3, base, off), K["lds_cost"], rel, use - 1, "ds_read",
                         tag=("frag", kidx, part)))

    # ---- epilogue of slab s (tile t of chain layer L), run inside slab s + 1
    def epilogue(s):
        L, t = layer_of(s), s % 8
        W, st = write_set(L), s & 1
        copy, sig = L == 0, L == 1
        slot = out_slot(L)
        items = []
        rows_n = 2
        for i in range(4):
            a, b = ACC(st, 0) + 4 * i, ACC(st, 1) + 4 * i
            rh = x3_reg(W, 0, 2 * t + (i >> 1)) + 2 * (i & 1)
            rl = x3_reg(W, 1, 2 * t + (i >> 1)) + 2 * (i & 1)
            V = lambda text, writes=(), tag=None: items.append(("valu", text, writes, tag, "acc"))
            if sig:
                items.append(("ds_read", "ds_read_b128 v[%d:%d], %%[vs] offset:%d" % (SIGT, SIGT + 3, (16 * t + 4 * i) * 4), (), ("sigt", s, i), "acc"))
            if K["pk"]:
                V("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (a, a + 1, a, a + 1, b, b + 1), (a, a + 1))
                V("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (a + 2, a + 3, a + 2, a + 3, b + 2, b + 3), (a + 2, a + 3))
            else:
                for e in range(4):
                    V("v_add_f32 v%d, v%d, v%d" % (a + e, a + e, b + e), (a + e,))
            if sig:                                                  # x += sigma.weight[f] * g_sigma  (fused, as __builtin_fmaf in the C++ kernel)
                for e in range(4):
                    V("v_fmac_f32 v%d, v%d, %%[gsig]" % (a + e, SIGT + e), (a + e,), ("sigt", s, i))
            if not copy:                                             # g_y = g_h [h > 0]: bit k <- value 0, 16 + k <- value 1, k + 1 / 17 + k <- values 2 / 3
                k = 2 * i + 8 * (t & 1)
                word = "%%[sw%d_%d]" % (slot, t >> 1)
                for e, bit in enumerate((k, k + 16, k + 1, k + 17)):
                    V("v_bfe_i32 v%d, %s, %d, 1" % (b + e, word, bit), (b + e,))
                for e in range(4):
                    V("v_and_b32 v%d, v%d, v%d" % (a + e, b + e, a + e), (a + e,))
            V("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (b, a, a + 1), (b,))
            V("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (b + 1, a + 2, a + 3), (b + 1,))
            V("v_accvgpr_write_b32 a%d, v%d" % (rh, b), (("a", rh),))
            V("v_accvgpr_write_b32 a%d, v%d" % (rh + 1, b + 1), (("a", rh + 1),))
            for half in range(2):
                V("v_lshlrev_b32 v%d, 16, v%d" % (TMP0, b + half), (TMP0,))
                V("v_and_b32 v%d, 0xffff0000, v%d" % (TMP0 + 1, b + half), (TMP0 + 1,))
                x = a + 2 * half
                if K["pk"]:
                    V("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d] neg_lo:[0,1] neg_hi:[0,1]" % (x, x + 1, x, x + 1, TMP0, TMP0 + 1), (x, x + 1))
                else:
                    V("v_sub_f32 v%d, v%d, v%d" % (x, x, TMP0), (x,))
                    V("v_sub_f32 v%d, v%d, v%d" % (x + 1, x + 1, TMP0 + 1), (x + 1,))
            V("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (b + 2, a, a + 1), (b + 2,))
            V("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (b + 3, a + 2, a + 3), (b + 3,))
            V("v_accvgpr_write_b32 a%d, v%d" % (rl, b + 2), (("a", rl),))
            V("v_accvgpr_write_b32 a%d, v%d" % (rl + 1, b + 3), (("a", rl + 1),))
            items.append(("swap", "v_permlane32_swap_b32 v%d, v%d" % (b, b + 2), (b, b + 2), None, "acc", (b, b + 2)))
            items.append(("swap", "v_permlane32_swap_b32 v%d, v%d" % (b + 1, b + 3), (b + 1, b + 3), None, "acc", (b + 1, b + 3)))
            items.append(("ds_write", "ds_write_b128 %%[stw%d], v[%d:%d]" % (i, b, b + 3), (), None, "acc"))
        rows = []
        def rd(n):
            ro = ROW_A if (n % rows_n) == 0 else ROW_B
            rows.append(("ds_read", "ds_read_b128 v[%d:%d], %%[str] offset:%d" % (ro, ro + 3, 1024 * n), (), ("ro", s, n), "post"))
        def st_(n):
            ro = ROW_A if (n % rows_n) == 0 else ROW_B
            rows.append(("vstore", "global_store_dwordx4 %%[vo], v[%d:%d], s[%d:%d] offset:%d nt" % (ro, ro + 3, SGPR_G, SGPR_G + 1, 128 * t),
                         (), ("ro", s, n), "post", (SGPR_G, SGPR_G + 1)))
            if n < 3:
                rows.append(("valu", "v_add_u32 %[vo], 8192, %[vo]", ("vo",), None, "post"))
            else:
                rows.append(("valu", "v_subrev_u32 %[vo], 24576, %[vo]", ("vo",), None, "post"))
        rd(0); rd(1)
        for n in range(4):
            st_(n)
            if n + 2 < 4:
                rd(n + 2)
        if t == 7 and L < 8:                                         # next layer: G[slot - 1]
            rows.append(("salu", "s_sub_u32 s%d, s%d, %%[srlo]" % (SGPR_G, SGPR_G), (SGPR_G,), None, "post"))
            rows.append(("salu", "s_subb_u32 s%d, s%d, %%[srhi]" % (SGPR_G + 1, SGPR_G + 1), (SGPR_G + 1,), None, "post"))
        return items + rows

    COST = {"ds_read": K["lds_cost"], "ds_write": K["lds_cost"], "valu": K["valu_cost"], "swap": K["valu_cost"],
            "vstore": K["dma_cost"], "salu": K["salu_cost"]}
    def as_filler(item, rel, dl):
        kind, text, writes, tag, _cls = item[:5]
        reads = item[5] if len(item) > 5 else ()
        return T.Filler(text, COST[kind], rel, dl, kind, reads=reads, writes=writes, tag=tag)

    epi_tail, epi_fillers = [], []
    for s in range(N_SLABS):
        L, t = layer_of(s), s % 8
        flat = epilogue(s)
        if s + 1 >= N_SLABS:
            epi_tail = flat
            continue
        rel0 = first[s + 1] + K["epi_from"]
        n_gaps = first[s + 2] - first[s + 1]
        hard_dl = first[s + 2] - 1 if s + 2 < N_SLABS else len(mf) - 1
        post_dl = first[s + 3] - 1 if s + 3 < N_SLABS else len(mf) - 1
        dl = hard_dl
        if t == 7 and L + 1 <= 8:                                   # the next layer's first slab reads k-steps 14, 15 of the written set
            dl = min(dl, idx_of[(8 * (L + 1), 14, 0)] - 2)
        per_gap = max(2, -(-len(flat) // max(1, n_gaps - 4)))
        for j, item in enumerate(flat):
            d = dl if item[4] == "acc" else max(dl, post_dl)
            epi_fillers.append(add(as_filler(item, min(rel0 + j // per_gap, d), d)))
    for a, b in zip(reversed(epi_fillers[:-1]), reversed(epi_fillers[1:])):      # one program-ordered stream: deadlines never decrease
        if a.deadline > b.deadline:
            a.deadline = b.deadline
            a.release = min(a.release, a.deadline)

    # ---- sync point + weight stream: slab s + 1 awaited, then the pieces of slab s + 3 (s >= 69: the next point tile's slabs 0..2)
    for s in range(N_SLABS):
        add(T.Filler("", 0.5, bar[s], bar[s], "bar", tag=s + 1))
        tgt = s + DMA_DIST
        real = tgt % N_SLABS
        pieces = slab_bytes(real) // 4096
        span = first[s + 1] - 2 - bar[s] - 1
        stride = max(1, span // pieces)
        dl = first[s + 1] - 1
        for p in range(pieces):
            rel = min(bar[s] + 1 + p * stride, dl)
            add(T.Filler("s_add_u32 m0, %%[sm], %d" % ((real % N_SLOTS) * SLOT_BYTES + p * 4096), K["salu_cost"], rel, dl, "m0"))
            if tgt >= N_SLABS and real == 0 and p == 0:               # the stream wraps: back to the start of the blob
                add(T.Filler("v_subrev_u32 %%[goff], %d, %%[goff]" % TOTAL_BYTES, K["valu_cost"], rel, dl, "valu", writes=("goff",)))
            add(T.Filler("global_load_lds_dwordx4 %[goff], %[blob]", K["dma_cost"], rel, dl, "dma", tag=tgt))
            add(T.Filler("v_add_u32 %[goff], 4096, %[goff]", K["valu_cost"], rel, dl, "valu", writes=("goff",)))

    # ---- emission ------------------------------------------------------------------------------------------------------
    fillers.sort(key=lambda f: (f.release, f.seq))
    pending, fi = [], 0
    for dst, src in ((SGPR_G, "gplo"), (SGPR_G + 1, "gphi")):
        g.emit("s_mov_b32 s%d, %%[%s]" % (dst, src))
        g.last_salu_write[dst] = g.n_states - 1
    for kidx in range(D):
        s, ks = kstep_list[kidx]
        for part in range(2):
            r = ring(kidx) + 4 * part
            base, off = frag_addr(s, ks, part)
            g.emit("ds_read_b128 v[%d:%d], %s offset:%d" % (r, r + 3, base, off)); g.lgkm.append(("frag", kidx, part))

    def pop_ready():
        nonlocal pending
        pending.sort(key=lambda f: (f.deadline, f.seq))
        budget, n = K["cap"], 0
        for f in pending:
            if budget < f.cost - 1e-9:
                break
            g.run_filler(f); budget -= f.cost; n += 1
        pending = pending[n:]

    for i, (s, ks, term) in enumerate(mf):
        while fi < len(fillers) and fillers[fi].release <= i - 1:
            pending.append(fillers[fi]); fi += 1
        pending.sort(key=lambda f: (f.deadline, f.seq))
        keep = []
        for f in pending:
            if f.deadline <= i - 1:
                g.run_filler(f); g.stats["forced"] += 1
            else:
                keep.append(f)
        pending = keep
        kidx = gk[(s, ks)]
        part = 1 if term == 1 else 0
        if term == 0:
            g.wait_lgkm({("frag", kidx, 0), ("frag", kidx, 1)})      # one counted wait per k-step (the lo read follows the hi read)
        st = s & 1
        c0, c1 = (0, 1) if (ks & 1) == 0 else (1, 0)
        d = ACC(st, c1 if term == 1 else c0)
        c_txt = "0" if (ks == 0 and term <= 1) else "v[%d:%d]" % (d, d + 15)      # both chains start from zero (no bias in the chain)
        bpart = 1 if term == 2 else 0
        b0 = x3_reg(read_set(layer_of(s)), bpart, ks)
        g.pad_valu_to_mfma([("a", b0 + e) for e in range(4)])
        a_reg = ring(kidx) + 4 * part
        g.emit("v_mfma_f32_32x32x16_bf16 v[%d:%d], v[%d:%d], a[%d:%d], %s" % (d, d + 15, a_reg, a_reg + 3, b0, b0 + 3, c_txt), states=8)
        g.mfma_count += 1
        while fi < len(fillers) and fillers[fi].release <= i:
            pending.append(fillers[fi]); fi += 1
        pop_ready()

    while fi < len(fillers):
        pending.append(fillers[fi]); fi += 1
    pending.sort(key=lambda f: (f.deadline, f.seq))
    for f in pending:
        g.run_filler(f)
    g.nop(20)
    for item in epi_tail:
        g.run_filler(as_filler(item, 0, 0))
    if g.lgkm:
        g.emit("s_waitcnt lgkmcnt(0)")
        g.lgkm = []
    return g


def main():
    out_path = sys.argv[1]
    knobs = dict(KNOBS)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        knobs[k] = type(KNOBS[k])(float(v)) if isinstance(KNOBS[k], float) else int(v)
    g = gen(knobs)
    T.write_inc(out_path, g, "SN_X3_CHAIN", "tools/gen_x3_chain.py " + " ".join(sys.argv[2:]))
    n_other = len(g.out) - g.mfma_count
    print("x3 chain: %d MFMAs, %d other (%.2f / MFMA), nops %d, waits %d, forced %d"
          % (g.mfma_count, n_other, n_other / g.mfma_count, g.stats["nop"], g.stats["wait"], g.stats["forced"]))


if __name__ == "__main__":
    main()
