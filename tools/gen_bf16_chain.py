#!/usr/bin/env python3
"""Generator of the hand-scheduled slab loop of the bf16-state backward chain (sinnerf_amd/csrc/sn_mlp_bwd_bf16_t.hip).

The chain = input-gradient propagation g_x = W^T g_y, g_y = g_h (.) act'(.) through dir_encoding^T, xyz_encoding_final^T and
xyz_encoding_8..2^T -- what torch autograd derives from models/nerf.py:122-148 -- for one wave's two 32-point tiles: the 72
transposed weight slabs of csrc/sn_layout.h ("Backward-chain blob, bf16 operands"), 2176 v_mfma_f32_32x32x16_bf16.  Like the
training forward (tools/gen_bf16_trunk.py store=1) it is emitted as ONE asm statement whose instruction stream is laid out by
the same list scheduler (class Gen): MFMAs back to back, everything else dealt into their shadows with counted waits.

Per output tile (slab s, tile t of a layer) the deferred epilogue, run inside slab s+1:
  * xyz_encoding_final^T only: the sigma head's term x += sigma.weight[f] * g_sigma (nerf.py:136) on the fp32 accumulators;
  * conversion to packed bf16 pairs IN PLACE over accumulator blocks 0 / 1 of the point tile (as in the forward's store mode);
  * ReLU mask from the SIGN WORD the training forward left for this (layer, tile): shift the pair's two bits down, isolate
    (c01 = 0x00010001), subtract 1 per half -> 0xffff where the forward value was positive, AND (csrc/sn_mlp_bf16.h
    epi_relu_bits; step order: point tile outermost, except layer 8 whose forward epilogue runs quad outermost);
  * hand-over to the next transposed layer: v_accvgpr_write into the other activation set;
  * the masked pairs ARE the pre-activation gradients g_y the weight-gradient kernels read: four v_permlane32_swap_b32, two
    conflict-free ds_write_b128 into the wave's staging planes, every second tile eight ds_read_b128 + global_store_dwordx4 of
    whole 128-byte rows of G[slot] (non-temporal).
Sign words: one dword per lane and tile, ALL 64 of a point tile loaded at the top of the statement into a register file
(v[64:127]) behind which the eight short dir_encoding^T slabs run before the first mask is needed.  Loads inside the slab loop --
even a layer ahead of their use -- sit in the same issue-ordered vmcnt queue as the weight DMA: every barrier's counted wait for
the next slab's pieces then also waits for the youngest sign-word load in front of them, an HBM read queued behind 4 TB/s of row
stores (measured: the chain lost 0.23 ms to its stores, the forward -- no loads in its trunk -- 0.13; TCP->TCC write latency is
only ~320 cycles, so it is not the stores themselves that hold the queue).

Weight ring: 4 slots of 16 KB (the widest transposed slab), slab s in slot s % 4 (72 = 0 mod 4: static), staged 3 slabs ahead,
slabs 69..71 stage the next point tile's slabs 0..2 (the stream wraps).  No bias: the first k-step of a slab takes C = 0.

Register plan inside the statement (v[128:255] declared as clobbers where named):
  v[128:191] accumulators [set][point tile][16]      v[192:199] two 4-register rows in flight (staging read -> global store)
  v[200:207] sigma^T weights (2 x 4, double-buffered)  v[208:231] A-fragment ring (6 entries)
  v232/v233 mask temporaries   v234 sign-load offset   v235 odd-row staging read address   v[64:127] sign-word file [acts slot][tile]
  s[84:85] running pointer into G[slot] (- slot_rows * 512 B per layer), s[86:87] pointer to the sign-word rows

usage: gen_bf16_chain.py out.inc [knob=value ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_bf16_trunk as T                      # noqa: E402  (Gen, Filler, accumulator / AGPR numbering, staging-plane constants)

KNOBS = dict(prefetch=4, cap=6.0, dma_cost=2.0, valu_cost=1.0, lds_cost=1.0, salu_cost=0.5, bar_gap=3,
             abl_vstore=1, abl_stage=1, abl_mask=1)           # timing ablations (0 = leave out: WRONG results)

N_SLABS = 72
N_SLOTS, SLOT_BYTES, DMA_DIST = 4, 16384, 3
RO, SIGT, M0, VSG, STR1, SWF = 192, 200, 232, 234, 235, 64
SGPR_G, SGPR_SIGN = 84, 86
ACC, RING0, act_reg = T.ACC, T.RING0, T.act_reg
ST_PT, ST_B3, ST_E = T.ST_PT, T.ST_B3, T.ST_E


def nk_of(s): return 8 if s < 8 else 16
def layer_of(s): return s // 8                # 0 dir_encoding^T, 1 xyz_encoding_final^T, 2.. xyz_encoding_{li+1}^T with li = 9 - layer
def slab_bytes(s): return nk_of(s) * 1024
def read_set(L): return 0 if L == 0 else 1 if L == 1 else (0 if (9 - L) & 1 else 1)
def write_set(L): return 1 - read_set(L)
def out_slot(L): return 8 if L == 0 else 7 if L == 1 else 8 - L       # G slot the layer writes = acts slot of its ReLU mask
T_out_slot = out_slot
TOTAL_BYTES = sum(slab_bytes(s) for s in range(N_SLABS))


def gen(knobs):
    K = knobs
    g = T.Gen(dict(T.KNOBS, store=1, **{k: v for k, v in K.items() if k in T.KNOBS}))
    g.vm = [1, 1, 2, 2]                           # entry: at most the two pieces each of slabs 1, 2 in flight (older kernel operations
                                                  # still in flight only make the first counted waits stricter)
    D = K["prefetch"]
    R = D + 2
    assert R <= 6
    ring = lambda kidx: RING0 + 4 * (kidx % R)

    mf, gk, first = [], {}, {}
    kc = 0
    for s in range(N_SLABS):
        first[s] = len(mf)
        for ks in range(nk_of(s)):
            gk[(s, ks)] = kc
            kc += 1
            for pt in range(2):
                mf.append((s, ks, pt))
    first[N_SLABS] = len(mf)
    kstep_list = [(s, ks) for s in range(N_SLABS) for ks in range(nk_of(s))]
    idx_of = {m: i for i, m in enumerate(mf)}

    fillers = []
    seq = [0]
    def add(f):
        f.seq = seq[0]; seq[0] += 1
        fillers.append(f)

    def frag_addr(s, ks):
        return (s % N_SLOTS) * SLOT_BYTES + ks * 1024

    # ---- A fragments (first D k-steps in the preamble: slab 0 is resident)
    for kidx, (s, ks) in enumerate(kstep_list):
        if kidx < D:
            continue
        use = idx_of[(s, ks, 0)]
        prev_user = idx_of[(kstep_list[kidx - R][0], kstep_list[kidx - R][1], 1)] if kidx - R >= 0 else -1
        want = idx_of[(kstep_list[kidx - D][0], kstep_list[kidx - D][1], 0)]
        rel = max(prev_user, want)
        bar_ok = first[s - 1] + K["bar_gap"] if s > 0 else -1
        rel = max(rel, bar_ok + 1)
        add(T.Filler("ds_read_b128 v[%d:%d], %%[va0] offset:%d" % (ring(kidx), ring(kidx) + 3, frag_addr(s, ks)), K["lds_cost"], rel,
                     use - 1, "ds_read", tag=("frag", kidx)))

    # ---- sign words: the word of (chain layer L >= 1, tile t) = acts slot 8 - L ... v[SWF + 8 slot + t], loaded in the preamble
    swreg = lambda L, t: SWF + 8 * T_out_slot(L) + t
    # ---- epilogue of slab s (run inside slab s+1)
    def epilogue(s):
        L, t = layer_of(s), s % 8
        W, st = write_set(L), s & 1
        copy, sig = L == 0, L == 1
        PKR = lambda pt, n: ACC(st, pt) + n
        slot_of = {0: (0, 1), 2: (2, 3), 1: (4, 5), 3: (6, 7)}
        sigt = lambda i: SIGT + 4 * (i & 1)
        items = []
        def sig_load(i):
            items.append(("ds_read", "ds_read_b128 v[%d:%d], %%[vst] offset:%d" % (sigt(i), sigt(i) + 3, (16 * t + 4 * i) * 4), (), ("sigt", s, i), "acc"))
        def block(pt, i):
            a = ACC(st, pt) + 4 * i
            t0, t1 = PKR(pt, slot_of[i][0]), PKR(pt, slot_of[i][1])
            q = 2 * i
            r0 = act_reg(W, 2 * t + (q >> 2), pt) + (q & 3)
            if sig:                                         # + sigma.weight[f] * g_sigma (nerf.py:136), fp32
                for e in range(4):
                    items.append(("valu", "v_fmac_f32 v%d, v%d, %%[gs%d]" % (a + e, sigt(i) + e, pt), (a + e,), ("sigt", s, i), "acc"))
            items.append(("valu", "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (t0, a, a + 1), (t0,), None, "acc"))
            items.append(("valu", "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (t1, a + 2, a + 3), (t1,), None, "acc"))
            if not copy and K["abl_mask"]:
                j0 = (4 * i + 2 * pt) if sig else (8 * pt + 2 * i)
                sw = swreg(L, t)
                items.append(("valu", "v_lshrrev_b32 v%d, %d, v%d" % (M0, j0, sw), (M0,), ("vm", ("sw", out_slot(L), t)), "acc"))
                items.append(("valu", "v_lshrrev_b32 v%d, %d, v%d" % (M0 + 1, j0 + 1, sw), (M0 + 1,), None, "acc"))
                items.append(("valu", "v_and_b32 v%d, %%[c01], v%d" % (M0, M0), (M0,), None, "acc"))
                items.append(("valu", "v_and_b32 v%d, %%[c01], v%d" % (M0 + 1, M0 + 1), (M0 + 1,), None, "acc"))
                items.append(("valu", "v_pk_sub_u16 v%d, v%d, %%[c01]" % (M0, M0), (M0,), None, "acc"))
                items.append(("valu", "v_pk_sub_u16 v%d, v%d, %%[c01]" % (M0 + 1, M0 + 1), (M0 + 1,), None, "acc"))
                items.append(("valu", "v_and_b32 v%d, v%d, v%d" % (t0, t0, M0), (t0,), None, "acc"))
                items.append(("valu", "v_and_b32 v%d, v%d, v%d" % (t1, t1, M0 + 1), (t1,), None, "acc"))
            items.append(("valu", "v_accvgpr_write_b32 a%d, v%d" % (r0, t0), (("a", r0),), None, "acc"))
            items.append(("valu", "v_accvgpr_write_b32 a%d, v%d" % (r0 + 1, t1), (("a", r0 + 1),), None, "acc"))
        def finish_pt(pt):
            for x, y in [(PKR(pt, 0), PKR(pt, 2)), (PKR(pt, 1), PKR(pt, 3)), (PKR(pt, 4), PKR(pt, 6)), (PKR(pt, 5), PKR(pt, 7))]:
                items.append(("swap", "v_permlane32_swap_b32 v%d, v%d" % (x, y), (x, y), None, "acc", (x, y)))
            for e in range(2):
                off = pt * ST_PT + (t & 1) * ST_B3 + e * ST_E
                items.append(("ds_write", "ds_write_b128 %%[stw], v[%d:%d] offset:%d" % (PKR(pt, 4 * e), PKR(pt, 4 * e) + 3, off), (), None, "acc"))
        if sig:
            sig_load(0); sig_load(1)
            for i in range(4):
                for pt in range(2):
                    block(pt, i)
                if i + 2 < 4:
                    sig_load(i + 2)
            finish_pt(0); finish_pt(1)
        else:
            for pt in range(2):
                for i in range(4):
                    block(pt, i)
                finish_pt(pt)
        readout = []
        if t & 1:
            main_items, items = items, readout
            tp = t >> 1
            rows = [(pt, i) for pt in range(2) for i in range(4)]
            def rd(n):
                pt, i = rows[n]
                ro = RO + 4 * (n % 2)
                src = "v%d" % STR1 if (i & 1) else "%[str0]"
                items.append(("ds_read", "ds_read_b128 v[%d:%d], %s offset:%d" % (ro, ro + 3, src, pt * ST_PT + 256 * i), (), ("ro", s, n), "post"))
            def stw(n):
                ro = RO + 4 * (n % 2)
                items.append(("vstore", "global_store_dwordx4 %%[vo], v[%d:%d], s[%d:%d] offset:%d nt" % (ro, ro + 3, SGPR_G, SGPR_G + 1, 128 * tp),
                              (), ("ro", s, n), "post", (SGPR_G, SGPR_G + 1)))
                items.append(("valu", ("v_add_u32 %[vo], 4096, %[vo]" if n < 7 else "v_subrev_u32 %[vo], 28672, %[vo]"), ("vo",), None, "post"))
            rd(0); rd(1)
            for n in range(8):
                stw(n)
                if n + 2 < 8:
                    rd(n + 2)
            if t == 7:                                       # next layer: G[slot - 1]
                items.append(("salu", "s_sub_u32 s%d, s%d, %%[srlo]" % (SGPR_G, SGPR_G), (SGPR_G,), None, "post"))
                items.append(("salu", "s_subb_u32 s%d, s%d, %%[srhi]" % (SGPR_G + 1, SGPR_G + 1), (SGPR_G + 1,), None, "post"))
            items = main_items
        fin = [it for it in items if it[0] in ("swap", "ds_write")]
        items = [it for it in items if it[0] not in ("swap", "ds_write")]
        if not K["abl_vstore"]:
            readout = [it if it[0] != "vstore" else ("valu", "s_nop 0", (), it[3], "post") for it in readout]
        if not K["abl_stage"]:
            fin, readout = [], []
        return items, fin, readout

    def interleave(a, b):
        if not b:
            return list(a)
        out, j = [], 0
        for i, x in enumerate(a):
            out.append(x)
            while j < len(b) and (j + 1) * len(a) <= (i + 1) * len(b):
                out.append(b[j]); j += 1
        return out + b[j:]

    def flat_of(s):
        """block items of tile s with the PREVIOUS tile pair's row stores dealt evenly between them, then the staging writes (which
        re-use the planes those stores read): see store_flat in gen_bf16_trunk.py"""
        items, fin, readout = epilogue(s)
        prev = epilogue(s - 1)[2] if s > 0 else []
        out = interleave(items, prev) + fin
        if s == N_SLABS - 1:
            out += readout
        return out

    COST = {"ds_read": K["lds_cost"], "ds_write": K["lds_cost"], "valu": K["valu_cost"], "swap": K["valu_cost"],
            "vstore": K["dma_cost"], "salu": K["salu_cost"]}
    def as_filler(item, rel, dl):
        kind, text, writes, tag, _cls = item[:5]
        reads = item[5] if len(item) > 5 else ()
        return T.Filler(text, COST[kind], rel, dl, kind, reads=reads, writes=writes, tag=tag)

    epi_tail, epi_fillers = [], []
    for s in range(N_SLABS):
        L, t = layer_of(s), s % 8
        flat = flat_of(s)
        if s + 1 >= N_SLABS:
            epi_tail = flat
            continue
        rel0 = first[s + 1] + 1
        n_gaps = first[s + 2] - first[s + 1] if s + 2 <= N_SLABS else 32
        hard_dl = first[s + 2] - 1 if s + 2 < N_SLABS else len(mf) - 1
        post_dl = first[s + 3] - 1 if s + 3 < N_SLABS else len(mf) - 1
        per_gap = max(2, -(-len(flat) // max(1, n_gaps - 4)))
        dl = hard_dl
        if t == 7 and s + 1 < N_SLABS:                     # the next layer's first slab reads k-steps 14, 15 of the written set last
            dl = min(dl, idx_of[(s + 1, nk_of(s + 1) - 2, 0)] - 2)
        for j, item in enumerate(flat):
            rel = rel0 + j // per_gap
            d = dl if item[4] == "acc" else max(dl, post_dl)
            f = as_filler(item, min(rel, d), d)
            add(f)
            epi_fillers.append(f)
    for a, b in zip(reversed(epi_fillers[:-1]), reversed(epi_fillers[1:])):      # one program-ordered sequence (gen_bf16_trunk.py)
        if a.deadline > b.deadline:
            a.deadline = b.deadline
            a.release = min(a.release, a.deadline)

    # ---- barrier + weight stream: at slab s (behind the sync point) the pieces of slab s + 3
    for s in range(N_SLABS):
        b = first[s] + K["bar_gap"]
        add(T.Filler("", 0.5, b, b, "bar", tag=s + 1))
        v = s + DMA_DIST
        real = v % N_SLABS
        plist = []
        nbytes = slab_bytes(real)
        for p in range(nbytes // 4096):
            wrap = v >= N_SLABS and real == 0 and p == 0
            plist.append(((real % N_SLOTS) * SLOT_BYTES + p * 4096, wrap, v))
        n_g = first[s + 1] - first[s]
        gaps_avail = max(1, n_g - K["bar_gap"] - 3)
        stride = max(1, gaps_avail // max(1, len(plist)))
        for p, (lds_off, wrap, tag) in enumerate(plist):
            rel = b + 1 + p * stride
            dl = first[s + 1] - 1
            add(T.Filler("s_add_u32 m0, %%[wv1k], %d" % lds_off, K["salu_cost"], min(rel, dl), dl, "m0"))
            if wrap:                                          # back to the start of the blob
                add(T.Filler("v_subrev_u32 %%[goff], %d, %%[goff]" % TOTAL_BYTES, K["valu_cost"], min(rel, dl), dl, "valu", writes=("goff",)))
            add(T.Filler("global_load_lds_dwordx4 %[goff], %[blob]", K["dma_cost"], min(rel, dl), dl, "dma", tag=tag))
            add(T.Filler("v_add_u32 %[goff], 4096, %[goff]", K["valu_cost"], min(rel, dl), dl, "valu", writes=("goff",)))

    # ---- emission (the list scheduler of gen_bf16_trunk.py)
    fillers.sort(key=lambda f: (f.release, f.seq))
    pending = []
    fi = 0
    for dst, src in ((SGPR_G, "gplo"), (SGPR_G + 1, "gphi"), (SGPR_SIGN, "sglo"), (SGPR_SIGN + 1, "sghi")):
        g.emit("s_mov_b32 s%d, %%[%s]" % (dst, src))
        g.last_salu_write[dst] = g.n_states - 1
    g.emit("v_mov_b32 v%d, %%[str0]" % STR1)
    g.emit("v_lshrrev_b32 v%d, 2, %%[va0]" % VSG)
    if K["abl_mask"]:
        g.nop(4)                                                                  # SALU write of the sign pointer -> VMEM address
        for slot in range(8):                                                    # rows p_wave + 8 slot + t, 512 B each
            for t in range(8):
                g.emit("global_load_dword v%d, v%d, s[%d:%d] offset:%d nt" % (SWF + 8 * slot + t, VSG, SGPR_SIGN, SGPR_SIGN + 1, 512 * t))
                g.vm.append(("sw", slot, t))
            if slot < 7:
                g.emit("v_add_u32 v%d, 4096, v%d" % (VSG, VSG))
    for kidx in range(D):
        s, ks = kstep_list[kidx]
        g.emit("ds_read_b128 v[%d:%d], %%[va0] offset:%d" % (ring(kidx), ring(kidx) + 3, frag_addr(s, ks))); g.lgkm.append(("frag", kidx))

    def pop_ready(i):
        nonlocal pending
        pending.sort(key=lambda f: (f.deadline, f.seq))
        budget = K["cap"]
        n = 0
        for f in pending:
            if budget < f.cost - 1e-9:
                break
            g.run_filler(f); budget -= f.cost; n += 1
        pending = pending[n:]

    for i, (s, ks, pt) in enumerate(mf):
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
        g.wait_lgkm({("frag", kidx)})
        a_reg = ring(kidx)
        d = ACC(s & 1, pt)
        b0 = act_reg(read_set(layer_of(s)), ks, pt)
        g.pad_valu_to_mfma([("a", b0 + e) for e in range(4)])
        c_txt = "0" if ks == 0 else "v[%d:%d]" % (d, d + 15)
        g.emit("v_mfma_f32_32x32x16_bf16 v[%d:%d], v[%d:%d], a[%d:%d], %s" % (d, d + 15, a_reg, a_reg + 3, b0, b0 + 3, c_txt), states=8)
        g.mfma_count += 1
        while fi < len(fillers) and fillers[fi].release <= i:
            pending.append(fillers[fi]); fi += 1
        pop_ready(i)

    while fi < len(fillers):
        pending.append(fillers[fi]); fi += 1
    pending.sort(key=lambda f: (f.deadline, f.seq))
    for f in pending:
        g.run_filler(f)
    g.nop(12)
    for item in epi_tail:
        g.run_filler(as_filler(item, 0, 0))
    if g.lgkm:
        g.emit("s_waitcnt lgkmcnt(0)")
        g.lgkm = []
    g.nop(2)
    return g


def main():
    out_path = sys.argv[1]
    knobs = dict(KNOBS)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        knobs[k] = float(v) if isinstance(KNOBS[k], float) else int(v)
    g = gen(knobs)
    T.write_inc(out_path, g, "SN_BF16_CHAIN", "tools/gen_bf16_chain.py " + " ".join(sys.argv[2:]), v_first=64)
    n_other = len(g.out) - g.mfma_count
    print("chain: %d MFMAs, %d other (%.2f / MFMA), nops %d, waits %d, forced %d"
          % (g.mfma_count, n_other, n_other / g.mfma_count, g.stats["nop"], g.stats["wait"], g.stats["forced"]))


if __name__ == "__main__":
    main()
