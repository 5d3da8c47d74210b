#!/usr/bin/env python3
"""Generator of the hand-scheduled trunk of the bf16x3 TRAINING forward (sinnerf_amd/csrc/sn_mlp_fwd_bf16x3_t.hip).

The trunk = layers xyz_encoding_1..8 + xyz_encoding_final of models/nerf.py:122-140 for one wave's ONE 32-point tile in the 3-term
split arithmetic of csrc/sn_mlp_x3.h (W.x ~= Wh.xh + Wl.xh + Wh.xl on v_mfma_f32_32x32x16_bf16, two accumulator chains in strict
alternation  A B A | B A B, chain A starts from the bias, chain B from the literal 0): 72 weight slabs of K x 128 B (per k-step the hi
fragment then the lo fragment, csrc/sn_layout.h DT_BF16X3), 3 264 MFMAs.  Same arithmetic, same accumulation order, same stored
training state ("x3 state": (hi, lo) pairs in slots 0..8, ReLU sign words in the unused half of slot 9) as the compiler-scheduled
mlp_fwd_bf16x3_kernel<false, 0, true> it replaces -- held to it bit for bit on the device (tests/test_bf16x3_gpu.py) and executed on
the CPU by tools/gcn_sim.py (tests/test_streams_cpu.py).  What changes is who lays out the instruction stream: the compiler-scheduled
kernel issued 6.1 non-MFMA instructions per MFMA with the waves parked 24 % of the time and ran its matrix pipe 45-50 % busy
(profiles/r04_x3_train_kernels.txt; this stream: 4.7 per MFMA, 54-58 % busy, -14 % cycles -- most of which the power-limited chip returns as clock:
profiles/r05_x3_stage_ab.txt); here the whole trunk is ONE asm statement from the list scheduler of tools/gen_bf16_trunk.py
(class Gen): MFMAs back to back, everything else dealt into their 32-cycle shadows with counted lgkmcnt / vmcnt waits.

Per slab (one 32-row output tile x full K) the fillers are
  * A fragments: per k-step two ds_read_b128 (hi, lo) into a ring of four 8-register entries, prefetch distance 3 k-steps;
  * the bias of the next slab (4 x ds_read_b128 into v[192:207], the C operand of chain A's first MFMA);
  * the deferred epilogue of the previous slab's tile, per block of four accumulator registers (31 VALU + 1 LDS; 27 with pk=1):
      v = A + B (v_add_f32) [ReLU: v_max_f32]; hi = cvt_pk(v) -> AGPRs; lo = cvt_pk(v - float(hi)) (v_sub_f32) -> AGPRs;
      [layer 8: sigma head v_fmac_f32 on the fp32 ReLU outputs]; ReLU sign bits (v_pk_min_u16 + v_lshl_or_b32) into the layer's
      sign words; two v_permlane32_swap_b32 give lanes 0..31 the whole 16-byte hi chunk of 8 features and lanes 32..63 the lo chunk;
      ONE ds_write_b128 into the wave's staging tile [32 rows x 128 B, 16-byte chunk c of row j at chunk c ^ (j & 7)] -- writes
      (8 contiguous lanes over 32 banks) and row reads (lane (g, k): row 8 i + g, chunk k ^ g) both conflict-free in the guide's
      bank model (tools/gcn_sim.py counts them);
  * the row stores of that tile: 4 x (ds_read_b128 of a row group + global_store_dwordx4 nt of 8 whole 128-byte point segments);
  * the weight stream, 3 slots of 40 KB (160 KB of LDS hold no fourth), slab s + 2 staged behind slab s's FIRST sync point B1
    ("every wave has left slab s - 1": its slot is free), slab s + 1 awaited at the SECOND one B2 (counted vmcnt + barrier, four
    k-steps before the slab ends: the first fragments of slab s + 1 are prefetched right behind it) -- the two-barrier protocol of
    csrc/sn_mlp_x3.h slab_x3, here with the counted waits DERIVED from the emitted order of vector-memory operations (class Gen
    tracks the issue-ordered queue; the compiler-scheduled kernel's hand-counted VMW rule is what raced in round 4).  The 4-k-step
    slabs of layer 0 have one sync point doing both jobs.
  76 slabs per point tile = 1 mod 3: the slot of a slab rotates from tile to tile, so the three slot addresses are OPERANDS
  (%[vaA..C] lane addresses of the fragment reads, %[smA..C] m0 bases of the DMA) the kernel rotates per tile.

Register plan inside the statement (v[128:255] named as clobbers; the kernel keeps v0..v127):
  v[128:191]  accumulators [set][chain A / B][16]         v[192:207]  bias of the slab in flight
  v[208:239]  A-fragment ring: 4 entries x (hi 4 + lo 4)   v[240:243], v[244:247]  row buffers (staging read -> global store);
  v[248:249]  float(hi) temporaries (an even pair)           the second one holds the sigma-head weights in layer 8
  v250  sign-bit temporary    v[252:255]  the four sign words of the layer (one 16-byte store per lane and layer)
  s[84:85] running pointer into acts[layer] (+ slot_rows * 1024 B per layer), s[86:87] pointer to the layer's sign-word rows

usage: gen_x3_trunk.py out.inc [knob=value ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_bf16_trunk as T                      # noqa: E402  (Gen, Filler, write_inc)

KNOBS = dict(prefetch=3,        # A-fragment prefetch distance in k-steps (ring = 4 entries)
             cap=5.0,           # issue budget per MFMA gap
             dma_cost=2.0, valu_cost=1.0, lds_cost=1.0, salu_cost=0.5,
             b1_gap=3,          # B1 sits behind this MFMA of a slab (3 = behind k-step 0)
             b2_lead=4,         # B2 sits this many k-steps before the end of the slab
             epi_from=1,        # the previous tile's epilogue starts behind this MFMA of the slab
             one_bar=0,         # experiment: ONE sync point per slab doing both jobs (as the 4-k-step slabs always do)
             abl_sign=1, abl_relu=1, abl_vstore=1, abl_stage=1,   # timing ablations (0 = leave the work out: WRONG results): sign bits, ReLU,
                                # the global row stores, the whole staging round trip (swaps + LDS write / read + stores)
             pk=0)              # 1: v_pk_add_f32 for the chain sum and the remainder (two instructions less per pair -- but packed fp32 VALU
                                # beside MFMAs is an anti-lever on this chip: MI355X_MICROARCH.md, and measured here: -8 % cycles, -20 % clock)

N_SLABS = 72                    # trunk slabs; the kernel's dir_encoding section runs slabs 72..75
SLOT_BYTES = 40960
ACC = lambda st, ch: 128 + st * 32 + ch * 16
BIAS = 192
RING0, RING_N = 208, 4
ROW_A, ROW_B = 240, 244
TMP0, VM = 248, 250
SGW = 252
SGPR_ACTS, SGPR_SIGN = 84, 86
READ_SET, WRITE_SET = T.READ_SET, T.WRITE_SET


def nk_of(s): return 4 if s < 8 else 20 if 32 <= s < 40 else 16
def nx_of(s): return 4 if (s < 8 or 32 <= s < 40) else 0          # leading k-steps whose B operands are the embedded xyz (VGPRs)
def slab_bytes(s): return (64 if s < 8 else 320 if 32 <= s < 40 else 288 if s >= 72 else 256) * 128
def layer_of(s): return s // 8
def x3_reg(st, part, ks): return st * 128 + part * 64 + ks * 4
def slot_op(s, kind): return "%%[%s%s]" % (kind, "ABC"[s % 3])    # logical slot of slab s -> the rotated operand


def gen(knobs):
    K = knobs
    g = T.Gen(dict(T.KNOBS, store=1, cap=K["cap"]))
    g.vm = [1, 1]                 # entry: at most the two pieces of slab 1 in flight (older operations only make the first waits stricter)
    D = K["prefetch"]
    assert 1 <= D <= RING_N - 1
    ring = lambda kidx: RING0 + 8 * (kidx % RING_N)

    # ---- backbone: mf[i] = (slab, k-step, term); term 0: c0 += Wh.xh, 1: c1 += Wl.xh, 2: c0 += Wh.xl; c0 = chain A on even k-steps
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

    # sync points: b1[s] / b2[s] = index of the MFMA behind which the barrier sits
    b1, b2 = {}, {}
    for s in range(N_SLABS):
        b1[s] = first[s] + K["b1_gap"] - 1
        b2[s] = first[s + 1] - 1 - 3 * K["b2_lead"] if (nk_of(s) >= 8 and not K["one_bar"]) else b1[s]
        assert b2[s] >= b1[s]

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
            rel = max(rel, b2[s - 1] + 1)                         # slab s is visible behind B2 of slab s - 1
        base = slot_op(s, "va")
        for part in range(2):
            r = ring(kidx) + 4 * part
            add(T.Filler("ds_read_b128 v[%d:%d], %s offset:%d" % (r, r + 3, base, ks * 2048 + part * 1024), K["lds_cost"], rel,
                         use - 1, "ds_read", tag=("frag", kidx, part)))

    # ---- bias of slab s: free once chain A's first MFMA of slab s - 1 has issued
    for s in range(1, N_SLABS):
        rel, dl = first[s - 1] + 3, first[s] - 1
        for q in range(4):
            r = max(rel, first[s] - 12 + 2 * q) if first[s] - rel > 12 else rel
            add(T.Filler("ds_read_b128 v[%d:%d], %%[vb] offset:%d" % (BIAS + 4 * q, BIAS + 4 * q + 3, s * 128 + q * 16), K["lds_cost"],
                         min(r, dl), dl, "ds_read", tag=("bias", s)))

    # ---- epilogue of slab s (tile t of layer L), run inside slab s + 1: items (kind, text, writes, tag, cls[, reads])
    def epilogue(s):
        L, t = layer_of(s), s % 8
        W, st = WRITE_SET[L], s & 1
        relu, sigma, copy = L <= 7, L == 7, L == 8
        items = []
        sgw = SGW + (t >> 1)
        rows_n = 1 if sigma else 2                                  # layer 8's sigma weights live in the second row buffer
        for i in range(4):
            a, b = ACC(st, 0) + 4 * i, ACC(st, 1) + 4 * i
            rh = x3_reg(W, 0, 2 * t + (i >> 1)) + 2 * (i & 1)
            rl = x3_reg(W, 1, 2 * t + (i >> 1)) + 2 * (i & 1)
            V = lambda text, writes=(), tag=None: items.append(("valu", text, writes, tag, "acc"))
            if sigma:
                items.append(("ds_read", "ds_read_b128 v[%d:%d], %%[vs] offset:%d" % (ROW_B, ROW_B + 3, (16 * t + 4 * i) * 4), (), ("sigw", s, i), "acc"))
            if K["pk"]:
                V("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (a, a + 1, a, a + 1, b, b + 1), (a, a + 1))
                V("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (a + 2, a + 3, a + 2, a + 3, b + 2, b + 3), (a + 2, a + 3))
            else:
                for e in range(4):
                    V("v_add_f32 v%d, v%d, v%d" % (a + e, a + e, b + e), (a + e,))
            if relu and K["abl_relu"]:
                for e in range(4):
                    V("v_max_f32 v%d, 0, v%d" % (a + e, a + e), (a + e,))
            V("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (b, a, a + 1), (b,))
            V("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (b + 1, a + 2, a + 3), (b + 1,))
            V("v_accvgpr_write_b32 a%d, v%d" % (rh, b), (("a", rh),))
            V("v_accvgpr_write_b32 a%d, v%d" % (rh + 1, b + 1), (("a", rh + 1),))
            if sigma:                                                # nerf.py:136 on the fp32 ReLU outputs, in the order of the C++ kernel
                for e in range(4):
                    V("v_fmac_f32 %%[sg], v%d, v%d" % (ROW_B + e, a + e), (), ("sigw", s, i))
            for half in range(2):                                    # v - float(hi): the remainder that becomes the lo part
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
            if relu and K["abl_sign"]:                               # sign bits of the post-ReLU pairs: bit k <- low value, 16 + k <- high value
                for half in range(2):
                    k = 2 * i + half + 8 * (t & 1)
                    V("v_pk_min_u16 v%d, v%d, %%[c01]" % (VM, b + half), (VM,))
                    if (t & 1) == 0 and i == 0 and half == 0:
                        V("v_mov_b32 v%d, v%d" % (sgw, VM), (sgw,))
                    else:
                        V("v_lshl_or_b32 v%d, v%d, %d, v%d" % (sgw, VM, k, sgw), (sgw,))
            # lanes 0..31 <- the hi chunk [h0 h1 | partner's h0 h1], lanes 32..63 <- the lo chunk [partner's l0 l1 | l0 l1]
            if K["abl_stage"]:
                items.append(("swap", "v_permlane32_swap_b32 v%d, v%d" % (b, b + 2), (b, b + 2), None, "acc", (b, b + 2)))
                items.append(("swap", "v_permlane32_swap_b32 v%d, v%d" % (b + 1, b + 3), (b + 1, b + 3), None, "acc", (b + 1, b + 3)))
                items.append(("ds_write", "ds_write_b128 %%[stw%d], v[%d:%d]" % (i, b, b + 3), (), None, "acc"))
        # the tile's row stores: row group n = rows 8 n .. 8 n + 7 of the staged tile, lane (g, k) = row g, 16-byte chunk k
        rows = []
        def rd(n):
            ro = ROW_A if (n % rows_n) == 0 else ROW_B
            rows.append(("ds_read", "ds_read_b128 v[%d:%d], %%[str] offset:%d" % (ro, ro + 3, 1024 * n), (), ("ro", s, n), "post"))
        def st_(n):
            ro = ROW_A if (n % rows_n) == 0 else ROW_B
            if not K["abl_vstore"]:
                rows.append(("valu", "s_nop 0", (), ("ro", s, n), "post"))      # timing ablation: the staged row is still waited for
                return
            rows.append(("vstore", "global_store_dwordx4 %%[vo], v[%d:%d], s[%d:%d] offset:%d nt" % (ro, ro + 3, SGPR_ACTS, SGPR_ACTS + 1, 128 * t),
                         (), ("ro", s, n), "post", (SGPR_ACTS, SGPR_ACTS + 1)))
            if n < 3:
                rows.append(("valu", "v_add_u32 %[vo], 8192, %[vo]", ("vo",), None, "post"))
            else:
                rows.append(("valu", "v_subrev_u32 %[vo], 24576, %[vo]", ("vo",), None, "post"))
        if rows_n == 2:
            rd(0); rd(1)
            for n in range(4):
                st_(n)
                if n + 2 < 4:
                    rd(n + 2)
        else:
            for n in range(4):
                rd(n); st_(n)
        if t == 7:
            if relu:                                                 # the layer's four sign words: one 16-byte store per lane
                rows.append(("vstore", "global_store_dwordx4 %%[vsg], v[%d:%d], s[%d:%d] nt" % (SGW, SGW + 3, SGPR_SIGN, SGPR_SIGN + 1), (), None, "post",
                             (SGPR_SIGN, SGPR_SIGN + 1)))
                rows.append(("salu", "s_add_u32 s%d, s%d, 4096" % (SGPR_SIGN, SGPR_SIGN), (SGPR_SIGN,), None, "post"))
                rows.append(("salu", "s_addc_u32 s%d, s%d, 0" % (SGPR_SIGN + 1, SGPR_SIGN + 1), (SGPR_SIGN + 1,), None, "post"))
            rows.append(("salu", "s_add_u32 s%d, s%d, %%[srlo]" % (SGPR_ACTS, SGPR_ACTS), (SGPR_ACTS,), None, "post"))      # next layer: acts[L + 1]
            rows.append(("salu", "s_addc_u32 s%d, s%d, %%[srhi]" % (SGPR_ACTS + 1, SGPR_ACTS + 1), (SGPR_ACTS + 1,), None, "post"))
        if not K["abl_stage"]:
            rows = [r for r in rows if r[0] == "salu"]
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
        rel0 = first[s + 1] + K["epi_from"]                        # two MFMAs behind the slab's last one: results readable
        n_gaps = first[s + 2] - first[s + 1]
        hard_dl = first[s + 2] - 1 if s + 2 < N_SLABS else len(mf) - 1      # slab s + 2 overwrites the accumulator set
        post_dl = first[s + 3] - 1 if s + 3 < N_SLABS else len(mf) - 1
        dl = hard_dl
        if t == 7 and L + 1 <= 8:                                   # the next layer's first slab reads k-steps 14, 15 of the written set
            nxt = 8 * (L + 1)
            dl = min(dl, idx_of[(nxt, nx_of(nxt) + 14, 0)] - 2)
        per_gap = max(2, -(-len(flat) // max(1, n_gaps - 4)))
        for j, item in enumerate(flat):
            d = dl if item[4] == "acc" else max(dl, post_dl)
            f = add(as_filler(item, min(rel0 + j // per_gap, d), d))
            epi_fillers.append(f)
    # ONE program-ordered stream (packed registers, sign words, staging tile, row buffers, running offsets are reused from tile to
    # tile): deadlines must not decrease along it, or the list scheduler (earliest deadline first) lets a later tile's work overtake
    for a, b in zip(reversed(epi_fillers[:-1]), reversed(epi_fillers[1:])):
        if a.deadline > b.deadline:
            a.deadline = b.deadline
            a.release = min(a.release, a.deadline)

    # ---- sync points + weight stream: behind B1 of slab s the pieces of slab s + 2, spread over the gaps up to B2
    for s in range(N_SLABS):
        two = b2[s] != b1[s]
        add(T.Filler("", 0.5, b1[s], b1[s], "bar", tag=(-1 if two else s + 1)))
        if two:
            add(T.Filler("", 0.5, b2[s], b2[s], "bar", tag=s + 1))
        tgt = s + 2
        pieces = slab_bytes(tgt) // 4096
        span = (b2[s] if two else first[s + 1] - 2) - b1[s] - 1
        stride = max(1, span // pieces)
        dl = first[s + 1] - 1
        for p in range(pieces):
            rel = min(b1[s] + 1 + p * stride, dl)
            add(T.Filler("s_add_u32 m0, %s, %d" % (slot_op(tgt, "sm"), p * 4096), K["salu_cost"], rel, dl, "m0"))
            add(T.Filler("global_load_lds_dwordx4 %[goff], %[blob]", K["dma_cost"], rel, dl, "dma", tag=tgt))
            add(T.Filler("v_add_u32 %[goff], 4096, %[goff]", K["valu_cost"], rel, dl, "valu", writes=("goff",)))

    # ---- emission ------------------------------------------------------------------------------------------------------
    fillers.sort(key=lambda f: (f.release, f.seq))
    pending, fi = [], 0
    for dst, src in ((SGPR_ACTS, "aplo"), (SGPR_ACTS + 1, "aphi"), (SGPR_SIGN, "sglo"), (SGPR_SIGN + 1, "sghi")):
        g.emit("s_mov_b32 s%d, %%[%s]" % (dst, src))
        g.last_salu_write[dst] = g.n_states - 1
    for q in range(4):
        g.emit("ds_read_b128 v[%d:%d], %%[vb] offset:%d" % (BIAS + 4 * q, BIAS + 4 * q + 3, q * 16)); g.lgkm.append(("bias", 0))
    for kidx in range(D):
        s, ks = kstep_list[kidx]
        for part in range(2):
            r = ring(kidx) + 4 * part
            g.emit("ds_read_b128 v[%d:%d], %s offset:%d" % (r, r + 3, slot_op(s, "va"), ks * 2048 + part * 1024)); g.lgkm.append(("frag", kidx, part))

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
        # ONE counted wait per k-step: the lo fragment is read right behind the hi one (LDS reads return in order), so waiting for it
        # in front of term 0 covers both -- a second s_waitcnt per k-step is an issue slot for nothing
        need = {("frag", kidx, 0), ("frag", kidx, 1)} if term == 0 else set()
        if ks == 0 and term == 0:
            need.add(("bias", s))
        if need:
            g.wait_lgkm(need)
        st = s & 1
        c0, c1 = (0, 1) if (ks & 1) == 0 else (1, 0)
        d = ACC(st, c1 if term == 1 else c0)
        if ks == 0 and term == 0:
            c_txt = "v[%d:%d]" % (BIAS, BIAS + 15)                   # chain A starts from the bias
        elif ks == 0 and term == 1:
            c_txt = "0"                                              # chain B starts from zero
        else:
            c_txt = "v[%d:%d]" % (d, d + 15)
        bpart = 1 if term == 2 else 0                                # B operand: x hi for terms 0, 1; x lo for term 2
        nx = nx_of(s)
        if ks < nx:
            b_txt, regs = "%%[x%s%d]" % ("hl"[bpart], ks), ()
        else:
            b0 = x3_reg(READ_SET[layer_of(s)], bpart, ks - nx)
            b_txt, regs = "a[%d:%d]" % (b0, b0 + 3), [("a", b0 + e) for e in range(4)]
        g.pad_valu_to_mfma(regs)
        a_reg = ring(kidx) + 4 * part
        g.emit("v_mfma_f32_32x32x16_bf16 v[%d:%d], v[%d:%d], %s, %s" % (d, d + 15, a_reg, a_reg + 3, b_txt, c_txt), states=8)
        g.mfma_count += 1
        while fi < len(fillers) and fillers[fi].release <= i:
            pending.append(fillers[fi]); fi += 1
        pop_ready()

    # ---- tail: everything still pending, the last tile's epilogue and row stores (MFMA results: 20 wait states), drain ------------
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
    g.nop(2)                                                         # accvgpr_write -> the compiler's first dir_encoding MFMA
    return g


def main():
    out_path = sys.argv[1]
    knobs = dict(KNOBS)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        knobs[k] = type(KNOBS[k])(float(v)) if isinstance(KNOBS[k], float) else int(v)
    g = gen(knobs)
    T.write_inc(out_path, g, "SN_X3_TRUNK", "tools/gen_x3_trunk.py " + " ".join(sys.argv[2:]))
    n_other = len(g.out) - g.mfma_count
    print("x3 trunk: %d MFMAs, %d other (%.2f / MFMA), nops %d, waits %d, forced %d"
          % (g.mfma_count, n_other, n_other / g.mfma_count, g.stats["nop"], g.stats["wait"], g.stats["forced"]))


if __name__ == "__main__":
    main()
