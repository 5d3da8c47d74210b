#!/usr/bin/env python3
"""Generator of the hand-scheduled trunk of the bf16 fused-MLP forward kernel (sinnerf_amd/csrc/sn_mlp_fwd_bf16_v3.hip).

The trunk = layers xyz_encoding_1..8 + xyz_encoding_final of models/nerf.py:122-140 for one wave's two 32-point tiles:
72 weight slabs (one 32-row output tile x full K each, csrc/sn_layout.h), 2176 v_mfma_f32_32x32x16_bf16.  At the bf16 MFMA
rate (32 cycles per MFMA per SIMD, one wave per SIMD) the kernel's speed is decided by WHERE the ~3 non-MFMA instructions
per MFMA sit: an instruction issued in the 32-cycle shadow of an MFMA is free, a clump of them (a 48-instruction epilogue,
an `s_waitcnt lgkmcnt(0)` behind a just-issued ds_read) is not (round 1: 49 % of peak with 4.9 instructions per MFMA, the
compiler's lgkmcnt(0) before every 4th k-step and the deferred epilogue in one block).  hipcc cannot be told where to put
instructions relative to inline-asm MFMAs, so the whole trunk is emitted as ONE asm statement whose stream is laid out here:

  * backbone: the MFMAs in execution order (slab, k-step, point tile); two accumulator sets alternate between slabs, the
    bias enters as the C operand of a slab's first k-step (no accumulator initialisation moves);
  * fillers, dealt into the gaps after each MFMA by a list scheduler (release gap, deadline, per-gap issue budget):
      - A-fragment ds_read_b128, PREFETCH k-steps ahead through a register ring, waited for with COUNTED lgkmcnt(N);
      - bias ds_reads of the next slab;
      - the deferred epilogue of the previous slab (cvt_pk / pk_max / accvgpr_write into the other activation set),
        two instructions per gap;
      - the weight stream: one 4 KB LDS-DMA piece at a time for the slab SIX ahead (7-slot ring with a static slot per
        slab, see N_SLOTS below), one counted-vmcnt barrier per slab;
  * hazards the compiler would pad (VALU write -> MFMA read, MFMA write -> VALU read, m0 -> LDS-DMA) are tracked on the
    emitted stream and padded with s_nop only where the schedule leaves them too close; tools/check_agpr.py re-checks
    the build.

Register plan inside the statement (physical registers, all declared as clobbers; the compiler keeps v0..v127):
  v[128:191]  accumulators  [set][point tile][16]
  v[192:207]  bias of the slab in flight (C operand of k-step 0)
  v[208:235]  A-fragment ring (prefetch + 2 entries x 4, at most 7)
  v[236:237]  LDS addresses of ring slots 3..5 and 6 (16-bit ds_read offsets reach three 20 KB slots)
  v[240:247]  epilogue temporaries (tmp_pairs of them in rotation)
  v[248:255]  sigma-head weights of the layer-8 epilogue (2 x 4, double-buffered)
  a[0:127], a[128:255]  the two activation sets (as before)

usage: gen_bf16_trunk.py out.inc [knob=value ...]     knobs: see KNOBS below (ablation builds for tools/ timing)
"""
import sys

# ------------------------------------------------------------------------------------------------------------------
KNOBS = dict(prefetch=4,        # A-fragment prefetch distance in k-steps (ring = prefetch + 2 entries)
             cap=4.0,           # issue budget per MFMA gap (VALU / LDS instruction = 1)
             dma_cost=2.0, valu_cost=1.0, lds_cost=1.0, salu_cost=0.5,
             epi=1, dma=1, frag=1, bar=1, sigma=1,    # ablation switches (0 = leave the work out: WRONG results, timing only)
             bar_gap=3,         # the barrier sits after this MFMA of a slab
             setprio=0,
             dma_exec=0,        # experiment: 1 = every DMA runs with EXEC = %[em] (a mask the kernel supplies), WRONG results
             dma_thin=1,        # experiment: issue only every dma_thin-th piece (WRONG results)
             tmp_pairs=2,       # epilogue temporaries: pairs of v[240:247] used in rotation (2 = v[240:243]; the registers a
                                # shorter rotation does not name go back to the compiler, see the clobber list)
             store=0,           # 1 = TRAINING forward (sn_mlp_fwd_bf16_t.hip): every output tile is also written to acts[] as bf16
                                # (whole 128-byte rows, non-temporal) together with the ReLU sign words -- see STORE MODE below
             swap_rev=0,        # store mode: operand order of v_permlane32_swap_b32 (0 = vdst lanes 32..63 <-> src lanes 0..31, confirmed by
                                # the bit-identity test against the compiler-scheduled kernel on the device)
             nt=1,              # store mode: non-temporal hint on the row stores
             dma_early=0,       # 1: a slab's DMA pieces in consecutive gaps right behind the sync point (in FRONT of the slab's row stores)
             spread=1,          # store mode: row stores dealt into the next tile's epilogue (0: burst behind the odd tile)
             abl_sign4=0,       # timing experiment (WRONG results): one dwordx4 sign store per FOUR tiles instead of a dword per tile
             abl_vstore=1, abl_stage=1, abl_sign=1,   # store mode timing ablations (0 = leave out: WRONG results): the global stores, the
                                # staging round trip (swaps + LDS writes / reads + stores), the sign-word arithmetic
             f16=0,             # 1 = the SAME stream with fp16 operands (round 6, SN_DTYPE_F16): v_cvt_pk_f16_f32 / v_mfma_f32_32x32x16_f16
                                # (same issue classes and latencies; v_pk_max_i16 is a ReLU on a packed fp16 pair as on a bf16 pair)
             abl_stw=1, abl_str=1)    # ... the staging ds_write_b128s alone / the staging ds_read_b128s alone (which of the two owns the
                                # LDS bank conflicts the counters see: VERDICT r3 item 8)

STORE_KNOBS = dict(store=1, cap=6.0)            # the build of sn_mlp_fwd_bf16_t.hip (csrc/Makefile passes the same)
V_FIRST = 128                                  # first physical VGPR the statement owns (clobbers v[V_FIRST:255])
ACC = lambda st, pt: 128 + st * 32 + pt * 16   # v[128:191]
BIAS = 192                                     # v[192:207]
RING0 = 208                                    # v[208:235]: up to 7 fragment-ring entries
VA1, VA2 = 236, 237                            # LDS address registers of ring slots 3..5 / 6 (va0 + 61440, + 122880)
TMP0 = 240                                     # v[240:247]
SIGW = 248                                     # v[248:255]: two 4-register buffers
N_SLABS_TRUNK = 72
N_SLABS = 76
# Weight ring: N_SLOTS slots of SLOT_BYTES, slot of slab s = s % N_SLOTS.  76 slabs per point tile + ONE virtual empty slab
# (index 76) make the period 77 = 0 mod 7, so the slot of every slab is static across point tiles.  Slab s is staged
# DMA_DIST slabs ahead: at slab s (behind its barrier) the pieces of virtual index s + DMA_DIST are issued into slot
# (s + DMA_DIST) % 7 = (s - 1) % 7, which every wave has left.  The LDS-DMA latency under load (~1.1 us issued -> landed,
# MI355X_MICROARCH.md) is two slab times: with a 4-slot ring (distance 3) the counted vmcnt wait in front of every barrier
# stalled (measured: 13 % of the kernel); distance 6 keeps ~5 slabs = 5 us in flight.
N_SLOTS = 7
SLOT_BYTES = 20480
DMA_DIST = 6
SLOTS_PER_BASE = 3                             # ds_read offsets are 16 bit: one address VGPR per 3 slots (va0, va1, va2)

# ---- STORE MODE (training forward, bf16 state) -----------------------------------------------------------------
# Ring: 4 slots, distance 3, no virtual slab (76 = 0 mod 4): 80 KB instead of 140, which leaves room for the staging tiles.
# The slab time of the training kernel is bounded by its HBM stores, not by the MFMAs: a shorter DMA lead is affordable.
# Registers: NO extra VGPRs over the inference statement.  The packed output words of the tile being finalised are written over
# the accumulator registers they were converted from -- blocks 0 and 1 of each point tile's 16 accumulators receive
# [t0_0 t1_0 t0_2 t1_2] and [t0_1 t1_1 t0_3 t1_3], laid out so that after four v_permlane32_swap_b32 each lane half holds two
# consecutive 4-register groups = two 16-byte chunks of its point row (every conversion reads its sources before a later one
# overwrites them); they are swapped and written to LDS before slab s+2 re-uses the accumulator set.  v[240:247] two 4-register
# rows in flight (staging read -> global store), v238/v239 conversion temporaries of layer 8 (whose fp32 ReLU outputs also feed
# the sigma head), v232 the ReLU sign word, v233..v235 derived lane addresses; s[84:85] running pointer into acts[layer]
# (+ slot_rows * 512 B per layer), s[86:87] pointer to the sign-word rows.
# Staging tile of a wave (LDS, 9216 B): [point tile: 4608][b3 = tile parity: 2304][e = which of the lane's two chunks: 1152]
# [point row j: 32 B][16 * (h ^ ((j >> 2) & 1))] -- the 16-byte chunk m = 4 b3 + 2 h + e of the 128-byte row of a tile PAIR.
#   write (ds_write_b128, lane = point row): served EIGHT contiguous lanes at a time over 32 banks (128 B) -- rows j .. j+3 take the
#   four 32-byte steps of the window, rows j+4 .. j+7 the same steps with the halves swapped: 8 distinct 16-byte slots;  read
#   (ds_read_b128, lane = (g, k): row 8 i + g, chunk k; 16 lanes over 64 banks): (8 e + 2 g + (h ^ ((g >> 2) & 1))) mod 16 is a
#   bijection on every lane group.  Both conflict-free in the guide's bank model (tools/gcn_sim.py counts them; the first layout
#   swapped the halves on (j >> 3) & 1 -- conflict-free for 16-lane groups, two-way for the 8-lane groups the stores really use:
#   profiles/r04_bf16_t_bank_conflict_by_class.txt, tools/ubench/lds_b128_banks.hip).  One lane-address VGPR for the writes, one for
#   the reads (a second register still carries a copy for the odd row groups), supplied by the kernel.
ST_RO, ST_T0, ST_SB = 240, 238, 232
ST_VS, ST_STR1, ST_VSG = 233, 234, 235      # derived inside the statement: sigma-weight address, odd-row read address, sign-store offset
ST_PT, ST_B3, ST_E = 4608, 2304, 1152
ST_SGPR_ACTS, ST_SGPR_SIGN = 84, 86
ST_N_SLOTS, ST_DMA_DIST = 4, 3

def staged_at(s, store=0):
    """real slabs whose pieces are issued behind the barrier of trunk slab s (virtual indices s + DMA_DIST; slab 0 also does the
    virtual slab's duty: index 76 + DMA_DIST - 77 = 5).  Store mode: slab s + 3, always of the same point tile (the dir section
    of the kernel stages slab 75 and the next tile's slabs 0..2)."""
    if store:
        assert s + ST_DMA_DIST < N_SLABS
        return [s + ST_DMA_DIST]
    out = []
    if s == 0:
        out.append(5)
    v = s + DMA_DIST
    if v < N_SLABS:
        out.append(v)
    elif v > N_SLABS:
        out.append(v - (N_SLABS + 1) + 1000)     # slab of the NEXT point tile (tagged +1000: offsets restart at 0)
    return out

# slab kinds of the trunk (csrc/sn_layout.h): (name, k-steps from xe VGPRs, k-steps from the AGPR set)
def slab_kind(s):
    if s < 8: return ("L0", 4, 0)
    if 32 <= s < 40: return ("SKIP", 4, 16)
    return ("H", 0, 16)

def slab_k(s):           # K of every slab of the network (bytes = K * 64)
    return 64 if s < 8 else 256 if s < 32 else 320 if s < 40 else 256 if s < 72 else 288

def layer_of(s): return s // 8
# activation set read / written by a layer (sn_mlp_fwd_bf16.hip): L0 writes 0; L1 r0 w1; L2 r1 w0; L3 r0 w1; L4 r1(+xe) w0;
# L5 r0 w1; L6 r1 w0; L7 r0 w1; FIN r1 w0
READ_SET = {1: 0, 2: 1, 3: 0, 4: 1, 5: 0, 6: 1, 7: 0, 8: 1}
WRITE_SET = {0: 0, 1: 1, 2: 0, 3: 1, 4: 0, 5: 1, 6: 0, 7: 1, 8: 0}

def act_reg(st, kstep, pt): return st * 128 + (kstep * 2 + pt) * 4


class Filler:
    __slots__ = ("text", "cost", "release", "deadline", "kind", "reads", "writes", "tag", "seq")
    def __init__(self, text, cost, release, deadline, kind, reads=(), writes=(), tag=None):
        self.text, self.cost, self.release, self.deadline, self.kind = text, cost, release, deadline, kind
        self.reads, self.writes, self.tag = set(reads), set(writes), tag
        self.seq = 0


class Gen:
    def __init__(self, knobs):
        self.k = knobs
        self.out = []                 # emitted instruction texts
        self.lgkm = []                # outstanding LDS reads (tags) in issue order
        self.vm = [1, 2, 3, 4]        # outstanding LDS-DMA pieces (issue-order tags) on entry: at most the single pieces of
                                      # slabs 1..4, issued by the previous tile's dir_encoding section / the prologue
        if knobs["store"]:
            self.vm = [1, 2]          # ... store mode (distance 3): slabs 1, 2.  Older memory operations of the kernel that are
                                      # still in flight only make the first counted waits stricter (vmcnt retires in order)
        self.last_salu_write = {}     # SGPR -> wait-state clock of the SALU instruction that wrote it (SALU -> VMEM address: 5 states)
        self.last_valu_write = {}     # reg -> index in self.out of the VALU instruction that wrote it
        self.n_states = 0             # wait states issued so far (every instruction = 1, s_nop n = n + 1)
        self.state_at = []            # wait-state clock of each emitted instruction
        self.last_m0 = -10
        self.mfma_count = 0
        self.stats = dict(nop=0, wait=0, forced=0)

    # ---- raw emission --------------------------------------------------------------------------------------------
    def emit(self, text, writes=(), valu=False, states=1):
        self.out.append(text)
        self.state_at.append(self.n_states)
        if valu:
            for r in writes:
                self.last_valu_write[r] = self.n_states
        self.n_states += states

    def nop(self, n):                 # n wait states
        while n > 0:
            c = min(n, 8)
            self.emit("s_nop %d" % (c - 1), states=c)
            self.stats["nop"] += 1
            n -= c

    def pad_valu_to_mfma(self, regs):
        """VALU write -> MFMA read of the same register needs 2 wait states in between."""
        need = 0
        for r in regs:
            w = self.last_valu_write.get(r)
            if w is not None:
                need = max(need, 3 - (self.n_states - w))      # writer at clock w; reader must be at >= w + 3
        if need > 0:
            self.nop(need)

    def pad_valu_to_swap(self, regs):
        """VALU write -> v_permlane32_swap_b32 read of the same register: 2 wait states (what hipcc inserts for its own code)."""
        self.pad_valu_to_mfma(regs)

    def pad_salu_to_vmem(self, sregs):
        need = 0
        for r in sregs:
            w = self.last_salu_write.get(r)
            if w is not None:
                need = max(need, 6 - (self.n_states - w))
        if need > 0:
            self.nop(need)

    def wait_lgkm(self, tags):
        """counted wait: every LDS read carrying one of `tags` has returned (LDS reads return in order)."""
        pos = -1
        for i, t in enumerate(self.lgkm):
            if t in tags:
                pos = i
        if pos < 0:
            return
        n = len(self.lgkm) - 1 - pos
        assert n <= 15, "lgkmcnt field is 4 bits"
        self.emit("s_waitcnt lgkmcnt(%d)" % n)
        self.stats["wait"] += 1
        del self.lgkm[:pos + 1]

    def wait_vm(self, tag):
        """counted wait on the vector-memory queue (retires in issue order): the operation carrying `tag` has completed.  The count
        field has 6 bits: a target with more than 63 younger operations is covered by vmcnt(63) (at most the 63 youngest remain)."""
        if tag not in self.vm:
            return
        pos = max(i for i, t in enumerate(self.vm) if t == tag)
        self.emit("s_waitcnt vmcnt(%d)" % min(63, len(self.vm) - 1 - pos))
        self.stats["wait"] += 1
        del self.vm[:pos + 1]

    def run_filler(self, f):
        k = f.kind
        if k == "ds_read":
            self.emit(f.text)
            self.lgkm.append(f.tag)
        elif k == "vload":                               # global load into registers of the statement (waited for with wait_vm)
            self.pad_salu_to_vmem(f.reads)
            self.emit(f.text)
            self.vm.append(f.tag)
        elif k == "valu":
            if f.tag is not None:                        # needs LDS data (sigma weights) / a loaded register (("vm", tag))
                if f.tag[0] == "vm":
                    self.wait_vm(f.tag[1])
                else:
                    self.wait_lgkm({f.tag})
            self.emit(f.text, writes=f.writes, valu=True)
        elif k == "ds_write":
            self.emit(f.text)
            self.lgkm.append(("stw",))
        elif k == "swap":
            self.pad_valu_to_swap(f.reads)
            self.emit(f.text, writes=f.writes, valu=True)
        elif k == "salu":
            self.emit(f.text)
            for r in f.writes:
                self.last_salu_write[r] = self.n_states - 1
        elif k == "vstore":
            if f.tag is not None:                        # the staged row has arrived in its registers
                self.wait_lgkm({f.tag})
            self.pad_salu_to_vmem(f.reads)
            self.emit(f.text)
            self.vm.append(10 ** 9)                      # never the target of a counted wait: only ever counts as "younger"
        elif k == "m0":
            self.emit(f.text)
            self.last_m0 = self.n_states - 1
        elif k == "dma":
            if self.n_states - self.last_m0 < 2:         # s_mov m0 -> LDS-DMA: one wait state
                self.nop(1)
            self.dma_seen = getattr(self, "dma_seen", 0) + 1
            if self.dma_seen % self.k["dma_thin"] == 0:
                if self.k["dma_exec"]:
                    self.emit("s_mov_b64 exec, %[em]")
                self.emit(f.text)
                if self.k["dma_exec"]:
                    self.emit("s_mov_b64 exec, -1")
                self.vm.append(f.tag)
        elif k == "bar":
            # own pieces of the NEXT slab have landed (later slabs may still be in flight: counted vmcnt), then all waves meet
            nxt = f.tag
            pos = -1
            for i, t in enumerate(self.vm):
                if isinstance(t, int) and t <= nxt:
                    pos = i
            if pos >= 0:
                self.emit("s_waitcnt vmcnt(%d)" % min(63, len(self.vm) - 1 - pos))   # (6-bit field: a larger count only waits for more)
                del self.vm[:pos + 1]
            self.emit("s_barrier")
        else:
            self.emit(f.text)


def gen(knobs):
    g = Gen(knobs)
    K = knobs
    D = K["prefetch"]
    R = D + 2
    ring = lambda kidx: RING0 + 4 * (kidx % R)           # kidx = global k-step counter over the trunk

    # ---- backbone -------------------------------------------------------------------------------------------------
    # mf[i] = (slab, ks, pt); gk[(slab, ks)] = global k-step index; first[(slab)] = index of its first MFMA
    mf, gk, first, nk = [], {}, {}, {}
    kc = 0
    for s in range(N_SLABS_TRUNK):
        _, nx, na = slab_kind(s)
        nk[s] = nx + na
        first[s] = len(mf)
        for ks in range(nk[s]):
            gk[(s, ks)] = kc
            kc += 1
            for pt in range(2):
                mf.append((s, ks, pt))
    first[N_SLABS_TRUNK] = len(mf)
    total_k = kc
    kstep_list = [(s, ks) for s in range(N_SLABS_TRUNK) for ks in range(nk[s])]
    idx_of = {m: i for i, m in enumerate(mf)}

    fillers = []
    seq = [0]
    def add(f):
        f.seq = seq[0]; seq[0] += 1
        fillers.append(f)

    assert R <= 7, "fragment ring: 7 entries of registers"
    STORE = K["store"]
    n_slots = ST_N_SLOTS if STORE else N_SLOTS
    dma_dist = ST_DMA_DIST if STORE else DMA_DIST
    assert not STORE or R <= 6, "store mode keeps v232..v235 for itself"
    def frag_addr(s, ks):
        slot = s % n_slots
        return ("%[va0]", "v%d" % VA1, "v%d" % VA2)[slot // SLOTS_PER_BASE], (slot % SLOTS_PER_BASE) * SLOT_BYTES + ks * 1024

    # ---- A fragments ------------------------------------------------------------------------------------------------
    # the first D k-steps of the trunk are loaded in the preamble (slab 0 is resident: the previous tile's / the prologue's
    # barrier guaranteed it)
    if K["frag"]:
        for kidx, (s, ks) in enumerate(kstep_list):
            use = idx_of[(s, ks, 0)]
            if kidx < D:
                continue
            # release: after the previous user of this ring entry has issued (k-step kidx - R, its pt-1 MFMA) and, when the
            # fragment belongs to a LATER slab than the one executing at that point, after that slab's barrier
            prev_user = idx_of[(kstep_list[kidx - R][0], kstep_list[kidx - R][1], 1)] if kidx - R >= 0 else -1
            want = idx_of[(kstep_list[kidx - D][0], kstep_list[kidx - D][1], 0)]      # ideal gap: D k-steps ahead
            rel = max(prev_user, want)
            bar_ok = first[s - 1] + K["bar_gap"] if s > 0 else -1                        # barrier(s-1) makes slab s visible
            rel = max(rel, bar_ok + 1) if K["bar"] else rel
            base, off = frag_addr(s, ks)
            add(Filler("ds_read_b128 v[%d:%d], %s offset:%d" % (ring(kidx), ring(kidx) + 3, base, off), K["lds_cost"], rel,
                       use - 1, "ds_read", tag=("frag", kidx)))

    # ---- bias of slab s: 4 x ds_read_b128 into v[192:207]; free once k-step 0 of slab s-1 has issued (+2 MFMAs) ----------
    for s in range(1, N_SLABS_TRUNK):
        rel = first[s - 1] + 3
        dl = first[s] - 1
        # spread over the last gaps of the previous slab
        span = max(1, (first[s] - 1) - rel)
        for q in range(4):
            r = max(rel, first[s] - 10 + 2 * q) if span > 10 else rel
            add(Filler("ds_read_b128 v[%d:%d], %%[vb] offset:%d" % (BIAS + 4 * q, BIAS + 4 * q + 3, s * 128 + q * 16),
                       K["lds_cost"], r, dl, "ds_read", tag=("bias", s)))

    # ---- epilogue of slab s, run inside slab s+1 (the last one is flushed after the backbone) ---------------------------
    # items: (kind, text, writes, tag, cls[, reads]);  cls "acc" = reads the slab's accumulator set (deadline: before slab s+2
    # overwrites it), "post" = works on packed words / LDS / memory only (deadline one slab later; same-deadline items keep their
    # program order, so a slab's post items still run before the next slab's accumulator items)
    def store_epilogue(s):
        L, t = layer_of(s), s % 8
        W, st = WRITE_SET[L], s & 1
        relu, sigma, copy = L <= 6, L == 7, L == 8
        PKR = lambda pt, n: ACC(st, pt) + n              # packed word n of the point tile: over accumulator blocks 0, 1
        # block i of a point tile -> its two packed words; after the swaps PK[0:3] = chunk e=0, PK[4:7] = chunk e=1 of the lane
        slot_of = {0: (0, 1), 2: (2, 3), 1: (4, 5), 3: (6, 7)}
        sigw = lambda i: SIGW + 4 * (i & 1)
        items = []
        def sig_load(i):
            items.append(("ds_read", "ds_read_b128 v[%d:%d], v%d offset:%d" % (sigw(i), sigw(i) + 3, ST_VS, (16 * t + 4 * i) * 4), (), ("sigw", s, i), "acc"))
        step = [0]
        def sign(word):
            if step[0] == 0:
                items.append(("valu", "v_and_b32 v%d, %%[sm], v%d" % (ST_SB, word), (ST_SB,), None, "acc"))
            else:
                items.append(("valu", "v_lshrrev_b32 v%d, 1, v%d" % (ST_SB, ST_SB), (ST_SB,), None, "acc"))
                items.append(("valu", "v_and_or_b32 v%d, v%d, %%[sm], v%d" % (ST_SB, word, ST_SB), (ST_SB,), None, "acc"))
            step[0] += 1
        def block(pt, i):
            a = ACC(st, pt) + 4 * i
            t0, t1 = PKR(pt, slot_of[i][0]), PKR(pt, slot_of[i][1])
            q = 2 * i
            r0 = act_reg(W, 2 * t + (q >> 2), pt) + (q & 3)
            if sigma:
                # layer 8: signs from the packed raw values, sigma head from the fp32 ReLU outputs (nerf.py:136, as in inference),
                # ReLU of the packed pair lands in the point tile's packed-word registers (pk_max(cvt(x), 0) == cvt(max(x, 0)))
                c0, c1 = ST_T0, ST_T0 + 1
                items.append(("valu", "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (c0, a, a + 1), (c0,), None, "acc"))
                items.append(("valu", "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (c1, a + 2, a + 3), (c1,), None, "acc"))
                sign(c0); sign(c1)
                for e in range(4):
                    items.append(("valu", "v_max_f32 v%d, 0, v%d" % (a + e, a + e), (a + e,), None, "acc"))
                for e in range(4):
                    items.append(("valu", "v_fmac_f32 %%[sg%d], v%d, v%d" % (pt, sigw(i) + e, a + e), (), ("sigw", s, i), "acc"))
                items.append(("valu", "v_pk_max_i16 v%d, v%d, 0" % (t0, c0), (t0,), None, "acc"))
                items.append(("valu", "v_pk_max_i16 v%d, v%d, 0" % (t1, c1), (t1,), None, "acc"))
            else:
                items.append(("valu", "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (t0, a, a + 1), (t0,), None, "acc"))
                items.append(("valu", "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (t1, a + 2, a + 3), (t1,), None, "acc"))
                if not copy:
                    sign(t0)
                    items.append(("valu", "v_pk_max_i16 v%d, v%d, 0" % (t0, t0), (t0,), None, "acc"))
                    sign(t1)
                    items.append(("valu", "v_pk_max_i16 v%d, v%d, 0" % (t1, t1), (t1,), None, "acc"))
            items.append(("valu", "v_accvgpr_write_b32 a%d, v%d" % (r0, t0), (("a", r0),), None, "acc"))
            items.append(("valu", "v_accvgpr_write_b32 a%d, v%d" % (r0 + 1, t1), (("a", r0 + 1),), None, "acc"))
        fin = []                                             # packed words -> staging planes (both point tiles), behind everything else
        def finish_pt(pt):
            pairs = [(PKR(pt, 0), PKR(pt, 2)), (PKR(pt, 1), PKR(pt, 3)), (PKR(pt, 4), PKR(pt, 6)), (PKR(pt, 5), PKR(pt, 7))]
            for x, y in pairs:
                if K["swap_rev"]:
                    x, y = y, x
                fin.append(("swap", "v_permlane32_swap_b32 v%d, v%d" % (x, y), (x, y), None, "acc", (x, y)))
            for e in range(2):
                off = pt * ST_PT + (t & 1) * ST_B3 + e * ST_E
                if K["abl_stw"]:
                    fin.append(("ds_write", "ds_write_b128 %%[stw], v[%d:%d] offset:%d" % (PKR(pt, 4 * e), PKR(pt, 4 * e) + 3, off), (), None, "acc"))
        if sigma:                                            # q outermost, both point tiles share a quad's sigma weights
            sig_load(0); sig_load(1)
            for i in range(4):
                for pt in range(2):
                    block(pt, i)
                if i + 2 < 4:
                    sig_load(i + 2)
        else:                                                # point tile outermost: the sign-word step order of the chain
            for pt in range(2):
                for i in range(4):
                    block(pt, i)
        finish_pt(0); finish_pt(1)
        if not copy and K["abl_sign4"]:
            if t % 4 == 3:
                items.append(("vstore", "global_store_dwordx4 v%d, v[%d:%d], s[%d:%d] nt" % (ST_VSG, ST_SB, ST_SB + 3, ST_SGPR_SIGN, ST_SGPR_SIGN + 1), (), None, "post",
                              (ST_SGPR_SIGN, ST_SGPR_SIGN + 1)))
        elif not copy:                                       # the tile's ReLU sign word: 256 contiguous bytes per wave
            items.append(("vstore", "global_store_dword v%d, v%d, s[%d:%d] nt" % (ST_VSG, ST_SB, ST_SGPR_SIGN, ST_SGPR_SIGN + 1), (), None, "post",
                          (ST_SGPR_SIGN, ST_SGPR_SIGN + 1)))
            items.append(("valu", "v_add_u32 v%d, 512, v%d" % (ST_VSG, ST_VSG), (ST_VSG,), None, "post"))
        readout = []
        if t & 1:                                            # tiles t-1, t of both point tiles leave as whole 128-byte rows
            main_items, items = items, readout
            tp = t >> 1
            rows = [(pt, i) for pt in range(2) for i in range(4)]
            def rd(n):
                pt, i = rows[n]
                ro = ST_RO + 4 * (n % 2)
                src = "v%d" % ST_STR1 if (i & 1) else "%[str0]"
                if K["abl_str"]:
                    items.append(("ds_read", "ds_read_b128 v[%d:%d], %s offset:%d" % (ro, ro + 3, src, pt * ST_PT + 256 * i), (), ("ro", s, n), "post"))
            def stw(n):
                ro = ST_RO + 4 * (n % 2)
                rokey = ("ro", s, n) if K["abl_str"] else None
                if K["abl_vstore"]:
                    items.append(("vstore", "global_store_dwordx4 %%[vo], v[%d:%d], s[%d:%d] offset:%d%s" % (ro, ro + 3, ST_SGPR_ACTS, ST_SGPR_ACTS + 1, 128 * tp, " nt" if K["nt"] else ""),
                                  (), rokey, "post", (ST_SGPR_ACTS, ST_SGPR_ACTS + 1)))
                else:                                        # timing ablation: the staged row is still waited for, nothing leaves
                    items.append(("valu", "s_nop 0", (), rokey, "post"))
                if n < 7:
                    items.append(("valu", "v_add_u32 %[vo], 4096, %[vo]", ("vo",), None, "post"))
                else:
                    items.append(("valu", "v_subrev_u32 %[vo], 28672, %[vo]", ("vo",), None, "post"))
            rd(0); rd(1)
            for n in range(8):
                stw(n)
                if n + 2 < 8:
                    rd(n + 2)
            if t == 7:                                       # next layer: acts[L + 1]
                items.append(("salu", "s_add_u32 s%d, s%d, %%[srlo]" % (ST_SGPR_ACTS, ST_SGPR_ACTS), (ST_SGPR_ACTS,), None, "post"))
                items.append(("salu", "s_addc_u32 s%d, s%d, %%[srhi]" % (ST_SGPR_ACTS + 1, ST_SGPR_ACTS + 1), (ST_SGPR_ACTS + 1,), None, "post"))
            items = main_items
        if not K["abl_stage"]:
            fin, readout = [], []
        if not K["abl_sign"]:
            items = [it for it in items if ("v%d" % ST_SB) not in it[1]]
        return items, fin, readout

    def interleave(a, b):
        """deal list b evenly into list a (both keep their own order)"""
        if not b:
            return list(a)
        out, j = [], 0
        for i, x in enumerate(a):
            out.append(x)
            while j < len(b) and (j + 1) * len(a) <= (i + 1) * len(b):
                out.append(b[j]); j += 1
        return out + b[j:]

    def store_flat(s):
        """epilogue stream of slab s in store mode.  The row stores of a finished tile PAIR are not issued as a burst behind the odd
        tile's epilogue: they are dealt evenly into the NEXT (even) tile's epilogue, in front of its staging writes (which re-use
        the planes) -- one store per ~4 MFMAs instead of eight within a few gaps.  A wave whose store cannot issue (the CU's address
        path is full) cannot issue its MFMAs either: without the staging round trip the kernel ran 0.51 ms, with it 0.72."""
        items, fin, readout = store_epilogue(s)
        prev = store_epilogue(s - 1)[2] if (s > 0 and K["spread"]) else []
        out = interleave(items, prev) + fin
        if not K["spread"] or s == N_SLABS_TRUNK - 1:
            out += readout                                   # (the last pair of the trunk: behind the backbone)
        return out

    def plain_epilogue(s):
        L, t = layer_of(s), s % 8
        W = WRITE_SET[L]
        st = s & 1
        sigma = (L == 7) and K["sigma"]
        copy = (L == 8)
        groups = []
        sigw = lambda i: SIGW + 4 * (i & 1)
        def sig_load(i):                                 # sigma-head weights of quad i of this tile (both point tiles share them)
            return [("ds_read", "ds_read_b128 v[%d:%d], %%[vs] offset:%d" % (sigw(i), sigw(i) + 3, (16 * t + 4 * i) * 4), (), ("sigw", s, i), "acc")]
        if sigma:
            groups.append(sig_load(0)); groups.append(sig_load(1))
        for i in range(4):
            for pt in range(2):
                a = ACC(st, pt) + 4 * i
                t0, t1 = TMP0 + 2 * ((2 * i + pt) % K["tmp_pairs"]), TMP0 + 2 * ((2 * i + pt) % K["tmp_pairs"]) + 1
                q = 2 * i
                r0 = act_reg(W, 2 * t + (q >> 2), pt) + (q & 3)
                ins = []
                if sigma:
                    for e in range(4):
                        ins.append(("valu", "v_max_f32 v%d, 0, v%d" % (a + e, a + e), (a + e,), None, "acc"))
                    for e in range(4):
                        ins.append(("valu", "v_fmac_f32 %%[sg%d], v%d, v%d" % (pt, sigw(i) + e, a + e), (), ("sigw", s, i), "acc"))
                    ins.append(("valu", "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (t0, a, a + 1), (t0,), None, "acc"))
                    ins.append(("valu", "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (t1, a + 2, a + 3), (t1,), None, "acc"))
                else:
                    ins.append(("valu", "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (t0, a, a + 1), (t0,), None, "acc"))
                    ins.append(("valu", "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (t1, a + 2, a + 3), (t1,), None, "acc"))
                    if not copy:
                        ins.append(("valu", "v_pk_max_i16 v%d, v%d, 0" % (t0, t0), (t0,), None, "acc"))
                        ins.append(("valu", "v_pk_max_i16 v%d, v%d, 0" % (t1, t1), (t1,), None, "acc"))
                ins.append(("valu", "v_accvgpr_write_b32 a%d, v%d" % (r0, t0), (("a", r0),), None, "acc"))
                ins.append(("valu", "v_accvgpr_write_b32 a%d, v%d" % (r0 + 1, t1), (("a", r0 + 1),), None, "acc"))
                groups.append(ins)
            if sigma and i + 2 < 4:                      # the buffer quad i used is free once its fmacs have issued (in order)
                groups.append(sig_load(i + 2))
        return [x for grp in groups for x in grp]

    COST = {"ds_read": K["lds_cost"], "ds_write": K["lds_cost"], "valu": K["valu_cost"], "swap": K["valu_cost"],
            "vstore": K["dma_cost"], "salu": K["salu_cost"]}
    def as_filler(item, rel, dl):
        kind, text, writes, tag, _cls = item[:5]
        reads = item[5] if len(item) > 5 else ()
        return Filler(text, COST[kind], rel, dl, kind, reads=reads, writes=writes, tag=tag)

    epi_tail = []
    epi_fillers = []
    for s in range(N_SLABS_TRUNK):
        if not K["epi"]:
            break
        L, t = layer_of(s), s % 8
        if s + 1 < N_SLABS_TRUNK:
            rel0 = first[s + 1] + 1                      # two MFMAs after the slab's last one: results readable
            n_gaps = first[s + 2] - first[s + 1] if s + 2 <= N_SLABS_TRUNK else 32
            hard_dl = first[s + 2] - 1 if s + 2 < N_SLABS_TRUNK else len(mf) - 1   # accumulator set is overwritten by slab s+2
            post_dl = first[s + 3] - 1 if s + 3 < N_SLABS_TRUNK else len(mf) - 1
        else:
            rel0 = None
        flat = store_flat(s) if STORE else plain_epilogue(s)
        if rel0 is None:
            epi_tail = flat
            continue
        # deadline of the activation registers: the next layer reads k-steps 2t, 2t+1 of set W.  Only the LAST tile of a
        # layer is read soon (k-steps 14, 15 of the next layer's first slab); everything else only has the accumulator
        # deadline.
        n = len(flat)
        per_gap = max(2, -(-n // max(1, (n_gaps - 4))))
        dl = hard_dl
        if t == 7 and L + 1 <= 8:
            nxt_first_slab = 8 * (L + 1)
            ks_needed = (4 if nxt_first_slab == 32 else 0) + 14      # skip layer: 4 xe k-steps come first
            dl = min(dl, idx_of[(nxt_first_slab, ks_needed, 0)] - 2)
        for j, item in enumerate(flat):
            rel = rel0 + j // per_gap
            d = dl if item[4] == "acc" else max(dl, post_dl)
            f = as_filler(item, min(rel, d), d)
            add(f)
            epi_fillers.append(f)
    if STORE:
        # the epilogue stream is ONE program-ordered sequence (packed-word registers, the sign word, the staging tile and the
        # running store offset are reused from tile to tile): deadlines must not decrease along it, or the list scheduler
        # (earliest deadline first) would let a later tile's conversions overtake an earlier tile's staging writes
        for a, b in zip(reversed(epi_fillers[:-1]), reversed(epi_fillers[1:])):
            if a.deadline > b.deadline:
                a.deadline = b.deadline
                a.release = min(a.release, a.deadline)

    # ---- barrier + weight stream: at slab s, barrier (slab s+1 visible), then DMA of slab s+3 ------------------------------
    for s in range(N_SLABS_TRUNK):
        b = first[s] + K["bar_gap"]
        if K["bar"]:
            add(Filler("", 0.5, b, b, "bar", tag=s + 1))
        if K["dma"]:
            targets = staged_at(s, STORE)
            plist = []                                    # (lds byte offset, goff bump, tag)
            for tgt in targets:
                real = tgt - 1000 if tgt >= 1000 else tgt
                nbytes = slab_k(real) * 64
                pieces = -(-nbytes // 4096)
                slot = real % n_slots
                for p in range(pieces):
                    bump = min(4096, nbytes - p * 4096)
                    if tgt >= 1000 and real == 0 and p == 0:
                        bump = None                       # the stream wraps: goff restarts at tid*16 (+4096 behind this piece)
                    plist.append((slot * SLOT_BYTES + p * 4096, bump, s + dma_dist if (tgt != 5 or STORE) else 5))
            n_g = first[s + 1] - first[s]
            gaps_avail = max(1, n_g - K["bar_gap"] - 3)
            stride = max(1, gaps_avail // max(1, len(plist)))
            if K["dma_early"]:
                stride = 1
            for p, (lds_off, bump, tag) in enumerate(plist):
                rel = b + 1 + p * stride
                dl = first[s + 1] - 1
                add(Filler("s_add_u32 m0, %%[wv1k], %d" % lds_off, K["salu_cost"], min(rel, dl), dl, "m0"))
                if bump is None:                  # back to the start of the blob: minus the whole weight stream (76 slabs)
                    total = sum(slab_k(x) * 64 for x in range(N_SLABS))
                    add(Filler("v_subrev_u32 %%[goff], %d, %%[goff]" % total, K["valu_cost"], min(rel, dl), dl, "valu", writes=("goff",)))
                    bump = 4096
                add(Filler("global_load_lds_dwordx4 %[goff], %[blob]", K["dma_cost"], min(rel, dl), dl, "dma", tag=tag))
                add(Filler("v_add_u32 %%[goff], %d, %%[goff]" % bump, K["valu_cost"], min(rel, dl), dl, "valu", writes=("goff",)))

    # ---- emission -----------------------------------------------------------------------------------------------------
    fillers.sort(key=lambda f: (f.release, f.seq))
    pending = []          # released, not yet emitted (kept in seq order within equal deadlines)
    fi = 0
    if K["setprio"]:
        g.emit("s_setprio %d" % K["setprio"])
    # preamble: address registers, first D fragments + bias of slab 0
    g.emit("v_add_u32 v%d, %d, %%[va0]" % (VA1, SLOTS_PER_BASE * SLOT_BYTES))
    if not STORE:
        g.emit("v_add_u32 v%d, %d, %%[va0]" % (VA2, 2 * SLOTS_PER_BASE * SLOT_BYTES))
    else:                                                # running pointers of the statement (physical SGPRs, declared as clobbers)
        for dst, src in ((ST_SGPR_ACTS, "aplo"), (ST_SGPR_ACTS + 1, "aphi"), (ST_SGPR_SIGN, "sglo"), (ST_SGPR_SIGN + 1, "sghi")):
            g.emit("s_mov_b32 s%d, %%[%s]" % (dst, src))
            g.last_salu_write[dst] = g.n_states - 1
        # lane addresses derived from the kernel's: vb = TAIL + 64 h -> sigma weights at TAIL + 4 (BIAS_FLOATS + 128 h) = 8 vb - 7 TAIL
        # + 4 BIAS_FLOATS (%[vsk] = that constant); the odd-row staging read address = a copy of str0; the sign-store offset = lane * 4
        g.emit("v_lshl_add_u32 v%d, %%[vb], 3, %%[vsk]" % ST_VS)
        g.emit("v_mov_b32 v%d, %%[str0]" % ST_STR1)
        g.emit("v_lshrrev_b32 v%d, 2, %%[va0]" % ST_VSG)
    for q in range(4):
        g.emit("ds_read_b128 v[%d:%d], %%[vb] offset:%d" % (BIAS + 4 * q, BIAS + 4 * q + 3, q * 16)); g.lgkm.append(("bias", 0))
    if K["frag"]:
        for kidx in range(D):
            s, ks = kstep_list[kidx]
            base, off = frag_addr(s, ks)
            g.emit("ds_read_b128 v[%d:%d], %s offset:%d" % (ring(kidx), ring(kidx) + 3, base, off)); g.lgkm.append(("frag", kidx))
    else:
        g.emit("ds_read_b128 v[%d:%d], %%[va0] offset:0" % (RING0, RING0 + 3)); g.lgkm.append(("frag", 0))

    def pop_ready(i):
        """the gap behind MFMA i: pending fillers in (deadline, seq) order while the issue budget lasts.  STRICT order: the
        first filler that does not fit closes the gap (dependent fillers -- m0 / DMA / address bump, cvt / max / write -- sit
        next to each other in this order and must never overtake one another)."""
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
        # release
        while fi < len(fillers) and fillers[fi].release <= i - 1:
            pending.append(fillers[fi]); fi += 1
        # forced fillers (deadline = before this MFMA)
        pending.sort(key=lambda f: (f.deadline, f.seq))
        keep = []
        for f in pending:
            if f.deadline <= i - 1:
                g.run_filler(f); g.stats["forced"] += 1
            else:
                keep.append(f)
        pending = keep
        # operands
        kidx = gk[(s, ks)]
        kind, nx, na = slab_kind(s)
        need = {("frag", kidx if K["frag"] else 0)}
        if ks == 0:
            need.add(("bias", s))
        g.wait_lgkm(need)
        a_reg = ring(kidx) if K["frag"] else RING0
        st = s & 1
        d = ACC(st, pt)
        c = BIAS if ks == 0 else d
        if ks < nx:
            b_txt = "%%[xe%d]" % (ks * 2 + pt)
            regs = ()
        else:
            rs = READ_SET[layer_of(s)]
            b0 = act_reg(rs, ks - nx, pt)
            b_txt = "a[%d:%d]" % (b0, b0 + 3)
            regs = [("a", b0 + e) for e in range(4)]
        g.pad_valu_to_mfma(regs)
        g.emit("v_mfma_f32_32x32x16_bf16 v[%d:%d], v[%d:%d], %s, v[%d:%d]" % (d, d + 15, a_reg, a_reg + 3, b_txt, c, c + 15), states=8)
        g.mfma_count += 1
        # the gap behind this MFMA
        while fi < len(fillers) and fillers[fi].release <= i:
            pending.append(fillers[fi]); fi += 1
        pop_ready(i)

    # ---- tail: everything still pending, the last slab's epilogue (MFMA results: 11 wait states), drain --------------------
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
    g.nop(2)                                             # accvgpr_write -> the compiler's first dir_encoding MFMA
    if K["setprio"]:
        g.emit("s_setprio 0")
    return g


def write_inc(out_path, g, prefix, header, v_first=V_FIRST):
    """the emitted stream as a C string macro <prefix>_ASM + the clobber list <prefix>_CLOBBERS (every physical register the text
    names, and the whole AGPR file)"""
    import re
    n_other = len(g.out) - g.mfma_count
    with open(out_path, "w") as f:
        f.write("// GENERATED by %s -- do not edit.\n" % header)
        f.write("// %d MFMAs, %d other instructions (%.2f per MFMA): %d s_nop, %d counted waits\n"
                % (g.mfma_count, n_other, n_other / g.mfma_count, g.stats["nop"], g.stats["wait"]))
        f.write("#define %s_ASM \\\n" % prefix)
        for line in g.out:
            f.write('  "%s\\n\\t" \\\n' % line)
        f.write('  ""\n')
        used, sused = set(), set()
        for line in g.out:
            for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
                used.update(range(int(m.group(1)), int(m.group(2)) + 1))
            for m in re.finditer(r"\bv(\d+)\b", line):
                used.add(int(m.group(1)))
            for m in re.finditer(r"\bs\[(\d+):(\d+)\]", line):
                sused.update(range(int(m.group(1)), int(m.group(2)) + 1))
            for m in re.finditer(r"\bs(\d+)\b", line):
                sused.add(int(m.group(1)))
        assert used and min(used) >= v_first, "the statement only names registers of its own range"
        f.write("#define %s_CLOBBERS " % prefix + ", ".join('"v%d"' % r for r in sorted(used)) + ", "
                + "".join('"s%d", ' % r for r in sorted(sused))
                + ", ".join('"a%d"' % r for r in range(256)) + ', "memory", "scc"\n')


def main():
    out_path = sys.argv[1]
    knobs = dict(KNOBS)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        knobs[k] = type(KNOBS[k])(float(v)) if isinstance(KNOBS[k], float) else int(v)
    g = gen(knobs)
    if knobs.get("f16"):
        g.out = [l.replace("v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32").replace("v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16") for l in g.out]
    n_other = len(g.out) - g.mfma_count
    with open(out_path, "w") as f:
        f.write("// GENERATED by tools/gen_bf16_trunk.py %s -- do not edit.\n" % " ".join(sys.argv[2:]))
        f.write("// %d MFMAs, %d other instructions (%.2f per MFMA): %d s_nop, %d counted waits\n"
                % (g.mfma_count, n_other, n_other / g.mfma_count, g.stats["nop"], g.stats["wait"]))
        f.write("#define SN_BF16_TRUNK_ASM \\\n")
        for line in g.out:
            f.write('  "%s\\n\\t" \\\n' % line)
        f.write('  ""\n')
        # the statement overwrites the WHOLE hand-managed AGPR file: declared, so that the compiler can never park a value in an
        # AGPR across it (under register pressure it otherwise hoists loop invariants into a0.., which the next tile reads back
        # after this statement has overwritten them -- tools/check_agpr.py flags any compiler-allocated AGPR for the same reason)
        # VGPRs: exactly the physical registers the emitted text names (a shorter fragment ring etc. hands registers back to
        # the compiler, which has to keep everything that lives across the statement in what is left of v0..v255)
        import re
        used = set()
        for line in g.out:
            for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
                used.update(range(int(m.group(1)), int(m.group(2)) + 1))
            for m in re.finditer(r"\bv(\d+)\b", line):
                used.add(int(m.group(1)))
        assert used and min(used) >= V_FIRST, "the statement only names registers of its own range"
        sused = set()
        for line in g.out:
            for m in re.finditer(r"\bs\[(\d+):(\d+)\]", line):
                sused.update(range(int(m.group(1)), int(m.group(2)) + 1))
            for m in re.finditer(r"\bs(\d+)\b", line):
                sused.add(int(m.group(1)))
        f.write("#define SN_BF16_TRUNK_CLOBBERS " + ", ".join('"v%d"' % r for r in sorted(used)) + ", "
                + "".join('"s%d", ' % r for r in sorted(sused))
                + ", ".join('"a%d"' % r for r in range(256)) + ', "memory", "scc"\n')
    print("trunk: %d MFMAs, %d other (%.2f / MFMA), nops %d, waits %d, forced %d"
          % (g.mfma_count, n_other, n_other / g.mfma_count, g.stats["nop"], g.stats["wait"], g.stats["forced"]))


if __name__ == "__main__":
    main()
