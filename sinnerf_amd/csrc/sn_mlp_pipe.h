// sn_mlp_pipe.h -- weight-slab pipeline of the fused MLP kernels (fp32 path), second generation.
//
// A workgroup (4 waves) consumes the packed weights slab by slab (one 32-row output tile x full K each, sn_layout.h).
// Slabs stream L2 -> LDS by global_load_lds DMA into a RING OF THREE buffers and the single per-slab barrier sits in the
// MIDDLE of a slab's MFMA sequence instead of at the slab boundary:
//
//   sync point of slab s (after its first GB MFMA groups):
//       s_waitcnt vmcnt(0)        own DMA pieces of slab s+1 (issued one slab ago) have landed
//       s_barrier                 -> slab s+1 is complete for every wave; every wave has left slab s-1
//       issue DMA of slab s+2     into the ring slot slab s-1 used; one 4 KB piece per MFMA group, in the MFMA shadow
//
// The kernels are PERSISTENT (one workgroup per CU walks its point tiles): the weight stream simply wraps around from the
// last slab of a tile to slab 0 of the next, so the ring never drains and there is no per-tile launch / prologue bubble.
// The transition slab s -> s+1 needs no barrier: the bias of slab s+1 and its first A fragment are requested while
// the last MFMAs of slab s are still issuing, the accumulator epilogue of slab s (ReLU, moves, activation stores) runs
// under the first MFMAs of slab s+1, and the barrier itself waits under an MFMA that is already in flight.
// (First generation: double buffer, barrier + DMA issue + bias/fragment reload on the critical path at every slab
//  boundary -- 85 % MFMA-busy; see profiles/r01_run2_pmc.json.)
#pragma once
#include "sn_mlp_common.h"

namespace snk {

constexpr int TAIL_LDS_BYTES = 12544;                       // biases + aux head table (12320 B), padded
constexpr int RING_SLOT_BYTES = snl::MAX_SLAB_K * 128;      // 40960
constexpr int MLP_F32_LDS_BYTES_V2 = TAIL_LDS_BYTES + 3 * RING_SLOT_BYTES;   // 135424
// Training forward: per-wave staging tile that turns the accumulator layout (lane = point, 4 consecutive features per
// register quad -> a 16-byte access per lane with a 1 KB lane stride, 64 cache lines per store instruction) into row
// accesses (8 lanes x 16 B = the 128 contiguous bytes one point owns in a 32-feature tile, 8 rows per instruction).
constexpr int XPOSE_PITCH = 36;                              // floats per point row (32 + 4: conflict-free b128 both ways)
constexpr int XPOSE_WAVE_BYTES = 32 * XPOSE_PITCH * 4;       // 4608
constexpr int XPOSE_LDS_BYTES = 4 * XPOSE_WAVE_BYTES;        // 18432
// Inference: per-wave staging of the epilogue's LDS round trip (epi32_relu_lds): quad q of lane l at q * 1024 + l * 16
constexpr int EPI_WAVE_BYTES = 4096;
constexpr int EPI_LDS_BYTES = 4 * EPI_WAVE_BYTES;            // 16384

template <int BYTES_PER_K, int SLOT_BYTES>
struct RingT {
  const char* blob;     // packed weights (slab 0)
  const char* gnext;    // global address of the next slab to stage
  char* base;           // LDS address of ring slot 0
  int n_used;           // slabs per point tile consumed by this kernel (76, or 64 for sigma_only)
  int stage_id;         // id (0..n_used-1) of the next slab to stage
  int stage_slot;       // ring slot (0..2) it goes to
  long remaining;       // slabs still to stage over the whole life of this (persistent) workgroup
  int tid;
  int wbase;            // wave-uniform LDS byte offset of this wave inside a 4 KB piece
  int pieces, piece;    // staging state of the slab being staged (pieces of 4096 B, the last one may be partial)
  int slab_bytes;
  const char* gp;       // per-lane global source of the next piece
  const char* gps;      // wave-uniform global source of the next piece (piece_static_s)
  char* lp;             // wave-uniform LDS destination of the next piece

  SN_DEV char* slot(int k) const { return base + k * SLOT_BYTES; }
  SN_DEV void begin_stage() {
    slab_bytes = slab_k_rt(stage_id) * BYTES_PER_K;
    pieces = (remaining > 0) ? ((slab_bytes + 4095) >> 12) : 0;
    piece = 0;
    gp = gnext + tid * 16;
    lp = slot(stage_slot) + wbase;
  }
  SN_DEV void issue_piece() {           // one 4096-byte piece (16 B per thread)
    if (piece < pieces) {
      if ((BYTES_PER_K * 32) % 4096 == 0 || piece * 4096 + wbase < slab_bytes)     // wave-uniform (1 KB per wave)
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)gp, (lds_void*)lp, 16, 0, 0);
      gp += 4096; lp += 4096;
      ++piece;
    }
  }
  // compile-time piece count: no per-piece compare/branch.  begin_static()/piece_static()/end_static<NP>() stage slab
  // `stage_id` unconditionally (a workgroup's last tiles over-stage the first slabs of a tile nobody consumes: harmless,
  // drained before exit).
  SN_DEV void begin_static() {
    gp = gnext + tid * 16;
    lp = slot(stage_slot) + wbase;
  }
  SN_DEV void piece_static() {
    // Inline asm, not __builtin_amdgcn_global_load_lds: hipcc books the builtin as a FLAT operation that may touch LDS *and* memory,
    // and while one is pending every wait it inserts is s_waitcnt vmcnt(0) / lgkmcnt(0) -- the A-fragment prefetch distance of the
    // slab loops collapsed to "whatever was issued last" (all 272 LDS waits of the bf16x3 inference kernel were lgkmcnt(0)).  The
    // kernels wait for their DMA pieces themselves (counted vmcnt at the sync points), so the compiler need not see them.
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(gp), "s"((unsigned)(size_t)lp) : "memory");
    gp += 4096; lp += 4096;
  }
  // ... in two halves, for callers that have an instruction of their own to put between the m0 write and the load (the wait state
  // the pair needs; piece_static() spends an s_nop on it).  Nothing that writes m0 may sit between the halves: the callers put LDS
  // reads there, and tools/check_agpr.py verifies on the generated code that every LDS-DMA load has its m0 write 2+ instructions up.
  SN_DEV void piece_m0() { asm volatile("s_mov_b32 m0, %0" :: "s"((unsigned)(size_t)lp) : "memory"); }
  SN_DEV void piece_load() {
    asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(gp) : "memory");
    gp += 4096; lp += 4096;
  }
  // ... with the wave-uniform source in an SGPR pair and ONE constant per-lane VGPR offset (tid * 16): no VALU per piece.
  // v_mfma_f32_32x32x2_f32 runs on the f32 VECTOR pipe: any VALU instruction between two of them costs 9.6 + 4 n cycles per gap
  // (tools/ubench/f32_gap_cost.hip, profiles/r06_ubench_f32_gap_cost.txt: SALU, waits, LDS, global and LDS-DMA instructions cost
  // nothing there), and the per-lane 64-bit `gp += 4096` of piece_static() is a v_lshl_add_u64 per piece -- 568 such gaps per point
  // tile of the fp32 forward.  Used by slab_f32a (fp32 forward and backward chain).
  SN_DEV void begin_static_s() {
    gps = gnext;
    lp = slot(stage_slot) + wbase;
  }
  SN_DEV void piece_static_s() {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %2"     /* SALU write of the address pair -> VMEM: 5 wait states (check_agpr.py rule 5) */
                 :: "v"((unsigned)(tid * 16)), "s"((unsigned)(size_t)lp), "s"(gps) : "memory");
    gps += 4096; lp += 4096;
  }
  SN_DEV void skip_static() { gp += 4096; lp += 4096; }      // a trailing partial piece this wave has no share of
  template <int NBYTES>
  SN_DEV void end_static_bytes() {
    gnext += NBYTES;
    stage_slot = (stage_slot == 2) ? 0 : stage_slot + 1;
    if (++stage_id == n_used) { stage_id = 0; gnext = blob; }
  }
  template <int NP>
  SN_DEV void end_static() {
    gnext += NP * 4096;
    stage_slot = (stage_slot == 2) ? 0 : stage_slot + 1;
    if (++stage_id == n_used) { stage_id = 0; gnext = blob; }
  }
  SN_DEV void end_stage() {
    if (pieces > 0) {
      gnext += slab_bytes;
      --remaining;
      stage_slot = (stage_slot == 2) ? 0 : stage_slot + 1;
      if (++stage_id == n_used) { stage_id = 0; gnext = blob; }    // next point tile: the weight stream wraps around
      pieces = 0;
    }
  }
  SN_DEV void stage_whole() {           // prologue only
    begin_stage();
#pragma unroll
    for (int i = 0; i < 10; ++i) issue_piece();
    end_stage();
  }
};
typedef RingT<128, RING_SLOT_BYTES> Ring;          // fp32 weights: 128 B per K per 32-row tile

// ---- fp32 slab, inline-asm MFMAs over a hand-managed AGPR file ------------------------------------------------------
// hipcc keeps every builtin-MFMA accumulator in AGPRs: each VALU use of a result costs a v_accvgpr_read, each accumulator
// hand-over a v_accvgpr_mov, and the 256 activation registers get shuffled between the two register files (v_accvgpr_write
// + s_nop pairs) -- a third of the non-MFMA instructions of the builtin version of this kernel.  So, as in the bf16 kernel:
//   accumulators (C/D)   architectural VGPRs, two sets used alternately by consecutive slabs (nothing is copied); the bias of
//                        the next slab is ds_read straight into the set the previous slab vacated
//   activations (B)      a[0:127] and a[128:255]: a layer reads its B operands from one set (MFMA reads B from AGPRs for
//                        free) while its epilogues v_accvgpr_write the next layer's into the other; roles swap every layer.
//                        AGPR numbers are immediates in the asm text, the compiler never sees these registers (it is told
//                        a255 is clobbered so the kernel is sized for all 256; tools/check_agpr.py verifies the build)
//   xyz / dir embeddings VGPRs (B operands of layer 0, the skip layer, dir_encoding)
// One slab = NG0 + NG1 groups of 4 k-steps (two K segments), barrier after group GB.
//   SET0/SET1  B operands of the segment: AGPR activation set 0/1 (register SET*128 + K-slot), or -1 = the VGPR array bv
//   acc        accumulator set of this slab, bias-initialised on entry
//   accn       the other set: the previous slab's result until pending(0..3) have consumed it (groups 0..3), then the
//              bias of slab s_next (requested in group 4)
//   af         A fragments of groups g, g+1 for even g: both are requested together two groups ahead (one s_waitcnt per
//              TWO groups).  Invariant at entry / exit: af = fragments of groups 0, 1 of the slab.
//   NP         4 KB pieces of the slab staged at this slab's sync point (the slab two ahead): compile-time, so a DMA piece
//              is m0 + address bump + global_load_lds with no compare/branch
// (hipcc pads an `s_nop 0` between two dependent asm MFMAs -- ~90 per slab; bundling four MFMAs into one asm where a group's
// gaps carry no work removes them and changes nothing measurable, so the simple form is kept.)
// Issue discipline (measured: T = 64.2 N_mfma + 4.6 N_other, a single dependent accumulator chain hides nothing): the
// non-MFMA work of a group is dealt over its four gaps --
//     MFMA0 | A-fragment prefetch | MFMA1 | one DMA piece | MFMA2 | epilogue slice / activation-store step | MFMA3
// pending(i), i = 0..3: slice i (4 accumulator registers) of the previous slab's epilogue, run in groups 0..3;
// late(i), i = 0..7 (groups S0 .. S0+7): the memory steps of the training kernels -- the row-group stores of the previous
// tile (steps 0..3) and, in the backward chain, the requests of the next activation tile (steps 4..7).
// (Reading a row group out of the staging tile one group AHEAD of its store, so that the store does not wait for the LDS
// latency in its own gap, was measured: 3.638 vs 3.641 ms on the training forward -- nothing, dropped.)
// ONE vector-memory instruction per group and per wave: the DMA pieces of the slab go to the groups [GB, S0) before them
// (two or three per group saturate the CU's address path and stall MFMA issue), and the chain's scattered activation
// loads (64 cache lines per instruction) come LAST: whatever is issued behind them queues up in the address path
// (measured on the chain, 4.4 ms of MFMAs: stores+loads beside the pieces 4.82-5.15 ms, pieces | loads | stores 4.98,
// pieces | stores | loads 4.77).  Waiting with a counted vmcnt for everything but the youngest stores (fence-less barrier,
// loads in inline asm) was measured too: no gain, dropped.
// The compiler does not know the asm is an MFMA: the MFMA -> VALU-read hazard (18 wait states for this 16-pass MFMA) is kept
// by construction -- pending(0) sits behind three MFMAs of the NEXT slab, at a layer end an explicit s_nop run is used.
// FIRST = first MFMA of a slab: its C operand may have just been written by compiler-inserted VALU copies (accumulator
// hand-over at control-flow joins), and a VALU write -> MFMA read needs 2 wait states the compiler cannot insert for asm.
template <bool FIRST>
SN_DEV void mma32_a(f32x16& acc, float a, int reg) {       // D = A.B + D; D, A in VGPRs, B = a[reg]
  if (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, a[%2], %0" : "+v"(acc) : "v"(a), "n"(reg));
  else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, a[%2], %0" : "+v"(acc) : "v"(a), "n"(reg));
}
template <bool FIRST>
SN_DEV void mma32_v(f32x16& acc, float a, float b) {       // ... B in a VGPR
  if (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// MFMA (16 passes) -> VALU read of its result at a layer end.  The accumulator is an operand: plain C++ arithmetic on it
// (the softplus epilogue) could otherwise be scheduled above the wait.
SN_DEV void mfma32_result_fence(f32x16& acc) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc)); }

template <int NG0, int NG1, int SET0, int SET1, int GB, int NP, int S0 = -1, class RingX, class Pending, class Late>
SN_DEV void slab_f32a(f32x16& acc, f32x16& accn, f32x4 (&af)[2], const char* lw, const float* bv, const char* lw_next,
                      const float* lds_bias, int s_next, int h, RingX& ring, Pending&& pending, Late&& late) {
  constexpr int NG = NG0 + NG1;
  constexpr int GL = S0 < 0 ? NG : S0;                       // first group of the late steps; DMA pieces go to [GB, GL)
  constexpr int PPG = (NP + (GL - GB) - 1) / (GL - GB);      // DMA pieces per group of that window
  static_assert(GB >= 2 && GB % 2 == 0 && GB <= 4 && NG >= 8 && NG % 2 == 0, "sync point inside the slab; fragments come in pairs");
  static_assert(GL >= 4 && GL > GB && GL <= NG, "late steps follow the pending steps and the DMA window");
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g == GB) {                               // sync point (one longer gap per slab)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      ring.begin_static_s();
    }
    if (g == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the LDS round trips of the epilogue slices (epi32_*_lds, groups 0..3)
                                                                      // have landed in their AGPRs: consumers are >= 1 slab (or 22 groups) away
    if (g == 4) accn = load_bias(lds_bias, s_next, h);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 a_cur = af[g & 1];
    auto mma = [&](int kk) __attribute__((always_inline)) {
      if (g == 0 && kk == 0) {
        if (SET0 < 0) mma32_v<true>(acc, a_cur[0], bv[0]); else mma32_a<true>(acc, a_cur[0], SET0 * 128);
      } else if (g < NG0) {
        if (SET0 < 0) mma32_v<false>(acc, a_cur[kk], bv[4 * g + kk]); else mma32_a<false>(acc, a_cur[kk], SET0 * 128 + 4 * g + kk);
      } else {
        if (SET1 < 0) mma32_v<false>(acc, a_cur[kk], bv[4 * (g - NG0) + kk]);
        else mma32_a<false>(acc, a_cur[kk], SET1 * 128 + 4 * (g - NG0) + kk);
      }
    };
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 n0, n1;
    if (g & 1) {                                 // gap 1 of odd groups: both fragments of the next pair of groups
      const int gn = g + 1;
      if (gn < NG) {
        n0 = *reinterpret_cast<const f32x4*>(lw + gn * 1024);
        n1 = *reinterpret_cast<const f32x4*>(lw + (gn + 1) * 1024);
      } else {
        n0 = *reinterpret_cast<const f32x4*>(lw_next);
        n1 = *reinterpret_cast<const f32x4*>(lw_next + 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
    if (g >= GB && g < GL) {                     // gap 2: weight DMA
#pragma unroll
      for (int i = 0; i < PPG; ++i) if ((g - GB) * PPG + i < NP) ring.piece_static_s();
      __builtin_amdgcn_sched_barrier(0);
    }
    mma(2);
    __builtin_amdgcn_sched_barrier(0);
    if (g < 4) pending(g);                       // gap 3: epilogue slice of the previous slab ...
    else if (g >= GL && g < GL + 8) late(g - GL); // ... later the memory steps of the training kernels: late(0..7)
    __builtin_amdgcn_sched_barrier(0);
    mma(3);
    if (g & 1) { af[0] = n0; af[1] = n1; }
  }
  ring.template end_static<NP>();
}

// Epilogue blocks: four accumulator values -> four consecutive registers of the hand-managed AGPR file a[reg..reg+3]
// (`reg` must fold to a constant, it is printed into the asm text).  One volatile asm per block: it keeps its program order
// relative to the MFMA asm, and the compiler cannot pad VALU <-> asm dependences with s_nops inside it.
SN_DEV void epi32_relu(int reg, float x0, float x1, float x2, float x3, float (&v)[4]) {     // v = relu(x) (nerf.py:73)
  asm volatile("v_max_f32 %0, 0, %4\n\tv_max_f32 %1, 0, %5\n\tv_max_f32 %2, 0, %6\n\tv_max_f32 %3, 0, %7\n\t"
               "v_accvgpr_write_b32 a[%8], %0\n\tv_accvgpr_write_b32 a[%9], %1\n\t"
               "v_accvgpr_write_b32 a[%10], %2\n\tv_accvgpr_write_b32 a[%11], %3"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "n"(reg), "n"(reg + 1), "n"(reg + 2), "n"(reg + 3));
}
SN_DEV void epi32_copy(int reg, float x0, float x1, float x2, float x3) {                     // no activation
  asm volatile("v_accvgpr_write_b32 a[%4], %0\n\tv_accvgpr_write_b32 a[%5], %1\n\t"
               "v_accvgpr_write_b32 a[%6], %2\n\tv_accvgpr_write_b32 a[%7], %3"
               :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "n"(reg), "n"(reg + 1), "n"(reg + 2), "n"(reg + 3));
}

// Epilogue blocks WITHOUT VALU work (round 6).  The f32-input MFMA shares its pipe with the VALU (above): the eight VALU of an
// epi32_relu block cost 9.6 + 32 cycles of matrix time, four blocks per slab.  LDS instructions cost nothing between MFMAs, and on
// gfx90a+ a DS load can target AGPRs directly: the four accumulator values go out with ONE ds_write_b128 into a wave-private staging
// tile, ReLU is four ds_max_i32 against 0 on the written words (as signed integers every negative float -- and -0.0 -- is < 0, every
// positive float is itself: max_i32(bits, 0) == bits of max(x, +0.0) for every non-NaN x; the LDS executes one wave's instructions
// in order), and ONE ds_read_b128 puts the result into a[reg..reg+3].  `zero` = a VGPR holding 0.  The consumer waits with
// s_waitcnt lgkmcnt(0) (slab_f32a, group 6).  addr = this lane's byte address of the staging quad; OFF = immediate offset.
SN_DEV void epi32_relu_lds(int reg, unsigned addr, int off, f32x4 x, unsigned zero) {
  asm volatile("ds_write_b128 %0, %1 offset:%3\n\t"
               "ds_max_i32 %0, %2 offset:%3\n\tds_max_i32 %0, %2 offset:%4\n\tds_max_i32 %0, %2 offset:%5\n\tds_max_i32 %0, %2 offset:%6\n\t"
               "ds_read_b128 a[%7:%8], %0 offset:%3"
               :: "v"(addr), "v"(x), "v"(zero), "n"(off), "n"(off + 4), "n"(off + 8), "n"(off + 12), "n"(reg), "n"(reg + 3) : "memory");
}
// ... the same round trip as SINGLE instructions, for kernels that deal them one per MFMA gap (sn_mlp_fwd_f32g.hip: issued as one
// burst per slice, the six LDS instructions of four lock-stepped waves fill the LDS data FIFO and the wave's next MFMA waits for the
// last of them to ISSUE -- 2 % of the fine pass; one LDS instruction per gap costs nothing, tools/ubench/f32_gap_cost.hip)
SN_DEV void lds_put_quad(unsigned addr, int off, f32x4 x) { asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(x), "n"(off) : "memory"); }
SN_DEV void lds_relu_word(unsigned addr, int off, unsigned zero) { asm volatile("ds_max_i32 %0, %1 offset:%2" :: "v"(addr), "v"(zero), "n"(off) : "memory"); }
SN_DEV void lds_get_quad_agpr(int reg, unsigned addr, int off) {
  asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" :: "v"(addr), "n"(reg), "n"(reg + 3), "n"(off) : "memory");
}
SN_DEV void epi32_copy_lds(int reg, unsigned addr, int off, f32x4 x) {       // no activation; also: values already activated on the VALU
  asm volatile("ds_write_b128 %0, %1 offset:%2\n\tds_read_b128 a[%3:%4], %0 offset:%2"
               :: "v"(addr), "v"(x), "n"(off), "n"(reg), "n"(reg + 3) : "memory");
}

// ReLU as ONE v_max_f32 (fmaxf() on an MFMA result makes hipcc emit a canonicalising v_max first: 2 VALU per value, and
// the epilogues of the bf16 path are VALU-issue-bound).
SN_DEV float relu1(float x) {
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}

// ShiftedSoftplus (models/activations.py:33-35) on hardware exp2/log2:  max(x-1,0) + log1p(exp(-|x-1|)).
// log1p(e) for e in (0,1]: u = fl(1+e), d = u-1 (exact).  log(u) has lost the bits 1+e rounded away; they come back as the first-order
// term (e - d)/u ~ e - d: |e - d| <= 2^-24, so replacing the factor 1/u in [1/2, 1] by 1 is an absolute error <= 3e-8 on a result >= e/2
// -- and it is exactly right where it matters (e << 1: u -> 1; e < 2^-24: d = 0, log(u) = 0, result = e).  Round 6: this form has no
// division, compare or select -- beside the f32-input MFMA every VALU instruction is matrix time (a reciprocal: 16 cycles) -- and the same
// accuracy as the `log(u) * e/d` form of rounds 1-5 (max rel. error 1.8e-6 over [-30, 30], the exp2 argument's rounding; numpy emulation).
SN_DEV float shifted_softplus_fast(float x) {
  const float sx = x - 1.0f;
  const float e = __builtin_amdgcn_exp2f(-fabsf(sx) * 1.44269504088896340736f);
  const float u = 1.0f + e;
  const float d = u - 1.0f;
  const float l = __builtin_amdgcn_logf(u) * 0.69314718055994530942f;
  return fmaxf(sx, 0.0f) + (l + (e - d));
}

}  // namespace snk
