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
// (First generation = Stager in sn_mlp_common.h: double buffer, barrier + DMA issue + bias/fragment reload on the
//  critical path at every slab boundary -- 85 % MFMA-busy; see profiles/r01_run2_pmc.json.)
#pragma once
#include "sn_mlp_common.h"

namespace snk {

constexpr int TAIL_LDS_BYTES = 12544;                       // biases + aux head table (12320 B), padded
constexpr int RING_SLOT_BYTES = snl::MAX_SLAB_K * 128;      // 40960
constexpr int MLP_F32_LDS_BYTES_V2 = TAIL_LDS_BYTES + 3 * RING_SLOT_BYTES;   // 135424

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
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)gp, (lds_void*)lp, 16, 0, 0);
    gp += 4096; lp += 4096;
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

// One slab: NG0 + NG1 groups of 4 k-steps (two K segments with B operands b0 / b1), barrier after group GB.
//   acc      in: bias-initialised accumulator of this slab; out: its result (bias + W.x)
//   a_cur    in: first A fragment of this slab (prefetched by the previous slab); out: first fragment of the next slab
//   acc_pre  out: bias of slab s_next (requested right after the sync point, consumed by the next slab)
//   pending  run after the first group's MFMAs are issued (the previous slab's epilogue)
// One dependent accumulator chain: measured on MI355X (tools/ubench/mfma_chain.hip) a dependent v_mfma_f32_32x32x2 chain
// with a ds_read_b128 + s_waitcnt every 4 MFMAs and a barrier every 32 sustains 152 TF from one wave per SIMD, the same as
// two interleaved chains -- so no second accumulator is spent.
// Issue discipline (measured with an MFMA-duplication experiment: every extra v_mfma costs exactly 64.2 cycles, and a
// FIXED ~1170 cycles per slab were lost on top): the wave issues in order, so a gap between two MFMAs hides at most the 64
// cycles the previous MFMA executes.  All non-MFMA work of a group therefore must not sit in ONE gap: it is dealt out over
// the four gaps of the group, each pinned with sched_barrier --
//     MFMA0 | A-fragment prefetch | MFMA1 | one DMA piece | MFMA2 | one slice of the previous slab's epilogue | MFMA3
// pending(i), i = 0..3: slice i (4 accumulator registers) of the previous slab's epilogue, run in groups 0..3.
// NP = number of 4 KB pieces of the slab staged at this slab's sync point (the slab two ahead): compile-time, so a DMA
// piece is m0 + address bump + global_load_lds with no compare/branch.
template <int NG0, int NG1, int GB, int NP, class Pending>
SN_DEV void slab_f32(f32x16& acc, f32x4 (&af)[2], f32x16& acc_pre, const char* lw, const float* b0, const float* b1,
                     const char* lw_next, const float* lds_bias, int s_next, int h, Ring& ring, Pending&& pending) {
  constexpr int NG = NG0 + NG1;
  constexpr int PPG = (NP + (NG - GB) - 1) / (NG - GB);      // DMA pieces per group after the sync point
  static_assert(GB >= 2 && GB % 2 == 0 && GB < NG && NG % 2 == 0, "sync point inside the slab; fragments come in pairs");
  // af[0], af[1] = A fragments of groups g, g+1 for even g: both are requested together two groups ahead (one s_waitcnt
  // per TWO groups instead of one per group).  Invariant at entry / exit: af = fragments of groups 0, 1 of the slab.
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const float* b = (g < NG0) ? (b0 + 4 * g) : (b1 + 4 * (g - NG0));
    if (g == GB) {                               // sync point (one longer gap per slab)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      ring.begin_static();
      acc_pre = load_bias(lds_bias, s_next, h);
      __builtin_amdgcn_sched_barrier(0);
    }
    const f32x4 a_cur = af[g & 1];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], b[0], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (g & 1) {                                 // gap 1 of odd groups: both fragments of the next pair of groups
      const int gn = g + 1;
      f32x4 n0, n1;
      if (gn < NG) {
        n0 = *reinterpret_cast<const f32x4*>(lw + gn * 1024);
        n1 = *reinterpret_cast<const f32x4*>(lw + (gn + 1) * 1024);
      } else {
        n0 = *reinterpret_cast<const f32x4*>(lw_next);
        n1 = *reinterpret_cast<const f32x4*>(lw_next + 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], b[1], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (g >= GB) {
#pragma unroll
        for (int j = 0; j < PPG; ++j) if ((g - GB) * PPG + j < NP) ring.piece_static();
      }
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], b[2], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (g < 4) pending(g);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], b[3], acc, 0, 0, 0);
      af[0] = n0; af[1] = n1;
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], b[1], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (g >= GB) {                             // gap 2: weight DMA
#pragma unroll
        for (int j = 0; j < PPG; ++j) if ((g - GB) * PPG + j < NP) ring.piece_static();
      }
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], b[2], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (g < 4) pending(g);                     // gap 3: epilogue slice of the previous slab
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], b[3], acc, 0, 0, 0);
    }
  }
  ring.template end_static<NP>();
}

// ReLU as ONE v_max_f32 (fmaxf() on an MFMA result makes hipcc emit a canonicalising v_max first: 2 VALU per value, and
// the epilogues of the bf16 path are VALU-issue-bound).
SN_DEV float relu1(float x) {
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}

// ShiftedSoftplus (models/activations.py:33-35) on hardware exp2/log2:  max(x-1,0) + log1p(exp(-|x-1|)).
// log1p(e) for e in (0,1]: u = 1+e; log(u) * e/(u-1) restores the bits lost in 1+e (u-1 is exact); e < 2^-24 -> e.
SN_DEV float shifted_softplus_fast(float x) {
  const float sx = x - 1.0f;
  const float e = __builtin_amdgcn_exp2f(-fabsf(sx) * 1.44269504088896340736f);
  const float u = 1.0f + e;
  const float d = u - 1.0f;
  const float l = __builtin_amdgcn_logf(u) * 0.69314718055994530942f;
  const float lp = (d == 0.0f) ? e : l * (e / d);
  return fmaxf(sx, 0.0f) + lp;
}

}  // namespace snk
