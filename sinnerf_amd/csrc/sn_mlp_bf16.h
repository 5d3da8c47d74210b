// sn_mlp_bf16.h -- building blocks shared by the bf16-operand fused MLP kernels (sn_mlp_fwd_bf16.hip, sn_mlp_bwd_bf16.hip):
// inline-asm MFMAs with VGPR accumulators over the hand-managed AGPR activation file, epilogue blocks, the slab loop.
// See the header of sn_mlp_fwd_bf16.hip for the register plan and the hazard rules (checked by tools/check_agpr.py).
#pragma once
#ifndef SN_SPREAD_MEM
#define SN_SPREAD_MEM 1
#endif
#include "sn_mlp_pipe.h"
#include <type_traits>

namespace snk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
constexpr int PT = 2;                                       // point tiles per wave
constexpr int RING_SLOT_BYTES_BF16 = snl::MAX_SLAB_K * 64;  // 20480
constexpr int MLP_BF16_LDS_BYTES = TAIL_LDS_BYTES + 3 * RING_SLOT_BYTES_BF16;   // 73984
constexpr int BF16_XPOSE_LDS_BYTES = 4 * PT * XPOSE_WAVE_BYTES;                  // training forward: staging tiles (36864)
typedef RingT<64, RING_SLOT_BYTES_BF16> RingB;

SN_DEV uint32_t pack2(float a, float b) {        // {bf16(a), bf16(b)}, RNE (asm: hipcc converts the halves separately + v_perm)
  uint32_t d;
  asm(SN_CVT_PK " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
SN_DEV uint32_t relu_pk(uint32_t x) {            // ReLU on a packed bf16 pair: v_pk_max_i16 x, 0
  const i16x2 z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, x), z));
}
SN_DEV u32x4 pack8(const float* v) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pack2(v[2 * i], v[2 * i + 1]);
  return o;
}
// Epilogue blocks: four fp32 accumulator values -> two packed dwords of the hand-managed AGPR file, a[reg], a[reg+1]
// (`reg` must fold to a constant, it is printed into the asm text).  One asm per block: dependent instructions are one
// slot apart (the compiler pads every VALU <-> inline-asm dependence with an s_nop for the dst_sel forwarding hazard
// it has to assume), and as volatile asm they keep their program order relative to the MFMA asm -- which is what keeps
// the MFMA-result hazard distance.
// (the packed dwords t0, t1 = the bf16 pairs written to the AGPR file are returned: the bf16-state training variants
//  store exactly these values)
SN_DEV void epi_relu(int reg, float x0, float x1, float x2, float x3, uint32_t& t0, uint32_t& t1) {   // pack, ReLU on the pairs
  asm volatile(SN_CVT_PK " %0, %2, %3\n\t" SN_CVT_PK " %1, %4, %5\n\t"
               "v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0\n\t"
               "v_accvgpr_write_b32 a[%6], %0\n\tv_accvgpr_write_b32 a[%7], %1"
               : "=&v"(t0), "=&v"(t1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "n"(reg), "n"(reg + 1));
}
SN_DEV void epi_relu(int reg, float x0, float x1, float x2, float x3) {
  uint32_t t0, t1;
  epi_relu(reg, x0, x1, x2, x3, t0, t1);
}
// ... and the same with the ReLU SIGN BITS kept for the backward chain (bf16-state training forward).  The two packed words
// of a block are steps j, j+1 of a tile's 16 (both point tiles share one register: pt 0 = steps 0..7, pt 1 = 8..15):
//     bits = (bits >> 1) | (word & 0x80008000)
// leaves, after the 16th step, the sign of the LOW value of step j at bit j and of the HIGH value at bit 16 + j -- one
// 32-bit word per lane and output tile (32 features x 64 points = 256 B per wave) instead of the 4 KB of activations the
// chain otherwise re-reads for [h > 0].
SN_DEV void epi_relu_bits(int reg, float x0, float x1, float x2, float x3, uint32_t& t0, uint32_t& t1, uint32_t& bits) {
  const uint32_t sm = 0x80008000u;              // in an SGPR: v_and_or_b32 takes no literal on gfx9, and no VGPR is spent on it
  asm volatile(SN_CVT_PK " %0, %3, %4\n\t" SN_CVT_PK " %1, %5, %6\n\t"
               "v_lshrrev_b32 %2, 1, %2\n\tv_and_or_b32 %2, %0, %9, %2\n\t"
               "v_pk_max_i16 %0, %0, 0\n\tv_lshrrev_b32 %2, 1, %2\n\t"
               "v_and_or_b32 %2, %1, %9, %2\n\tv_pk_max_i16 %1, %1, 0\n\t"
               "v_accvgpr_write_b32 a[%7], %0\n\tv_accvgpr_write_b32 a[%8], %1"
               : "=&v"(t0), "=&v"(t1), "+v"(bits)
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "n"(reg), "n"(reg + 1), "s"(sm));
}
// fp32 ReLU form (layer 8: the fp32 outputs also feed the sigma head): the signs are those of the fp32 inputs
SN_DEV void epi_relu_f32_bits(int reg, float x0, float x1, float x2, float x3, float (&v)[4], uint32_t& t0, uint32_t& t1,
                              uint32_t& bits) {
  const uint32_t sm = 0x80008000u;
  asm volatile(SN_CVT_PK " %0, %7, %8\n\t" SN_CVT_PK " %1, %9, %10\n\t"
               "v_max_f32 %2, 0, %7\n\tv_max_f32 %3, 0, %8\n\tv_max_f32 %4, 0, %9\n\tv_max_f32 %5, 0, %10\n\t"
               "v_lshrrev_b32 %6, 1, %6\n\tv_and_or_b32 %6, %0, %13, %6\n\t"
               "v_lshrrev_b32 %6, 1, %6\n\t" SN_CVT_PK " %0, %2, %3\n\t"
               "v_and_or_b32 %6, %1, %13, %6\n\t" SN_CVT_PK " %1, %4, %5\n\t"
               "v_accvgpr_write_b32 a[%11], %0\n\tv_accvgpr_write_b32 a[%12], %1"
               : "=&v"(t0), "=&v"(t1), "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "+v"(bits)
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "n"(reg), "n"(reg + 1), "s"(sm));
}
SN_DEV void epi_copy(int reg, float x0, float x1, float x2, float x3, uint32_t& t0, uint32_t& t1) {   // no activation
  asm volatile(SN_CVT_PK " %0, %2, %3\n\t" SN_CVT_PK " %1, %4, %5\n\t"
               "v_accvgpr_write_b32 a[%6], %0\n\tv_accvgpr_write_b32 a[%7], %1"
               : "=&v"(t0), "=&v"(t1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "n"(reg), "n"(reg + 1));
}
SN_DEV void epi_copy(int reg, float x0, float x1, float x2, float x3) {
  uint32_t t0, t1;
  epi_copy(reg, x0, x1, x2, x3, t0, t1);
}
SN_DEV void epi_relu_f32(int reg, float x0, float x1, float x2, float x3, float (&v)[4], uint32_t& t0, uint32_t& t1) {   // fp32 ReLU
  asm volatile("v_max_f32 %2, 0, %6\n\tv_max_f32 %3, 0, %7\n\tv_max_f32 %4, 0, %8\n\tv_max_f32 %5, 0, %9\n\t"
               SN_CVT_PK " %0, %2, %3\n\t" SN_CVT_PK " %1, %4, %5\n\t"
               "v_accvgpr_write_b32 a[%10], %0\n\tv_accvgpr_write_b32 a[%11], %1"
               : "=&v"(t0), "=&v"(t1), "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "n"(reg), "n"(reg + 1));
}
SN_DEV void epi_relu_f32(int reg, float x0, float x1, float x2, float x3, float (&v)[4]) {   // ... kept for the sigma head
  uint32_t t0, t1;
  epi_relu_f32(reg, x0, x1, x2, x3, v, t0, t1);
}
// Training variants that keep the state in bf16: per-wave staging tile of packed rows (32 points x 32 features x 2 B,
// pitch 80 B), read back as 16-byte chunks -- one store instruction writes sixteen whole 64-byte rows.
// ... STORE side: two consecutive 32-feature tiles are staged side by side (a point row = 64 features = 128 B, pitch 144 B) and
// leave as whole 128-byte rows every second tile.  One tile at a time a row piece is 64 B -- half a cache line, written
// non-temporally by two different slabs: those partial-line stores cost the chain 0.41 of 1.18 ms and the training forward
// 0.27 of 1.24 ms (timing builds without the stores), while the fp32-state kernels' 128-byte rows are free.
constexpr int XS16_PITCH = 144;
// 8-byte write into a wave's staging tile as inline asm: hipcc guards every LDS WRITE it sees with s_waitcnt vmcnt(0) while
// LDS-DMA pieces may be in flight (it cannot tell the ring slots from the staging tile) -- at the first staging write of an
// epilogue that drained the row stores issued two k-steps earlier, a full HBM store round trip per slab.  `lds` = byte offset
// in LDS (the kernels have no static __shared__: dynamic LDS starts at 0), off = compile-time part.
// Two separate registers (ds_write2_b32, dword offsets off/4 and off/4 + 1 <= 255): no v_mov pair to build a 64-bit operand.
SN_DEV void lds_write_b64(unsigned lds, int off, uint32_t lo, uint32_t hi) {
  asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(lds), "v"(lo), "v"(hi), "n"(off / 4), "n"(off / 4 + 1) : "memory");
}
static_assert(32 * XS16_PITCH <= XPOSE_WAVE_BYTES, "the tile-pair staging fits the fp32 staging tile");
constexpr int XP16_PITCH = 80;
constexpr int XP16_WAVE_BYTES = 32 * XP16_PITCH;            // 2560
// D = A.B + D, D and A in VGPRs; B = a[reg : reg+3] ...
// FIRST = first MFMA of a slab on this accumulator: its C operand was just written by VALU moves / its B operands by the
// previous layer's v_accvgpr_write, and a VALU write -> MFMA read needs 2 wait states the compiler cannot insert for asm.
template <bool FIRST>
SN_DEV void mma_a(f32x16& acc, const u32x4& a, int reg) {
  if (FIRST) asm volatile("s_nop 1\n\t" SN_MFMA_16 " %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(a), "n"(reg), "n"(reg + 3));
  else asm volatile(SN_MFMA_16 " %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(a), "n"(reg), "n"(reg + 3));
}
// ... or B in VGPRs
template <bool FIRST>
SN_DEV void mma_v(f32x16& acc, const u32x4& a, const u32x4& b) {
  if (FIRST) asm volatile("s_nop 1\n\t" SN_MFMA_16 " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  else asm volatile(SN_MFMA_16 " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// MFMA (8 passes) -> VALU read of its result: the wait states the compiler would insert for a builtin MFMA
SN_DEV void mfma_result_fence() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory"); }
// ShiftedSoftplus (activations.py:33-35) of four dir_encoding outputs + their contribution to the rgb head, on PACKED fp32
// VALU ops: v = max(x-1, 0) + ln2 * log2(1 + exp2(-|x-1| * log2 e)) (log(1+e) taken directly: its 2^-24 absolute error is far
// below the bf16 rounding of the layer's inputs), rgb sums kept as (even, odd) pairs.  26 instructions per four values
// instead of 40 (v_pk_add / v_pk_mul / v_pk_fma; exp, log and max stay scalar) -- this section is pure VALU time between the
// MFMA phases of the kernel.  The values v are bit-identical with the scalar form; only the rgb sums associate as pairs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
SN_DEV void ssp4_rgb(const float (&x)[4], const f32x4 (&w)[3], f32x2 (&c)[3], float (&v)[4]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f32x2 sx = {x[2 * p], x[2 * p + 1]};
    sx = sx - 1.0f;
    const f32x2 t = sx * 1.44269504088896340736f;
    f32x2 e, l, m;
    e.x = __builtin_amdgcn_exp2f(-__builtin_fabsf(t.x));
    e.y = __builtin_amdgcn_exp2f(-__builtin_fabsf(t.y));
    const f32x2 u = e + 1.0f;
    l.x = __builtin_amdgcn_logf(u.x);
    l.y = __builtin_amdgcn_logf(u.y);
    m.x = __builtin_fmaxf(sx.x, 0.0f);
    m.y = __builtin_fmaxf(sx.y, 0.0f);
    const f32x2 ln2 = {0.69314718055994530942f, 0.69314718055994530942f};
    f32x2 vv = __builtin_elementwise_fma(l, ln2, m);
    if (!SN_NEWACT) {                            // classic heads: ReLU after dir_encoding (nerf.py:94)
      vv.x = __builtin_fmaxf(x[2 * p], 0.0f);
      vv.y = __builtin_fmaxf(x[2 * p + 1], 0.0f);
    }
    v[2 * p] = vv.x;
    v[2 * p + 1] = vv.y;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f32x2 wk = {w[k][2 * p], w[k][2 * p + 1]};
      c[k] = __builtin_elementwise_fma(wk, vv, c[k]);
    }
  }
}
SN_DEV float hsum(f32x2 a) { return a.x + a.y; }
constexpr int act_reg(int set, int kstep, int pt) { return set * 128 + (kstep * PT + pt) * 4; }

// One slab: NK0 + NK1 k-steps (two K segments), barrier after k-step GB.
//   SET0/SET1  B operands of the segment: AGPR activation set 0/1, or -1 = the VGPR array bv ([k-step][PT])
//   acc        accumulator set of this slab, bias-initialised on entry
//   accn       the other set: holds the previous slab's result until pending() has consumed it (after k-step 0), then
//              receives the bias of slab s_next at the sync point
//   af         4-entry ring of A fragments, prefetch distance 3 k-steps (a bf16 k-step is only 2 x 32 MFMA cycles, one
//              step of lookahead does not cover the LDS latency).  Invariant at entry: fragments of k-steps 0,1,2 of this
//              slab sit in af[(PHASE+0..2) & 3]; at exit the same holds for the next slab with PHASE' = (PHASE + NK) & 3
//              (NK % 4 == 0 everywhere except the four dir_encoding slabs, whose phases 0,2,0,2 are still static).
//   NBYTES     size of the slab staged at this slab's sync point (the slab two ahead): compile-time -> no DMA branches
//              (a K = 288 slab ends in a half piece that only waves 0,1 carry).
//   VMW        counted wait at the sync point: the youngest VMW memory operations (row stores issued behind the previous slab's
//              DMA pieces) stay in flight; the barrier is then a raw s_barrier (a __syncthreads() fence drains stores).  Together
//              with the staging writes and the chain's sign-word load as inline asm (hipcc guards LDS writes / mixed loads and
//              stores with vmcnt(0)) this removes the per-slab store drain: -3 % (forward) / -7 % (chain) GPU cycles, waves
//              parked 41 -> 36 % in the chain -- the wall time of these launches did not move on the boxes measured (1.0 ms
//              either way: the first attempt at this, round 2 run 4, was dropped for that reason).
//   post_sync  (step, n_steps): the memory steps of the training kernels (activation-tile row stores, then the loads of the
//              next tile's masks), called once per k-step behind the sync point and behind the k-steps that carry the slab's
//              DMA pieces: the callee deals its operations evenly over the n_steps calls -- one vector-memory instruction
//              per k-step and wave instead of a burst of nine at the sync point (sn_mlp_pipe.h: the fp32 chain gained 6 %
//              from that placement).  They still follow the sync point so that they have most of a slab to complete before
//              the next s_waitcnt vmcnt(0); issued just in FRONT of it they expose the full HBM latency on every slab
//              (measured: 3.9 us per slab instead of ~1).
template <int NK0, int NK1, int SET0, int SET1, int GB, int PHASE, int NBYTES, int VMW = 0, class Pending, class PostSync>
SN_DEV void slab_bf16(f32x16 (&acc)[PT], f32x16 (&accn)[PT], u32x4 (&af)[4], const char* lw, const u32x4* bv,
                      const char* lw_next, const float* lds_bias, int s_next, int h, RingB& ring, Pending&& pending,
                      PostSync&& post_sync) {
  constexpr int NK = NK0 + NK1;
  constexpr int NP = (NBYTES + 4095) / 4096;
  constexpr int PPK = (NP + (NK - GB) - 1) / (NK - GB);      // DMA pieces per k-step after the sync point (1; 2 in layer 0)
  constexpr int NPS = (NP + PPK - 1) / PPK;                  // k-steps that carry a DMA piece
#if SN_SPREAD_MEM
  constexpr int MS0 = (NK - GB - NPS >= 3) ? NPS : 0;        // the memory steps start behind them when that leaves >= 3 k-steps
  constexpr int NMS = NK - GB - MS0;
#else
  constexpr int MS0 = 0, NMS = 1;                            // (comparison build: everything right at the sync point)
#endif
  static_assert(GB >= 1 && GB < NK && NK >= 4 && GB + 3 <= NK, "sync point inside the slab, not after the first next-slab fragment read");
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    if (ks == GB) {
      // VMW > 0: the youngest VMW memory operations are row stores issued BEHIND the previous slab's DMA pieces -- they may
      // stay in flight (vmcnt retires in issue order); the barrier then must not be a fence (__syncthreads() drains stores)
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VMW) : "memory");
      if (VMW == 0) __syncthreads();
      else __builtin_amdgcn_s_barrier();
      ring.begin_static();
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) accn[pt] = load_bias(lds_bias, s_next, h);
    }
    {   // AFTER the sync point: in layer 0 (NK = 4, GB = 1) the fragment of k-step 1+3 already belongs to the NEXT slab,
        // which is only guaranteed to have landed once this slab's barrier has been passed
      const int kn = ks + 3;
      af[(PHASE + kn) & 3] = (kn < NK) ? *reinterpret_cast<const u32x4*>(lw + kn * 1024)
                                       : *reinterpret_cast<const u32x4*>(lw_next + (kn - NK) * 1024);
    }
    if (ks >= GB) {
#pragma unroll
      for (int i = 0; i < PPK; ++i) {
        const int piece = (ks - GB) * PPK + i;
        if (piece < NP) {
          if ((piece + 1) * 4096 <= NBYTES || ring.wbase < NBYTES - piece * 4096) ring.piece_static();   // wave-uniform
          else ring.skip_static();
        }
      }
    }
    if (ks >= GB + MS0 && ks - GB - MS0 < NMS) post_sync(ks - GB - MS0, NMS);
    __builtin_amdgcn_sched_barrier(0);
    const u32x4 a_cur = af[(PHASE + ks) & 3];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      if (ks == 0) {
        if (SET0 < 0) mma_v<true>(acc[pt], a_cur, bv[pt]); else mma_a<true>(acc[pt], a_cur, act_reg(SET0, 0, pt));
      } else if (ks < NK0) {
        if (SET0 < 0) mma_v<false>(acc[pt], a_cur, bv[ks * PT + pt]); else mma_a<false>(acc[pt], a_cur, act_reg(SET0, ks, pt));
      } else {
        if (SET1 < 0) mma_v<false>(acc[pt], a_cur, bv[(ks - NK0) * PT + pt]);
        else mma_a<false>(acc[pt], a_cur, act_reg(SET1, ks - NK0, pt));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ks == 0) {
      pending();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  ring.template end_static_bytes<NBYTES>();
}

}  // namespace snk
