// sn_mlp_x3.h -- building blocks of the bf16x3 kernels (sn_mlp_fwd_bf16x3.hip, sn_mlp_bwd_bf16x3.hip): fp32-level accuracy on the
// bf16 MFMA from the 3-term split  W.x ~= Wh.xh + Wl.xh + Wh.xl  of (hi, lo) bf16 operand pairs, fp32 accumulation.
// A wave owns ONE 32-point tile; the hand-managed AGPR file holds two activation sets x (hi, lo) x 64 registers.
#pragma once
#include "sn_mlp_bf16.h"

namespace snk {

constexpr int X3_LDS_BYTES = MLP_F32_LDS_BYTES_V2;                      // tail + 3 x 40 KB
// AGPR of (activation set, part 0 = hi / 1 = lo, k-step): 4 registers each
constexpr int x3_reg(int set, int part, int ks) { return set * 128 + part * 64 + ks * 4; }

// ---- MFMAs: D (+)= A.B ; A fragment in VGPRs, B in AGPRs (immediates) or VGPRs; ZERO = chain B's first MFMA (C = 0)
template <bool FIRST, bool ZERO>
SN_DEV void x3_mma_a(f32x16& acc, const u32x4& a, int reg) {
  if (ZERO) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], 0" : "=&v"(acc) : "v"(a), "n"(reg), "n"(reg + 3));
  else if (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(a), "n"(reg), "n"(reg + 3));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(a), "n"(reg), "n"(reg + 3));
}
template <bool FIRST, bool ZERO>
SN_DEV void x3_mma_v(f32x16& acc, const u32x4& a, const u32x4& b) {
  if (ZERO) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
  else if (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// MFMA (8 passes) -> VALU read of its result at a layer end: the wait states the compiler would insert for a builtin MFMA.  The two
// chains are operands: plain C++ arithmetic on them (A + B) could otherwise be scheduled above the wait (tools/check_agpr.py).
SN_DEV void x3_result_fence(f32x16& a, f32x16& b) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(a), "+v"(b)); }

// (hi, lo) split of four fp32 values into packed bf16 pairs: h = RNE(x), l = RNE(x - float(h))
SN_DEV void x3_split4(const float (&x)[4], uint32_t& h0, uint32_t& h1, uint32_t& l0, uint32_t& l1) {
  float r0, r1, r2, r3;
  asm("v_cvt_pk_bf16_f32 %0, %8, %9\n\tv_cvt_pk_bf16_f32 %1, %10, %11\n\t"
      "v_lshlrev_b32 %4, 16, %0\n\tv_and_b32 %5, 0xffff0000, %0\n\tv_lshlrev_b32 %6, 16, %1\n\tv_and_b32 %7, 0xffff0000, %1\n\t"
      "v_sub_f32 %4, %8, %4\n\tv_sub_f32 %5, %9, %5\n\tv_sub_f32 %6, %10, %6\n\tv_sub_f32 %7, %11, %7\n\t"
      "v_cvt_pk_bf16_f32 %2, %4, %5\n\tv_cvt_pk_bf16_f32 %3, %6, %7"
      : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
}
SN_DEV void x3_split8(const float* f, u32x4& hi, u32x4& lo) {
  const float a[4] = {f[0], f[1], f[2], f[3]}, b[4] = {f[4], f[5], f[6], f[7]};
  uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
  x3_split4(a, h0, h1, l0, l1);
  x3_split4(b, h2, h3, l2, l3);
  hi = u32x4{h0, h1, h2, h3};
  lo = u32x4{l0, l1, l2, l3};
}

// Epilogue block: accumulator registers 4i..4i+3 of both chains -> v = act(A + B) (RELU: max with 0), its hi / lo pairs into
// a[rh], a[rh+1] / a[rl], a[rl+1].  One volatile asm: program order relative to the MFMA asm is what keeps the hazard distances.
template <bool RELU>
SN_DEV void x3_epi(int rh, int rl, const float (&a)[4], const float (&b)[4], float (&v)[4], uint32_t& h0, uint32_t& h1, uint32_t& l0,
                   uint32_t& l1) {
  float r0, r1, r2, r3;
  if (RELU)
    asm volatile("v_add_f32 %0, %12, %16\n\tv_add_f32 %1, %13, %17\n\tv_add_f32 %2, %14, %18\n\tv_add_f32 %3, %15, %19\n\t"
                 "v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3\n\t"
                 "v_cvt_pk_bf16_f32 %4, %0, %1\n\tv_cvt_pk_bf16_f32 %5, %2, %3\n\t"
                 "v_lshlrev_b32 %8, 16, %4\n\tv_and_b32 %9, 0xffff0000, %4\n\tv_lshlrev_b32 %10, 16, %5\n\tv_and_b32 %11, 0xffff0000, %5\n\t"
                 "v_accvgpr_write_b32 a[%20], %4\n\tv_accvgpr_write_b32 a[%21], %5\n\t"
                 "v_sub_f32 %8, %0, %8\n\tv_sub_f32 %9, %1, %9\n\tv_sub_f32 %10, %2, %10\n\tv_sub_f32 %11, %3, %11\n\t"
                 "v_cvt_pk_bf16_f32 %6, %8, %9\n\tv_cvt_pk_bf16_f32 %7, %10, %11\n\t"
                 "v_accvgpr_write_b32 a[%22], %6\n\tv_accvgpr_write_b32 a[%23], %7"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1),
                   "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
                   "n"(rh), "n"(rh + 1), "n"(rl), "n"(rl + 1));
  else
    asm volatile("v_add_f32 %0, %12, %16\n\tv_add_f32 %1, %13, %17\n\tv_add_f32 %2, %14, %18\n\tv_add_f32 %3, %15, %19\n\t"
                 "v_cvt_pk_bf16_f32 %4, %0, %1\n\tv_cvt_pk_bf16_f32 %5, %2, %3\n\t"
                 "v_lshlrev_b32 %8, 16, %4\n\tv_and_b32 %9, 0xffff0000, %4\n\tv_lshlrev_b32 %10, 16, %5\n\tv_and_b32 %11, 0xffff0000, %5\n\t"
                 "v_accvgpr_write_b32 a[%20], %4\n\tv_accvgpr_write_b32 a[%21], %5\n\t"
                 "v_sub_f32 %8, %0, %8\n\tv_sub_f32 %9, %1, %9\n\tv_sub_f32 %10, %2, %10\n\tv_sub_f32 %11, %3, %11\n\t"
                 "v_cvt_pk_bf16_f32 %6, %8, %9\n\tv_cvt_pk_bf16_f32 %7, %10, %11\n\t"
                 "v_accvgpr_write_b32 a[%22], %6\n\tv_accvgpr_write_b32 a[%23], %7"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1),
                   "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
                   "n"(rh), "n"(rh + 1), "n"(rl), "n"(rl + 1));
}

template <bool RELU>
SN_DEV void x3_epi(int rh, int rl, const float (&a)[4], const float (&b)[4], float (&v)[4]) {
  uint32_t h0, h1, l0, l1;
  x3_epi<RELU>(rh, rl, a, b, v, h0, h1, l0, l1);
}
// ReLU sign bits of a packed pair of post-ReLU bf16 values (>= 0: "positive" is "bits != 0") into a per-lane word: bit k <- the low
// value, bit 16 + k <- the high value.  c01 = 0x00010001 in a register (VOP3P takes no literal).  The training forward leaves one word
// per lane and PAIR of 32-feature tiles (k = pair index 0..7 + 8 x (tile & 1)) for the backward chain: 256 B per point and layer
// instead of the 1 KB of activations (the bf16-state kernels' idea, sn_mlp_bf16.h epi_relu_bits, in this kernel's tile shape).
SN_DEV void x3_sign_bits(uint32_t& word, uint32_t pk, int k, uint32_t c01) {
  uint32_t m;
  asm("v_pk_min_u16 %0, %2, %3\n\tv_lshl_or_b32 %1, %0, %4, %1" : "=&v"(m), "+v"(word) : "v"(pk), "v"(c01), "n"(k));
}
// ... and the chain's use of them: v = (bit ? x : 0) for the four values of a block (pairs d, d + 1 of tile parity par), then as x3_put
SN_DEV void x3_put_signed(int rh, int rl, const float (&x)[4], uint32_t word, int k, float (&v)[4], uint32_t& h0, uint32_t& h1,
                          uint32_t& l0, uint32_t& l1) {
  float r0, r1, r2, r3;
  asm volatile("v_bfe_i32 %4, %16, %17, 1\n\tv_bfe_i32 %5, %16, %18, 1\n\tv_bfe_i32 %6, %16, %19, 1\n\tv_bfe_i32 %7, %16, %20, 1\n\t"
               "v_and_b32 %8, %4, %12\n\tv_and_b32 %9, %5, %13\n\tv_and_b32 %10, %6, %14\n\tv_and_b32 %11, %7, %15\n\t"
               "v_cvt_pk_bf16_f32 %0, %8, %9\n\tv_cvt_pk_bf16_f32 %1, %10, %11\n\t"
               "v_lshlrev_b32 %4, 16, %0\n\tv_and_b32 %5, 0xffff0000, %0\n\tv_lshlrev_b32 %6, 16, %1\n\tv_and_b32 %7, 0xffff0000, %1\n\t"
               "v_accvgpr_write_b32 a[%21], %0\n\tv_accvgpr_write_b32 a[%22], %1\n\t"
               "v_sub_f32 %4, %8, %4\n\tv_sub_f32 %5, %9, %5\n\tv_sub_f32 %6, %10, %6\n\tv_sub_f32 %7, %11, %7\n\t"
               "v_cvt_pk_bf16_f32 %2, %4, %5\n\tv_cvt_pk_bf16_f32 %3, %6, %7\n\t"
               "v_accvgpr_write_b32 a[%23], %2\n\tv_accvgpr_write_b32 a[%24], %3"
               : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3),
                 "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(word), "n"(k), "n"(k + 16), "n"(k + 1), "n"(k + 17),
                 "n"(rh), "n"(rh + 1), "n"(rl), "n"(rl + 1));
}

// One slab: NK0 + NK1 k-steps (two K segments), barrier after k-step GB (the 3-slot protocol of sn_mlp_pipe.h).
//   SET0/SET1  B operands of the segment: AGPR activation set 0/1, or -1 = the VGPR arrays bh / bl ([k-step])
//   accA/accB  the two chains of this slab (accA bias-initialised on entry, accB started by its first MFMA with C = 0)
//   nA         chain A of the NEXT slab: receives that slab's bias behind k-step 4 (pending(0..3) have consumed the previous
//              slab's results -- which live in nA / nB -- behind k-steps 0..3)
//   af         ring of A-fragment PAIRS (hi, lo), prefetch distance 3 k-steps: fragments of k-steps 0, 1, 2 of this slab sit in
//              af[(PHASE + 0..2) & 3] at entry; PHASE' = (PHASE + NK) & 3 at exit
//   NBYTES     size of the slab staged at this slab's sync point (the slab two ahead), a multiple of 4 KB for every K
//   VMW        counted wait at the sync point (training kernels): the youngest VMW vector-memory operations of the wave at the START of
//              the slab are row stores issued BEHIND the previous slab's DMA pieces and may stay in flight; barriers are raw s_barriers
//   post(ks, NK, ST0, before)  memory operations of the caller, once per k-step behind the sync point (ST0: first k-step for row stores): with
//              before = true in FRONT of the k-step's DMA pieces (the chain's mask loads), with false BEHIND them (row stores:
//              x3_store_step deals the four row-group stores of the previous tile over k-steps ST0 .. NK - 1)
template <int NK0, int NK1, int SET0, int SET1, int GB, int PHASE, int NBYTES, int VMW = 0, class RingX, class Pending, class Post>
SN_DEV void slab_x3(f32x16& accA, f32x16& accB, f32x16& nA, u32x4 (&af)[4][2], const char* lw, const u32x4* bh, const u32x4* bl,
                    const char* lw_next, const float* lds_bias, int s_next, int h, RingX& ring, Pending&& pending, Post&& post) {
  constexpr int NK = NK0 + NK1;
  constexpr int NP = NBYTES / 4096;
  constexpr int PPK = (NP + (NK - GB) - 1) / (NK - GB);
  constexpr int NPS = (NP + PPK - 1) / PPK;                  // k-steps that carry DMA pieces: GB .. GB + NPS - 1
  static_assert(NBYTES % 4096 == 0 && GB >= 1 && GB + 3 <= NK && NK >= 4, "whole pieces; sync point inside the slab");
  static_assert(NK >= 8 ? GB + NPS <= NK : true, "one piece per k-step behind the sync point");
  // Two sync points per slab (3-slot ring): with ONE (wait + barrier at k-step GB) a slab's weights are requested exactly one slab
  // ahead of the wait -- 0.75 us at 48 MFMAs per slab, below the ~1.1 us an LDS-DMA piece takes to land under this load: the waves were
  // parked 20-24 % of the time (profiles/r04_x3_train_kernels.txt).  Splitting the jobs -- B1 early ("slot free": start the DMA),
  // B2 late ("next slab visible", just before its first fragments are prefetched at k-step NK - 3) -- gives the DMA 1.6 slabs.
  constexpr int GB2 = (NK >= 8) ? NK - 4 : GB;
  constexpr int ISSUED = ((GB2 - GB) * PPK < NP) ? (GB2 - GB) * PPK : NP;      // pieces of slab s+2 issued before B2
  static_assert(GB2 >= GB && GB2 + 3 <= NK, "B2 in front of the first prefetch of the next slab's fragments");
  // first k-step that may carry row stores of the caller: behind the slab's last DMA piece (x3_store_step), not before NK - 4 / the sync point
  constexpr int LASTP = NP > 0 ? GB + NPS - 1 : GB;
  constexpr int ST0 = (LASTP > NK - 4 ? LASTP : NK - 4) > GB ? (LASTP > NK - 4 ? LASTP : NK - 4) : GB;
  static_assert(ST0 >= LASTP && ST0 >= GB && ST0 < NK, "row stores behind the slab's last DMA piece: what VMW = 4 at the next sync point counts on");
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    if (ks == GB) {
      // B1: every wave has left slab s-1 -> its slot takes slab s+2 (one barrier does both jobs in the 4-k-step slabs)
      if (GB2 == GB) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VMW) : "memory");
      __builtin_amdgcn_s_barrier();
      ring.begin_static();
    }
    if (GB2 != GB && ks == GB2) {
      // B2: slab s+1 (requested at B1 of slab s-1) has landed for every wave.  Younger than its pieces: the previous slab's row stores
      // (VMW) and the pieces of slab s+2 issued since B1
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VMW + ISSUED) : "memory");
      __builtin_amdgcn_s_barrier();
    }
    const u32x4 a_hi = af[(PHASE + ks) & 3][0], a_lo = af[(PHASE + ks) & 3][1];
    // even k-steps: A += Wh.xh ; B += Wl.xh ; A += Wh.xl        odd: B += Wh.xh ; A += Wl.xh ; B += Wh.xl
    const bool even = (ks & 1) == 0;
    f32x16& c0 = even ? accA : accB;
    f32x16& c1 = even ? accB : accA;
    const bool seg0 = ks < NK0;
    const int kk = seg0 ? ks : ks - NK0;
    const int set = seg0 ? SET0 : SET1;
    // Everything else of a k-step sits in the shadows of its three MFMAs (in-order issue: what is issued behind the third MFMA runs
    // with the pipe empty as soon as that one retires):
    //     MFMA 1 | fragment prefetch (k-step + 3), DMA pieces, row-store / mask-load steps | MFMA 2 | an epilogue block (k-steps 0..3)
    //     or the next slab's bias (k-step 4) | MFMA 3
    __builtin_amdgcn_sched_barrier(0);
    if (ks == 0) {
      if (set < 0) x3_mma_v<true, false>(c0, a_hi, bh[kk]); else x3_mma_a<true, false>(c0, a_hi, x3_reg(set, 0, kk));
    } else {
      if (set < 0) x3_mma_v<false, false>(c0, a_hi, bh[kk]); else x3_mma_a<false, false>(c0, a_hi, x3_reg(set, 0, kk));
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool dma = ks >= GB && (ks - GB) * PPK < NP;       // this k-step carries DMA pieces; the first one's m0 write goes in FRONT of
    if (dma) ring.piece_m0();                                // the fragment reads (they are the wait state between it and the load)
    {
      const int kn = ks + 3;
      const char* src = (kn < NK) ? lw + kn * 2048 : lw_next + (kn - NK) * 2048;
      af[(PHASE + kn) & 3][0] = *reinterpret_cast<const u32x4*>(src);
      af[(PHASE + kn) & 3][1] = *reinterpret_cast<const u32x4*>(src + 1024);
    }
    if (ks >= GB) {
      post(ks, NK, ST0, true);
      if (dma) ring.piece_load();
#pragma unroll
      for (int i = 1; i < PPK; ++i)
        if ((ks - GB) * PPK + i < NP) ring.piece_static();
      post(ks, NK, ST0, false);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ks == 0) {
      if (set < 0) x3_mma_v<false, true>(c1, a_lo, bh[kk]); else x3_mma_a<false, true>(c1, a_lo, x3_reg(set, 0, kk));
    } else {
      if (set < 0) x3_mma_v<false, false>(c1, a_lo, bh[kk]); else x3_mma_a<false, false>(c1, a_lo, x3_reg(set, 0, kk));
    }
    __builtin_amdgcn_sched_barrier(0);
    // the previous tile's deferred epilogue: one block of four accumulator registers in each of the first four k-steps (the 4-k-step
    // slabs keep all four blocks behind k-step 0: their row stores start at k-step 1).  The next slab's bias goes into the vacated
    // chain A behind the last block.
    if (NK >= 8) {
      if (ks < 4) pending(ks);
      if (ks == 4) nA = load_bias(lds_bias, s_next, h);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (set < 0) x3_mma_v<false, false>(c0, a_hi, bl[kk]); else x3_mma_a<false, false>(c0, a_hi, x3_reg(set, 1, kk));
    __builtin_amdgcn_sched_barrier(0);
    if (NK < 8) {
      if (ks == 0) { pending(0); pending(1); pending(2); pending(3); }
      if (ks == 1) nA = load_bias(lds_bias, s_next, h);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  ring.template end_static<NP>();
}

// 16-byte write into a wave's staging tile as inline asm (hipcc guards every LDS write it sees with s_waitcnt vmcnt(0) while
// LDS-DMA pieces may be in flight, sn_mlp_bf16.h): lds = byte address in LDS (dynamic LDS starts at 0), off = compile-time part
SN_DEV void x3_lds_write_b128(unsigned lds, int off, const float (&v)[4]) {
  f32x4 o;
  o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(lds), "v"(o), "n"(off) : "memory");
}


// ... of the (hi, lo) pairs of four consecutive features of one point in the SPLIT state layout (sn_layout.h "x3 state": per 8 features
// 16 B of hi parts, then 16 B of lo parts): lds = the lane's staging address  row + 8 (lane >> 5), off = 32 x block
SN_DEV void x3_lds_write_split(unsigned lds, int off, uint32_t h0, uint32_t h1, uint32_t l0, uint32_t l1) {
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  const u32x2_ hh = {h0, h1}, ll = {l0, l1};
  asm volatile("ds_write_b64 %0, %1 offset:%3\n\tds_write_b64 %0, %2 offset:%4" :: "v"(lds), "v"(hh), "v"(ll), "n"(off), "n"(off + 16) : "memory");
}

// four fp32 values -> their (hi, lo) bf16 pairs in a[rh], a[rh+1] / a[rl], a[rl+1] (no activation: values computed on the VALU)
SN_DEV void x3_put(int rh, int rl, const float (&x)[4], uint32_t& h0, uint32_t& h1, uint32_t& l0, uint32_t& l1) {
  float r0, r1, r2, r3;
  asm volatile("v_cvt_pk_bf16_f32 %0, %8, %9\n\tv_cvt_pk_bf16_f32 %1, %10, %11\n\t"
               "v_lshlrev_b32 %4, 16, %0\n\tv_and_b32 %5, 0xffff0000, %0\n\tv_lshlrev_b32 %6, 16, %1\n\tv_and_b32 %7, 0xffff0000, %1\n\t"
               "v_accvgpr_write_b32 a[%12], %0\n\tv_accvgpr_write_b32 a[%13], %1\n\t"
               "v_sub_f32 %4, %8, %4\n\tv_sub_f32 %5, %9, %5\n\tv_sub_f32 %6, %10, %6\n\tv_sub_f32 %7, %11, %7\n\t"
               "v_cvt_pk_bf16_f32 %2, %4, %5\n\tv_cvt_pk_bf16_f32 %3, %6, %7\n\t"
               "v_accvgpr_write_b32 a[%14], %2\n\tv_accvgpr_write_b32 a[%15], %3"
               : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "n"(rh), "n"(rh + 1), "n"(rl), "n"(rl + 1));
}
SN_DEV void x3_put(int rh, int rl, const float (&x)[4]) {
  uint32_t h0, h1, l0, l1;
  x3_put(rh, rl, x, h0, h1, l0, l1);
}
// the four row-group stores of a finished tile over the memory steps of the next slab.  THE RULE the counted waits rest on (VMW of
// slab_x3): a slab issues its row stores BEHIND ITS LAST DMA PIECE -- in k-steps st0 = max(k-step of the last piece, NK - 4) .. NK - 1,
// behind that k-step's pieces -- so that at the next slab's sync point exactly these four stores are younger than the pieces the wait is
// for.  (Until round 4 the 4- and 8-k-step slabs dealt their stores 1 + 1 + 2 / 1 + 1 + 1 + 1 from the sync point on, BETWEEN their
// pieces: vmcnt(4) then let the last two or three pieces of the next slab's weights stay in flight across the barrier -- a race that
// showed as run-to-run differences of a few 1e-5 in a training render, tools/x3_determinism.py.)
// Where the slab is long enough a row group leaves the staging tile one k-step AHEAD of its store (rd(i) then, a k-step on, wr(i)):
// read + store in one step waits for the LDS latency in front of the store.  rd(0) never before k-step 4: pending(0..3) fill the tile
// behind k-steps 0..3.
template <class R, class W>
SN_DEV void x3_store_step(int ks, int nk, int st0, R&& rd, W&& wr) {
  const int m = nk - st0;                        // k-steps that carry stores (1..4)
  if (m == 4 && st0 >= 5) {                      // rd(0) | wr(0) rd(1) | wr(1) rd(2) | wr(2) rd(3) | wr(3)
    const int j = ks - (st0 - 1);
    if (j >= 1) wr(j - 1);
    if (j >= 0 && j < 4) rd(j);
  } else if (ks >= st0) {
    const int per = (4 + m - 1) / m, j = ks - st0;
    for (int i = j * per; i < (j + 1) * per && i < 4; ++i) { rd(i); wr(i); }      // (one row buffer: read, store, read, store)
  }
}

}  // namespace snk
