// sn_mlp_f32g.h -- the fragments-from-L2 slab loop shared by the round-6 fp32 kernels (sn_mlp_fwd_f32g.hip: inference + training forward;
// sn_mlp_bwd_f32g.hip: backward chain).  Why it looks the way it does: sn_mlp_fwd_f32g.hip's header (v_mfma_f32_32x32x2_f32 runs on the f32
// vector pipe: only VALU instructions cost matrix time, tools/ubench/f32_gap_cost.hip).
#pragma once
#include "sn_mlp_pipe.h"
#include <type_traits>

namespace snk {

// fragment ring depth, in groups of four k-steps (= 256 cycles of MFMAs each).  The stream of a point tile is 2 320 groups (1 920 for the
// sigma-only kernel): any divisor works, the ring index of a group is its stream index mod FD.
#ifndef SN_F32G_FD
#define SN_F32G_FD 8
#endif
constexpr int FD_INFER = SN_F32G_FD;
// training forward: the activation stores share the vector-memory counter with the fragment loads and retire after them (HBM write
// acknowledgements under 3 TB/s of stores: microseconds) -- a counted wait for a fragment also waits for every OLDER store, so the ring
// reaches 16 groups = 4 096 cycles ahead there
#ifndef SN_F32G_FD_STORE
#define SN_F32G_FD_STORE 16
#endif
constexpr int FD_STORE = SN_F32G_FD_STORE;
// steps of an epilogue program: slice q = step / 6 (write, four ReLU words, AGPR load) = 24; the training forward appends the row stores
// of the staged 32-point x 32-feature tile: two row-group reads ahead of four (store, next read) steps
constexpr int EPI_STEPS = 24, EPI_STEPS_STORE = 30;
// timing-build knobs (tools/build_variant_f32g.sh; results are WRONG with any of them set): SN_F32G_NO_ATOMICS leaves the ReLU's LDS
// integer max out, SN_F32G_NO_EPI the whole LDS round trip, SN_F32G_WAIT_G moves the epilogue's lgkmcnt wait, SN_F32G_NO_RAY_LOADS feeds
// constants instead of (rays, z_vals)
#ifndef SN_F32G_WAIT_G
#define SN_F32G_WAIT_G 6
#endif
constexpr int F32G_LDS_BYTES = TAIL_LDS_BYTES + EPI_LDS_BYTES;   // bias / head table + the epilogue staging of four waves
constexpr int F32G_LDS_BYTES_STORE = TAIL_LDS_BYTES + XPOSE_LDS_BYTES;   // ... training forward: the staging tiles of the activation stores

typedef __amdgpu_buffer_rsrc_t rsrc_t;

// one 1 KB A fragment (64 lanes x 16 B): wave-uniform byte offset in an SGPR, constant per-lane offset in a VGPR
SN_DEV f32x4 load_frag(rsrc_t rs, unsigned voff, unsigned soff) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return __builtin_bit_cast(f32x4, v);
}

// One slab = NG0 + NG1 groups of four k-steps (SET0 / SET1 / bv as in slab_f32a).  The packed weights of a point tile are ONE contiguous
// stream of TOT 1 KB fragments (slab after slab); G0 = stream index of this slab's first group.  fr = the fragment ring: entry i holds the
// fragment of the stream group == i (mod FD); on entry groups G0 .. G0 + FD - 2 are in flight or landed; group g's gap requests stream
// group G0 + g - 1 + FD (mod TOT: the stream wraps to the next point tile's slab 0) into the entry group g - 1 vacated.
// pending(step), step = 0 .. EPI_STEPS-1: the previous slab's epilogue as a PROGRAM OF SINGLE INSTRUCTIONS, one step behind every MFMA
// from the third on (the previous slab's last MFMA has retired by then) -- in the trunk every step is one LDS instruction.
// acc = this slab's accumulators (bias-initialised), accn = the previous slab's result until the steps have consumed it (the last
// reader is step 18, behind MFMA 20), then the bias of slab s_next (requested in group 6).
template <int NG0, int NG1, int SET0, int SET1, int G0, int TOT, int NSTEPS, int FD, class Pending>
SN_DEV void slab_f32g(f32x16& acc, f32x16& accn, f32x4 (&fr)[FD], rsrc_t rs, unsigned voff,
                      const float* bv, const float* lds_bias, int s_next, int h, Pending&& pending) {
  constexpr int NG = NG0 + NG1;
  static_assert(NG % 4 == 0 && TOT % FD == 0 && NG >= 8, "ring index = stream index mod FD");
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    // the epilogue's AGPR loads have landed before anything reads them: at the top of every slab (tiles 0..6 of a layer are read by the
    // NEXT layer) and, for the tile whose steps run in this very slab (the previous layer's tile 7, read from group 28 on), in group 12
    if (g == 0 || g == 12) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (g == 6) accn = load_bias(lds_bias, s_next, h);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 a_cur = fr[(G0 + g) % FD];
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
    auto step = [&](int kk) __attribute__((always_inline)) {
      const int st = 4 * g + kk - 2;
      if (st >= 0 && st < NSTEPS) pending(st);
    };
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    step(0);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
    // the fragment FD - 1 groups ahead, into the entry the previous group left
    fr[(G0 + g - 1 + FD) % FD] = load_frag(rs, voff, (unsigned)((G0 + g - 1 + FD) % TOT) * 1024u);
    step(1);
    __builtin_amdgcn_sched_barrier(0);
    mma(2);
    __builtin_amdgcn_sched_barrier(0);
    step(2);
    __builtin_amdgcn_sched_barrier(0);
    mma(3);
    __builtin_amdgcn_sched_barrier(0);
    step(3);
  }
}


}  // namespace snk
