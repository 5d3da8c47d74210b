// sn_mlp_bwd_bf16.hip -- backward "chain" of the fused NeRF MLP for gfx950 with bf16-operand contractions (mixed-precision
// training): input-gradient propagation  g_x = W^T g_y,  g_y = g_h (.) act'(.)  through every layer -- what torch autograd
// derives from models/nerf.py:122-148 (+ models/activations.py) -- on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
// Same machinery as the bf16 forward (sn_mlp_bf16.h): a wave owns two 32-point tiles, the gradient w.r.t. a layer's output
// is packed to bf16 straight from the fp32 accumulators into one of the two hand-managed AGPR sets and is the B operand of
// the next transposed layer; transposed bf16 weights stream L2 -> LDS through the 3-slot ring (sn_layout.h, "Backward-chain
// blob, bf16 operands").  The two narrow transposed heads run on the VALU in fp32: g_h2 = W_rgb^T g_y3 (3 FMAs per value)
// and the sigma term of g_h8 (1 FMA per value).
//
// Like the fp32 chain (sn_mlp_bwd.hip) the kernel WRITES the fp32 pre-activation gradients g_y of every layer
// (G[10][slot_rows][256], whole 128-byte rows through the per-wave staging tiles, non-temporal) -- the left operands of the
// weight-gradient contractions -- and reads the stored fp32 forward activations for the derivative masks:
//   ReLU (nerf.py:73) [h > 0];  ShiftedSoftplus (act.py:33) 1 - exp(-h2);  WidenedSigmoid (act.py:18) .2505 (1 - t^2).
#include "sn_mlp_bf16.h"
#ifndef SN_BF16_COUNTED
#define SN_BF16_COUNTED 1      // counted vmcnt waits of the bf16-state chain (0: comparison build)
#endif

namespace snk {

constexpr int BWD16_RING_SLOT = RING_SLOT_BYTES_BF16;                      // the forward's ring type (slots of 20480 B; widest slab here 16384)
constexpr int BWD16_TAIL_BYTES = snl::BB_TAIL_FLOATS * 4;                  // 11776: zero "bias" slots + aux table
constexpr int BWD16_LDS_BYTES = BWD16_TAIL_BYTES + 3 * BWD16_RING_SLOT + 4 * PT * XPOSE_WAVE_BYTES     // 110080
                                + 4 * PT * XP16_WAVE_BYTES;   // + per-wave staging of the incoming activation tiles (bf16 state): 130560

// masked epilogue block: four accumulator values x, four activations a -> v = (a > 0 ? x : 0) (returned for the store),
// packed to bf16 into a[reg], a[reg+1]
SN_DEV void epi_mask(int reg, float x0, float x1, float x2, float x3, float a0, float a1, float a2, float a3, float (&v)[4],
                     uint32_t& t0, uint32_t& t1) {
  asm volatile("v_cmp_lt_f32 vcc, 0, %10\n\tv_cndmask_b32 %2, 0, %6, vcc\n\t"
               "v_cmp_lt_f32 vcc, 0, %11\n\tv_cndmask_b32 %3, 0, %7, vcc\n\t"
               "v_cmp_lt_f32 vcc, 0, %12\n\tv_cndmask_b32 %4, 0, %8, vcc\n\t"
               "v_cmp_lt_f32 vcc, 0, %13\n\tv_cndmask_b32 %5, 0, %9, vcc\n\t"
               "v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5\n\t"
               "v_accvgpr_write_b32 a[%14], %0\n\tv_accvgpr_write_b32 a[%15], %1"
               : "=&v"(t0), "=&v"(t1), "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(reg), "n"(reg + 1)
               : "vcc");
}

// bf16 state: the mask comes straight from the PACKED activation pair (post-ReLU values are >= 0, so "h > 0" is "bits != 0"):
// min(u, 1) per half -> 0 / 1, 0 - that -> 0x0000 / 0xffff, AND with the packed gradient pair: 2.5 instructions per value
// instead of 4.5 (unpack, compare, select, convert).  c01 = 0x00010001.
SN_DEV void epi_mask16(int reg, float x0, float x1, float x2, float x3, uint32_t u0, uint32_t u1, uint32_t c01,
                       uint32_t& t0, uint32_t& t1) {
  uint32_t m0, m1;
  asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %6, %7\n\t"
               "v_pk_min_u16 %2, %8, %10\n\tv_pk_min_u16 %3, %9, %10\n\t"
               "v_pk_sub_u16 %2, 0, %2\n\tv_pk_sub_u16 %3, 0, %3\n\t"
               "v_and_b32 %0, %0, %2\n\tv_and_b32 %1, %1, %3\n\t"
               "v_accvgpr_write_b32 a[%11], %0\n\tv_accvgpr_write_b32 a[%12], %1"
               : "=&v"(t0), "=&v"(t1), "=&v"(m0), "=&v"(m1)
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(u0), "v"(u1), "v"(c01), "n"(reg), "n"(reg + 1));
}

// bf16 state, ReLU layers: the mask comes from the SIGN WORD the training forward left for this (layer, tile) (sn_mlp_bf16.h
// epi_relu_bits: one dword per lane; the low / high value of the packed pair of step j at bits j / 16 + j): shift the pair's
// two bits down, isolate them (c01 = 0x00010001), subtract 1 per half -> 0xffff where the forward value was positive, AND.
// Step order: hidden layers 8 pt + q (point tile outermost in the forward's epilogue), layer 8 (sigma epilogue) 2 q + 2 pt.
SN_DEV void epi_maskbits(int reg, int j0, float x0, float x1, float x2, float x3, uint32_t mw, uint32_t c01,
                         uint32_t& t0, uint32_t& t1) {
  uint32_t m0, m1;
  asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %6, %7\n\t"
               "v_lshrrev_b32 %2, %10, %8\n\tv_lshrrev_b32 %3, %11, %8\n\t"
               "v_and_b32 %2, %2, %9\n\tv_and_b32 %3, %3, %9\n\t"
               "v_pk_sub_u16 %2, %2, %9\n\tv_pk_sub_u16 %3, %3, %9\n\t"
               "v_and_b32 %0, %0, %2\n\tv_and_b32 %1, %1, %3\n\t"
               "v_accvgpr_write_b32 a[%12], %0\n\tv_accvgpr_write_b32 a[%13], %1"
               : "=&v"(t0), "=&v"(t1), "=&v"(m0), "=&v"(m1)
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(mw), "v"(c01), "n"(j0), "n"(j0 + 1), "n"(reg), "n"(reg + 1));
}

// S16: acts and G are bf16 arrays (SN_DTYPE_BF16_STATE): half the HBM traffic of this bandwidth-bound kernel; G then holds
// exactly the bf16 values the next transposed layer and the weight-gradient kernel consume.  The ReLU masks of layers 1..8
// are read as sign words from the unused half of acts slot 9 (256 B per point instead of 4 KB: sn_mlp_fwd_bf16.hip).
template <bool S16>
__global__ void __launch_bounds__(256)
mlp_bwd_chain_bf16_kernel(const char* __restrict__ bblob, const float* __restrict__ acts, const float* __restrict__ out_raw,
                          const float* __restrict__ g_raw, long P, long slot_rows, float* __restrict__ G,
                          float* __restrict__ g_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_zero = reinterpret_cast<float*>(smem);                       // the slab loop's "bias" slots: all zero
  const float* lds_aux = lds_zero + snl::BB_ZERO_FLOATS;
  asm volatile("" ::: "a0", "a255");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  constexpr int TILE_PTS = 4 * PT * 32;
  const long n_tiles = (P + TILE_PTS - 1) / TILE_PTS;

  RingB ring;
  ring.blob = bblob;
  ring.gnext = bblob;
  ring.base = smem + BWD16_TAIL_BYTES;
  ring.n_used = snl::NBB_SLABS;
  ring.stage_id = 0;
  ring.stage_slot = 0;
  ring.remaining = 0;
  ring.tid = tid;
  ring.wbase = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
  ring.pieces = 0; ring.piece = 0; ring.slab_bytes = 0;
  constexpr int B_D = 128 * 64, B_H = 256 * 64;                            // slab bytes: DIRT, FINT / LT
#pragma unroll
  for (int i = 0; i < 2; ++i) {                                            // slabs 0, 1 (both DIRT)
    ring.begin_static();
    ring.piece_static(); ring.piece_static();
    ring.template end_static_bytes<B_D>();
  }
  {
    const float4* gb = reinterpret_cast<const float4*>(bblob + snl::bb_tail_byte_offset());
    float4* lb = reinterpret_cast<float4*>(lds_zero);
    for (int i = tid; i < snl::BB_TAIL_FLOATS / 4; i += 256) lb[i] = gb[i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cslot = 0;
  u32x4 af[4];
#pragma unroll
  for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const u32x4*>(ring.slot(0) + lane * 16 + i * 1024);
  f32x16 acc0[PT], acc1[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) acc0[pt] = load_bias(lds_zero, 0, h);
  char* const xp = smem + BWD16_TAIL_BYTES + 3 * BWD16_RING_SLOT + wave * (PT * XPOSE_WAVE_BYTES);
  const unsigned xp_w = (unsigned)(j * XPOSE_PITCH + 4 * h) * 4u;
  const unsigned xp_r = (unsigned)((lane >> 3) * XPOSE_PITCH + 4 * (lane & 7)) * 4u;
  const unsigned g_off = (unsigned)((lane >> 3) * 256 + 4 * (lane & 7)) * 4u;
  const unsigned xp16_w = (unsigned)(j * XP16_PITCH + 8 * h);                        // incoming activation tile (softplus values)
  const unsigned xp16_r = (unsigned)((lane >> 2) * XP16_PITCH + 16 * (lane & 3));
  const unsigned xs16_w = (unsigned)(j * XS16_PITCH + 8 * h);                        // outgoing gradient tile PAIRS (sn_mlp_bf16.h)
  const unsigned xs16_lds = (unsigned)(BWD16_TAIL_BYTES + 3 * BWD16_RING_SLOT + wave * (PT * XPOSE_WAVE_BYTES)) + xs16_w;   // ... as an LDS byte address
  const unsigned xs16_r = (unsigned)((lane >> 3) * XS16_PITCH + 16 * (lane & 7));
  const unsigned g16_off = (unsigned)((lane >> 3) * 512 + 16 * (lane & 7));
  char* const xl = smem + BWD16_TAIL_BYTES + 3 * BWD16_RING_SLOT + 4 * PT * XPOSE_WAVE_BYTES + wave * (PT * XP16_WAVE_BYTES);

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p_wave = (tile * 4 + __builtin_amdgcn_readfirstlane(wave)) * (PT * 32);
    long p[PT];
    float gy3[PT][3], gsig[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const long p_raw = p_wave + pt * 32 + j;
      const bool valid = p_raw < P;
      p[pt] = valid ? p_raw : P - 1;
      const float4 g = reinterpret_cast<const float4*>(g_raw)[p[pt]];
      const float4 o = reinterpret_cast<const float4*>(out_raw)[p[pt]];
      const float k = 0.5f * 1.002f * 0.5f;                                // d/dy WidenedSigmoid = .2505 (1 - t^2)
      const float tx = (2.0f * o.x - 1.0f) * (1.0f / 1.002f), ty = (2.0f * o.y - 1.0f) * (1.0f / 1.002f),
                  tz = (2.0f * o.z - 1.0f) * (1.0f / 1.002f);
      if (SN_NEWACT) {
        gy3[pt][0] = valid ? g.x * k * (1.0f - tx * tx) : 0.0f;
        gy3[pt][1] = valid ? g.y * k * (1.0f - ty * ty) : 0.0f;
        gy3[pt][2] = valid ? g.z * k * (1.0f - tz * tz) : 0.0f;
      } else {                                   // Sigmoid (nerf.py:100): s (1 - s)
        gy3[pt][0] = valid ? g.x * o.x * (1.0f - o.x) : 0.0f;
        gy3[pt][1] = valid ? g.y * o.y * (1.0f - o.y) : 0.0f;
        gy3[pt][2] = valid ? g.z * o.z * (1.0f - o.z) : 0.0f;
      }
      gsig[pt] = valid ? g.w : 0.0f;
      if (valid && h == 0) {
        float4 gy;
        gy.x = gy3[pt][0]; gy.y = gy3[pt][1]; gy.z = gy3[pt][2]; gy.w = gsig[pt];
        reinterpret_cast<float4*>(g_out)[p_raw] = gy;                     // g_y of rgb.0 (3) and of sigma (1)
        // the same 4 values as a zero-padded 32-wide block in the unused half of slot 9 (columns 128..159): the A operand
        // of the rgb / sigma weight-gradient contractions (sn_dw.hip variants 4/5)
        if (S16) {
          uint4* row = reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(G) + ((long)9 * slot_rows + p_raw) * 256 + 128);
          row[0] = make_uint4(pack2(gy.x, gy.y), pack2(gy.z, gy.w), 0u, 0u);
#pragma unroll
          for (int q = 1; q < 4; ++q) row[q] = make_uint4(0u, 0u, 0u, 0u);
        } else {
          float4* row = reinterpret_cast<float4*>(G + ((long)9 * slot_rows + p_raw) * 256 + 128);
          row[0] = gy;
#pragma unroll
          for (int q = 1; q < 8; ++q) row[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
      }
    }

    // forward activation tile of both point tiles in the accumulator layout (4 x 16 B per lane and point tile), requested
    // one slab ahead of the epilogue that needs it
    f32x4 av[PT][4];                                                        // fp32 state (and the softplus tile)
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t au[PT][4];                                                      // bf16 state: the packed pairs, accumulator layout
    f32x4 ald[PT][2];                                                       // ... as loaded (row chunks), before the LDS turn
    // rows -> accumulator layout through the wave's second staging tile; run right before the epilogue that needs au / av
    auto act_turn = [&](bool values) __attribute__((always_inline)) {
      if (S16) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
          for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(xl + pt * XP16_WAVE_BYTES + xp16_r + 16 * i * XP16_PITCH) = ald[pt][i];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const u32x2_t u = *reinterpret_cast<const u32x2_t*>(xl + pt * XP16_WAVE_BYTES + xp16_w + 16 * q4);
            au[pt][q4] = u;
            if (values) {                                                   // softplus derivative needs the values
              av[pt][q4][0] = __builtin_bit_cast(float, u[0] << 16); av[pt][q4][1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
              av[pt][q4][2] = __builtin_bit_cast(float, u[1] << 16); av[pt][q4][3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
            }
          }
        }
      }
    };
    uint32_t c01 = 0x00010001u;
    asm volatile("" : "+v"(c01));
    uint32_t sign_word = 0;                                                 // bf16 state: ReLU sign word of the tile in flight
    // load operation k of the activation tile (slot, t): S16 ReLU masks: k = 0 is the tile's sign word (one dword per lane);
    // S16 values (softplus tile): k = 2 pt + i, two row-coalesced loads per point tile; fp32 state: k = 4 pt + q4, the
    // accumulator layout directly (4 x 16 B per lane and point tile)
    auto load_act_op = [&](int slot, int t, bool values, int k) __attribute__((always_inline)) {
      if (S16 && !values) {
        if (k == 0) {
          const char* src = reinterpret_cast<const char*>(acts) + (((long)9 * slot_rows + p_wave + 8 * slot + t) * 256 + 128) * 2;
#if SN_BF16_COUNTED
          // inline asm: the WAIT is ours (sign_wait) -- hipcc treats loads and stores in flight as unordered and would drain the
          // row stores issued behind this load with vmcnt(0) (s_nop 4: SALU write of the base -> its use as a VMEM address)
          unsigned so = (unsigned)lane * 4u;
          asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 nt" : "=v"(sign_word) : "v"(so), "s"(src) : "memory");
#else
          sign_word = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(src + lane * 4));
#endif
        }
      } else if (S16) {
        // row-coalesced: lane -> (row lane>>2 [+16], 16-byte chunk lane&3) of the 32-point x 64-byte tile; 2 loads per
        // point tile instead of 4 scattered 8-byte ones (64 cache lines per instruction: the TA, not HBM, was the limit)
        if (k < 2 * PT) {
          const int pt = k >> 1, i = k & 1;
          const long row = p_wave + pt * 32 + 16 * i + (lane >> 2);
          const long rc = row < P ? row : P - 1;
          const char* src = reinterpret_cast<const char*>(acts) + (((long)slot * slot_rows + rc) * 256 + 32 * t) * 2 + 16 * (lane & 3);
          ald[pt][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
        }
      } else if (k < 4 * PT) {
        const int pt = k >> 2, q4 = k & 3;
        const float* src = acts + ((long)slot * slot_rows + p[pt]) * 256 + 32 * t + 4 * h;
        av[pt][q4] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 8 * q4));
      }
    };
    auto load_act = [&](int slot, int t, bool values = false) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 4 * PT; ++k) load_act_op(slot, t, values, k);
    };
    auto stage = [&](int pt, int t, int qq, const float (&v)[4], uint32_t t0, uint32_t t1) __attribute__((always_inline)) {
      if (S16) {
        lds_write_b64(xs16_lds + pt * XPOSE_WAVE_BYTES, 64 * (t & 1) + 16 * qq, t0, t1);       // (inline asm: sn_mlp_bf16.h)
      } else {
        f32x4 o;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
        *reinterpret_cast<f32x4*>(xp + pt * XPOSE_WAVE_BYTES + xp_w + 32 * qq) = o;
      }
    };
    // row-group store k = 4 pt + i of finished tile t (rows >= P receive the zeros their lanes hold)
    auto store_op = [&](int slot, int t, int k) __attribute__((always_inline)) {
      const int pt = k >> 2, i = k & 3;
      if (S16) {
        if (t & 1) {                             // tiles t-1, t: whole 128-byte rows
          const f32x4 o = *reinterpret_cast<const f32x4*>(xp + pt * XPOSE_WAVE_BYTES + xs16_r + 8 * i * XS16_PITCH);
          char* base = reinterpret_cast<char*>(G) + (((long)slot * slot_rows + p_wave + pt * 32 + 8 * i) * 256 + 32 * (t - 1)) * 2;
          unsigned go = g16_off;
          asm volatile("" : "+v"(go));
          __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(base + go));
        }
      } else {
        const f32x4 o = *reinterpret_cast<const f32x4*>(xp + pt * XPOSE_WAVE_BYTES + xp_r + 8 * i * XPOSE_PITCH * 4);
        char* base = reinterpret_cast<char*>(G) + (((long)slot * slot_rows + p_wave + pt * 32 + 8 * i) * 256 + 32 * t) * 4;
        unsigned go = g_off;
        asm volatile("" : "+v"(go));
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(base + go));
      }
    };
    auto store_tile = [&](int slot, int t) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 4 * PT; ++k) store_op(slot, t, k);
    };
    // memory step `step` of `n` of a slab (sn_mlp_bf16.h): the load operations of the activation tile t_next of slot mslot
    // (if any) FIRST -- a bf16 slab is only ~1 us long and the next slab's epilogue consumes them (requested last they cost
    // 7 % of the launch) -- then the 4 PT row stores of the finished tile t_done (if any), dealt evenly over the n calls
    constexpr int N_LD_OPS = S16 ? 1 : 4 * PT;
    constexpr int N_MEM_OPS = 4 * PT + N_LD_OPS;
    auto mem_step = [&](int oslot, int t_done, int mslot, int t_next, bool mask, int step, int n) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < N_MEM_OPS; ++k)
        if (k >= step * N_MEM_OPS / n && k < (step + 1) * N_MEM_OPS / n) {
          if (k < N_LD_OPS) { if (mask) load_act_op(mslot, t_next, false, k); }
          else if (t_done >= 0) store_op(oslot, t_done, k - N_LD_OPS);
        }
    };

    // ---- rgb.0^T on the VALU: g_h2 = W_r^T g_y3 ; g_y2 = g_h2 (1 - exp(-h2)); written to set 0 (K-slots 16t + r)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      load_act(9, t, true);
      act_turn(true);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 2 * q + i;
            const float w0 = lds_aux[snl::BB_AUX_RGBT + 0 * 128 + h * 64 + 16 * t + r];
            const float w1 = lds_aux[snl::BB_AUX_RGBT + 1 * 128 + h * 64 + 16 * t + r];
            const float w2 = lds_aux[snl::BB_AUX_RGBT + 2 * 128 + h * 64 + 16 * t + r];
            const float gh = __builtin_fmaf(w2, gy3[pt][2], __builtin_fmaf(w1, gy3[pt][1], w0 * gy3[pt][0]));
            v[i] = SN_NEWACT ? gh * (1.0f - __expf(-av[pt][r >> 2][r & 3])) : (av[pt][r >> 2][r & 3] > 0.0f ? gh : 0.0f);   // ReLU (nerf.py:94)
          }
          uint32_t t0, t1;
          epi_copy(act_reg(0, 2 * t + (q >> 2), pt) + (q & 3), v[0], v[1], v[2], v[3], t0, t1);
          stage(pt, t, q >> 1, v, t0, t1);
        }
      store_tile(9, t);
    }

    int s = 0;
    int mask_slot = 0;                                                     // acts slot of the ReLU mask of the running layer
    int out_slot = 0;                                                      // G slot the running layer writes
    auto copy_tile = [&](auto wset, int t, const f32x16 (&r)[PT]) __attribute__((always_inline)) {   // g_final: no activation
      constexpr int W = decltype(wset)::value;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          uint32_t t0, t1;
          epi_copy(act_reg(W, 2 * t + (q >> 2), pt) + (q & 3), r[pt][2 * q], r[pt][2 * q + 1], r[pt][2 * q + 2], r[pt][2 * q + 3], t0, t1);
          const float v[4] = {r[pt][2 * q], r[pt][2 * q + 1], r[pt][2 * q + 2], r[pt][2 * q + 3]};
          stage(pt, t, q >> 1, v, t0, t1);
        }
    };
    // g_y = g_h [h > 0]; with_sigma: g_h8 also gets the sigma head's term  sigma.weight[f] g_sigma  (nerf.py:136)
    auto mask_tile_impl = [&](auto wset, auto with_sigma, int t, const f32x16 (&r)[PT]) __attribute__((always_inline)) {
      constexpr int W = decltype(wset)::value;
      constexpr bool SIG = decltype(with_sigma)::value;
      if (!S16) act_turn(false);
#if SN_BF16_COUNTED
      // the sign word of tile t was requested in slab t; behind it only the eight row stores of tile t-1 (odd t-1) were issued
      if (S16) {
        if (t >= 2 && (t & 1) == 0) asm volatile("s_waitcnt vmcnt(8)" : "+v"(sign_word) :: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(sign_word) :: "memory");
      }
#endif
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          float x[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            x[i] = r[pt][2 * q + i];
            if (SIG) x[i] = __builtin_fmaf(lds_aux[snl::BB_AUX_SIGT + h * 128 + 16 * t + 2 * q + i], gsig[pt], x[i]);
          }
          float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          uint32_t t0, t1;
          if (S16) {
            epi_maskbits(act_reg(W, 2 * t + (q >> 2), pt) + (q & 3), SIG ? 2 * q + 2 * pt : 8 * pt + q, x[0], x[1], x[2], x[3], sign_word, c01, t0, t1);
          } else {
            const f32x4 a = av[pt][q >> 1];
            epi_mask(act_reg(W, 2 * t + (q >> 2), pt) + (q & 3), x[0], x[1], x[2], x[3], a[0], a[1], a[2], a[3], v, t0, t1);
          }
          stage(pt, t, q >> 1, v, t0, t1);
        }
    };
    auto mask_tile = [&](auto wset, int t, const f32x16 (&r)[PT]) __attribute__((always_inline)) {
      mask_tile_impl(wset, std::false_type{}, t, r);
    };
    auto mask_sigma_tile = [&](auto wset, int t, const f32x16 (&r)[PT]) __attribute__((always_inline)) {
      mask_tile_impl(wset, std::true_type{}, t, r);
    };
#define SNC_LW_CUR (ring.slot(cslot) + lane * 16)
#define SNC_LW_NEXT (ring.slot(cslot == 2 ? 0 : cslot + 1) + lane * 16)
#define SNC_SNEXT (s + 1 == snl::NBB_SLABS ? 0 : s + 1)
#define SNC_W(W_) std::integral_constant<int, W_>{}
    // slab of output tile T_ (literal).  The deferred epilogue of tile T_-1 runs behind the first MFMA pair; its row stores and
    // the loads of the activation tile THIS slab's epilogue needs (MASK_) are the slab's memory steps (behind the sync point
    // and the DMA pieces, one per k-step: sn_mlp_bf16.h).
#define SNC_SLAB(T_, NK_, SET_, NB_, EPI_, W_, MASK_)                                                              \
  do {                                                                                                             \
    /* bf16 state: the eight row stores of an odd finished tile sit behind the DMA pieces (and the sign-word load,  \
       already waited for) of their slab: the next sync point leaves them in flight (sn_mlp_bf16.h VMW) */            \
    constexpr int VW_ = (S16 && SN_BF16_COUNTED && (NK_) >= 16 && ((T_) == 0 || (((T_) & 1) && (T_) >= 3))) ? 8 : 0; \
    if (((T_) & 1) == 0)                                                                                           \
      slab_bf16<NK_, 0, SET_, SET_, 2, 0, NB_, VW_>(acc0, acc1, af, SNC_LW_CUR, static_cast<const u32x4*>(nullptr), SNC_LW_NEXT, lds_zero, SNC_SNEXT, h, \
          ring, [&]() __attribute__((always_inline)) { if ((T_) > 0) EPI_(SNC_W(W_), (T_) - 1, acc1); },           \
          [&](int st, int n) __attribute__((always_inline)) { mem_step(out_slot, (T_) - 1, mask_slot, T_, MASK_, st, n); }); \
    else                                                                                                           \
      slab_bf16<NK_, 0, SET_, SET_, 2, 0, NB_, VW_>(acc1, acc0, af, SNC_LW_CUR, static_cast<const u32x4*>(nullptr), SNC_LW_NEXT, lds_zero, SNC_SNEXT, h, \
          ring, [&]() __attribute__((always_inline)) { EPI_(SNC_W(W_), (T_) - 1, acc0); },                         \
          [&](int st, int n) __attribute__((always_inline)) { mem_step(out_slot, (T_) - 1, mask_slot, T_, MASK_, st, n); }); \
    ++s; cslot = (cslot == 2) ? 0 : cslot + 1;                                                                     \
  } while (0)
#define SNC_LAYER(NK_, SET_, NBA_, NBB_, EPI_, W_, MASK_)   \
  do {                                                      \
    SNC_SLAB(0, NK_, SET_, NBA_, EPI_, W_, MASK_);          \
    SNC_SLAB(1, NK_, SET_, NBA_, EPI_, W_, MASK_);          \
    SNC_SLAB(2, NK_, SET_, NBA_, EPI_, W_, MASK_);          \
    SNC_SLAB(3, NK_, SET_, NBA_, EPI_, W_, MASK_);          \
    SNC_SLAB(4, NK_, SET_, NBA_, EPI_, W_, MASK_);          \
    SNC_SLAB(5, NK_, SET_, NBA_, EPI_, W_, MASK_);          \
    SNC_SLAB(6, NK_, SET_, NBB_, EPI_, W_, MASK_);          \
    SNC_SLAB(7, NK_, SET_, NBB_, EPI_, W_, MASK_);          \
    mfma_result_fence();                                    \
    EPI_(SNC_W(W_), 7, acc1);                               \
    store_tile(out_slot, 7);                                \
  } while (0)

    // ---- dir_encoding.0^T (first 256 inputs): g_final = W_d[:, :256]^T g_y2; reads set 0 (8 k-steps), writes set 1
    out_slot = 8;
    SNC_LAYER(8, 0, B_D, B_H, copy_tile, 1, false);
    // ---- xyz_encoding_final^T (+ sigma^T on the VALU): g_y8 = (W_f^T g_final + w_sigma g_sigma) [h8 > 0]; set 1 -> set 0
    mask_slot = 7; out_slot = 7;
    SNC_LAYER(16, 1, B_H, B_H, mask_sigma_tile, 0, true);
    // ---- xyz_encoding_{li+1}^T, li = 7..1: g_y_{li-1} = (W^T g_y_li) [h_li > 0]; odd li reads set 0 and writes set 1
#pragma unroll 1
    for (int li = 7; li >= 1; --li) {
      mask_slot = li - 1; out_slot = li - 1;
      if (li == 1) SNC_LAYER(16, 0, B_H, B_D, mask_tile, 1, true);          // tiles 6,7 stage the next point tile's DIRT slabs
      else if (li & 1) SNC_LAYER(16, 0, B_H, B_H, mask_tile, 1, true);
      else SNC_LAYER(16, 1, B_H, B_H, mask_tile, 0, true);
    }
#undef SNC_LW_CUR
#undef SNC_LW_NEXT
#undef SNC_SNEXT
#undef SNC_W
#undef SNC_SLAB
#undef SNC_LAYER
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace snk

extern "C" int SN_LAUNCH_NAME(sn_mlp_backward_chain_bf16)(const void* bblob, const float* acts, const float* out_raw,
                                                 const float* g_raw, long n_points, long slot_rows, float* G,
                                                 float* g_out, int state_bf16, hipStream_t stream) {
  using namespace snk;
  if (n_points <= 0) return 0;
  const long tiles = (n_points + 255) / 256;
  if (slot_rows < tiles * 256) return -1;
  const int n_cu = snh::cu_count();
#define SN_LAUNCH(S16_)                                                                                            \
  do {                                                                                                             \
    auto kfn = mlp_bwd_chain_bf16_kernel<S16_>;                                                                    \
    SN_ENSURE_DYN_LDS(kfn, BWD16_LDS_BYTES);                                                                       \
    hipLaunchKernelGGL(kfn, dim3((unsigned)(tiles < n_cu ? tiles : n_cu)), dim3(256), BWD16_LDS_BYTES, stream,     \
                       reinterpret_cast<const char*>(bblob), acts, out_raw, g_raw, n_points, slot_rows, G, g_out); \
  } while (0)
  if (state_bf16) SN_LAUNCH(true); else SN_LAUNCH(false);
#undef SN_LAUNCH
  return (int)hipGetLastError();
}
