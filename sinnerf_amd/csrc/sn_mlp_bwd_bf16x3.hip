// sn_mlp_bwd_bf16x3.hip -- backward "chain" of the fused NeRF MLP at FP32-LEVEL accuracy on the bf16 matrix cores
// (SN_DTYPE_BF16X3 of sn_mlp_backward_chain): g_x = W^T g_y, g_y = g_h (.) act'(.) through every layer -- what torch autograd
// derives from models/nerf.py:122-148 (+ models/activations.py) -- with the 3-term split of sn_mlp_x3.h:
//     W^T g ~= Wh^T gh + Wl^T gh + Wh^T gl      (hi / lo bf16 pairs of the transposed weights and of the gradient, fp32 accumulate)
// Same data as the fp32 chain (sn_mlp_bwd.hip), except that the ReLU masks [h > 0] come from the SIGN WORDS the bf16x3 training
// forward left in the unused half of acts slot 9 (256 B per point and layer instead of 1 KB of activations; sn_mlp_x3.h
// x3_sign_bits): it READS those, the dir_encoding outputs (ShiftedSoftplus derivative 1 - exp(-h2)) and the forward output
// (WidenedSigmoid derivative) -- so `acts` must be the array sn_mlp_forward_train(SN_DTYPE_BF16X3) wrote -- and WRITES the fp32 pre-activation gradients
// G[10][slot_rows][256] (+ the 4-wide head block in slot 9) that the weight-gradient contractions consume -- the fp32 training
// state, value for value at fp32 rounding level.  Same machinery as the bf16x3 forward: one 32-point tile per wave, two
// accumulator chains, (hi, lo) activation sets in the hand-managed AGPR file, transposed (hi, lo) weight slabs of K x 128 B
// through a 3-slot LDS ring (csrc/sn_layout.h "Backward-chain blob, bf16x3").
//
// Memory operations of a slab (its time is 48 MFMAs x 32 cycles: no drain per slab is affordable): the four row-group stores of the
// previous tile in its last four k-steps, behind the DMA pieces; the sync point waits for the pieces with those stores in flight
// (COUNTED vmcnt, fence-less barrier); the four sign words of a layer are ONE 16-byte inline-asm load per lane issued a whole
// layer ahead.  Staging writes are inline asm (sn_mlp_bf16.h).  (First version: the fp32 activation tiles as masks, four scattered
// 16-byte loads per lane and slab one slab ahead -- 3.26 ms per 524 288 points against the forward's 1.99: latency- and
// address-path-bound.)
#include "sn_mlp_x3.h"

namespace snk {

constexpr int BX3_SLOT = 256 * 128;                                        // widest slab: K = 256
constexpr int BX3_TAIL_BYTES = snl::BB_TAIL_FLOATS * 4;                    // 11776: zero "bias" slots + aux table (the bf16 chain's tail)
constexpr int BX3_LDS_BYTES = BX3_TAIL_BYTES + 3 * BX3_SLOT + XPOSE_LDS_BYTES;     // 128512
typedef RingT<128, BX3_SLOT> RingX3B;

// the 16-byte load of a layer's sign words as inline asm: no compiler wait (hipcc treats loads and stores in flight as unordered and
// would drain the row stores issued behind this load with vmcnt(0)); it is covered by the counted waits of the eight sync points
// that follow before its first use.  s_nop 4: SALU write of the base -> its use as a VMEM address
SN_DEV void x3_load_b128(u32x4& dst, unsigned voff, const char* base) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}

__global__ void __launch_bounds__(256)
mlp_bwd_chain_bf16x3_kernel(const char* __restrict__ bblob, const float* __restrict__ acts, const float* __restrict__ out_raw,
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
  constexpr int TILE_PTS = 4 * 32;
  const long n_tiles = (P + TILE_PTS - 1) / TILE_PTS;

  RingX3B ring;
  ring.blob = bblob;
  ring.gnext = bblob;
  ring.base = smem + BX3_TAIL_BYTES;
  ring.n_used = snl::NBB_SLABS;
  ring.stage_id = 0;
  ring.stage_slot = 0;
  ring.remaining = 0;
  ring.tid = tid;
  ring.wbase = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
  ring.pieces = 0; ring.piece = 0; ring.slab_bytes = 0;
  constexpr int B_D = 128 * 128, B_H = 256 * 128;                          // slab bytes: DIRT, FINT / LT
#pragma unroll
  for (int i = 0; i < 2; ++i) {                                            // slabs 0, 1 (both DIRT)
    ring.begin_static();
    ring.piece_static(); ring.piece_static(); ring.piece_static(); ring.piece_static();
    ring.template end_static_bytes<B_D>();
  }
  {
    const float4* gb = reinterpret_cast<const float4*>(bblob + snl::bbx_tail_byte_offset());
    float4* lb = reinterpret_cast<float4*>(lds_zero);
    for (int i = tid; i < snl::BB_TAIL_FLOATS / 4; i += 256) lb[i] = gb[i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cslot = 0;
  u32x4 af[4][2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    af[i][0] = *reinterpret_cast<const u32x4*>(ring.slot(0) + lane * 16 + i * 2048);
    af[i][1] = *reinterpret_cast<const u32x4*>(ring.slot(0) + lane * 16 + i * 2048 + 1024);
  }
  f32x16 a0, b0, a1, b1;
  a0 = load_bias(lds_zero, 0, h);
  const char* const xp = smem + BX3_TAIL_BYTES + 3 * BX3_SLOT + wave * XPOSE_WAVE_BYTES;
  const unsigned xp_w_lds = (unsigned)(BX3_TAIL_BYTES + 3 * BX3_SLOT + wave * XPOSE_WAVE_BYTES) + (unsigned)(j * XPOSE_PITCH + 4 * h) * 4u;
  const unsigned xp_s_lds = xp_w_lds - 8u * (unsigned)h;           // split-state tiles (slots 0..8): 8 B of hi parts per lane, lo parts 16 B on
  const unsigned xp_r = (unsigned)((lane >> 3) * XPOSE_PITCH + 4 * (lane & 7)) * 4u;
  const unsigned g_off = (unsigned)((lane >> 3) * 256 + 4 * (lane & 7)) * 4u;
  const unsigned s_off = (unsigned)((lane >> 4) * 1024 + (lane & 15) * 16);    // this lane's sign words inside a layer's four rows

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p_wave = (tile * 4 + __builtin_amdgcn_readfirstlane(wave)) * 32;
    const long p_raw = p_wave + j;
    const bool valid = p_raw < P;
    const long p = valid ? p_raw : P - 1;
    float gy3[3], gsig;
    {
      const float4 g = reinterpret_cast<const float4*>(g_raw)[p];
      const float4 o = reinterpret_cast<const float4*>(out_raw)[p];
      const float k = 0.5f * 1.002f * 0.5f;                                // d/dy WidenedSigmoid = .2505 (1 - t^2)
      const float tx = (2.0f * o.x - 1.0f) * (1.0f / 1.002f), ty = (2.0f * o.y - 1.0f) * (1.0f / 1.002f),
                  tz = (2.0f * o.z - 1.0f) * (1.0f / 1.002f);
      if (SN_NEWACT) {
        gy3[0] = valid ? g.x * k * (1.0f - tx * tx) : 0.0f;
        gy3[1] = valid ? g.y * k * (1.0f - ty * ty) : 0.0f;
        gy3[2] = valid ? g.z * k * (1.0f - tz * tz) : 0.0f;
      } else {                                   // Sigmoid (nerf.py:100): s (1 - s)
        gy3[0] = valid ? g.x * o.x * (1.0f - o.x) : 0.0f;
        gy3[1] = valid ? g.y * o.y * (1.0f - o.y) : 0.0f;
        gy3[2] = valid ? g.z * o.z * (1.0f - o.z) : 0.0f;
      }
      gsig = valid ? g.w : 0.0f;
      if (valid && h == 0) {
        float4 gy;
        gy.x = gy3[0]; gy.y = gy3[1]; gy.z = gy3[2]; gy.w = gsig;
        reinterpret_cast<float4*>(g_out)[p_raw] = gy;                     // g_y of rgb.0 (3) and of sigma (1)
        // the same 4 values as a zero-padded 32-wide block in the unused half of slot 9 (columns 128..159): the A operand of the
        // rgb / sigma weight-gradient contractions (sn_dw.hip)
        float4* row = reinterpret_cast<float4*>(G + ((long)9 * slot_rows + p_raw) * 256 + 128);
        row[0] = gy;
#pragma unroll
        for (int q = 1; q < 8; ++q) row[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }

    // ReLU sign words (sn_mlp_fwd_bf16x3.hip): layer l (acts slot l) = rows p_wave + 4 l .. + 3 of slot 9, bytes 512.. of each row
    u32x4 sw, swn;                                                         // the running layer's four words; the next layer's, in flight
    auto sign_base = [&](int slot) __attribute__((always_inline)) {
      return reinterpret_cast<const char*>(acts) + (((long)9 * slot_rows + p_wave + 4 * slot) * 256 + 128) * 4;
    };
    sw = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sign_base(7) + s_off));      // xyz_encoding_8's (the first masked layer)
    swn = sw;
    auto stage = [&](int qq, const float (&v)[4]) __attribute__((always_inline)) { x3_lds_write_b128(xp_w_lds, 32 * qq, v); };   // fp32 (slot 9)
    // slots 0..8: the (hi, lo) pairs the epilogue builds for the next transposed layer ARE the stored gradient state
    auto stage_split = [&](int qq, uint32_t h0, uint32_t h1, uint32_t l0, uint32_t l1) __attribute__((always_inline)) {
      x3_lds_write_split(xp_s_lds, 32 * qq, h0, h1, l0, l1);
    };
    f32x4 rowbuf[1];                                         // row groups between their ds_read and their store (x3_store_step)
    auto rows_read = [&](int i) __attribute__((always_inline)) {
      rowbuf[0] = *reinterpret_cast<const f32x4*>(xp + xp_r + 8 * i * XPOSE_PITCH * 4);
    };
    auto rows_write = [&](int slot, int t, int i) __attribute__((always_inline)) {
      const char* base = reinterpret_cast<const char*>(G) + (((long)slot * slot_rows + p_wave + 8 * i) * 256 + 32 * t) * 4;
      unsigned go = g_off;
      asm volatile("" : "+v"(go));
      __builtin_nontemporal_store(rowbuf[0], reinterpret_cast<f32x4*>(const_cast<char*>(base) + go));
    };
    auto store_rows = [&](int slot, int t, int i) __attribute__((always_inline)) { rows_read(i); rows_write(slot, t, i); };

    // ---- rgb.0^T on the VALU: g_h2 = W_r^T g_y3 ; g_y2 = g_h2 (1 - exp(-h2)); written to set 0 (K-slots 16t + r).  All four
    //      softplus tiles are requested first (one latency, not four)
    {
      f32x4 h2[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* src = acts + ((long)9 * slot_rows + p) * 256 + 32 * t + 8 * i + 4 * h;
          h2[t][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
        }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 2 * q + i;
            const float w0 = lds_aux[snl::BB_AUX_RGBT + 0 * 128 + h * 64 + 16 * t + r];
            const float w1 = lds_aux[snl::BB_AUX_RGBT + 1 * 128 + h * 64 + 16 * t + r];
            const float w2 = lds_aux[snl::BB_AUX_RGBT + 2 * 128 + h * 64 + 16 * t + r];
            const float gh = __builtin_fmaf(w2, gy3[2], __builtin_fmaf(w1, gy3[1], w0 * gy3[0]));
            v[i] = SN_NEWACT ? gh * (1.0f - expf(-h2[t][r >> 2][r & 3])) : (h2[t][r >> 2][r & 3] > 0.0f ? gh : 0.0f);   // ReLU (nerf.py:94)
          }
          x3_put(x3_reg(0, 0, 2 * t + (q >> 2)) + (q & 3), x3_reg(0, 1, 2 * t + (q >> 2)) + (q & 3), v);
          stage(q >> 1, v);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) store_rows(9, t, i);
      }
    }

    int s = 0;
    int out_slot = 0;                                                      // G slot the running layer writes
    // (blk = 0..3: the block of accumulator registers 4 blk .. 4 blk + 3, q = 2 blk -- one block behind each of a slab's first k-steps)
    auto copy_tile = [&](auto wset, int t, const f32x16& ra, const f32x16& rb, int blk) __attribute__((always_inline)) {   // g_final: no activation
      constexpr int W = decltype(wset)::value;
      {
        const int q = 2 * blk;
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = ra[2 * q + i] + rb[2 * q + i];
        uint32_t h0, h1, l0, l1;
        x3_put(x3_reg(W, 0, 2 * t + (q >> 2)) + (q & 3), x3_reg(W, 1, 2 * t + (q >> 2)) + (q & 3), x, h0, h1, l0, l1);
        stage_split(q >> 1, h0, h1, l0, l1);
      }
    };
    // g_y = g_h [h > 0]; with_sigma: g_h8 also gets the sigma head's term  sigma.weight[f] g_sigma  (nerf.py:136)
    auto mask_tile_impl = [&](auto wset, auto with_sigma, int t, const f32x16& ra, const f32x16& rb, int blk) __attribute__((always_inline)) {
      constexpr int W = decltype(wset)::value;
      constexpr bool SIG = decltype(with_sigma)::value;
      const uint32_t word = sw[t >> 1];
      {
        const int q = 2 * blk;
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[i] = ra[2 * q + i] + rb[2 * q + i];
          if (SIG) x[i] = __builtin_fmaf(lds_aux[snl::BB_AUX_SIGT + h * 128 + 16 * t + 2 * q + i], gsig, x[i]);
        }
        float v[4];
        uint32_t h0, h1, l0, l1;
        x3_put_signed(x3_reg(W, 0, 2 * t + (q >> 2)) + (q & 3), x3_reg(W, 1, 2 * t + (q >> 2)) + (q & 3), x, word, q + 8 * (t & 1), v, h0, h1, l0, l1);
        stage_split(q >> 1, h0, h1, l0, l1);
      }
    };
    auto mask_tile = [&](auto wset, int t, const f32x16& ra, const f32x16& rb, int blk) __attribute__((always_inline)) {
      mask_tile_impl(wset, std::false_type{}, t, ra, rb, blk);
    };
    auto mask_sigma_tile = [&](auto wset, int t, const f32x16& ra, const f32x16& rb, int blk) __attribute__((always_inline)) {
      mask_tile_impl(wset, std::true_type{}, t, ra, rb, blk);
    };
#define SNY_LW_CUR (ring.slot(cslot) + lane * 16)
#define SNY_LW_NEXT (ring.slot(cslot == 2 ? 0 : cslot + 1) + lane * 16)
#define SNY_SNEXT (s + 1 == snl::NBB_SLABS ? 0 : s + 1)
#define SNY_W(W_) std::integral_constant<int, W_>{}
    // Slab of output tile T_ (literal), staging NB_ bytes (the slab two ahead).  The youngest four vector-memory operations of a wave
    // at the sync point are the row stores the previous slab posted behind its DMA pieces (T_ = 1: its predecessor posts none; T_ = 0:
    // the previous layer's last eight stores, and possibly the next layer's sign-word load behind them)
#define SNY_SLAB(T_, NK_, SET_, NB_, EPI_, W_)                                                                     \
  do {                                                                                                             \
    constexpr int VW_ = ((T_) != 1) ? 4 : 0;                                                                       \
    if (((T_) & 1) == 0)                                                                                           \
      slab_x3<NK_, 0, SET_, SET_, 2, 0, NB_, VW_>(a0, b0, a1, af, SNY_LW_CUR, static_cast<const u32x4*>(nullptr), static_cast<const u32x4*>(nullptr), \
          SNY_LW_NEXT, lds_zero, SNY_SNEXT, h, ring,                                                               \
          [&](int blk) __attribute__((always_inline)) { if ((T_) > 0) EPI_(SNY_W(W_), (T_) - 1, a1, b1, blk); },   \
          [&](int ks, int nk, int st0, bool before) __attribute__((always_inline)) {                                         \
            if (!before && (T_) > 0) x3_store_step(ks, nk, st0, rows_read, [&](int i) __attribute__((always_inline)) { rows_write(out_slot, (T_) - 1, i); }); }); \
    else                                                                                                           \
      slab_x3<NK_, 0, SET_, SET_, 2, 0, NB_, VW_>(a1, b1, a0, af, SNY_LW_CUR, static_cast<const u32x4*>(nullptr), static_cast<const u32x4*>(nullptr), \
          SNY_LW_NEXT, lds_zero, SNY_SNEXT, h, ring,                                                               \
          [&](int blk) __attribute__((always_inline)) { EPI_(SNY_W(W_), (T_) - 1, a0, b0, blk); },                 \
          [&](int ks, int nk, int st0, bool before) __attribute__((always_inline)) {                                         \
            if (!before) x3_store_step(ks, nk, st0, rows_read, [&](int i) __attribute__((always_inline)) { rows_write(out_slot, (T_) - 1, i); }); }); \
    ++s; cslot = (cslot == 2) ? 0 : cslot + 1;                                                                     \
  } while (0)
    // the 8 output tiles of a transposed layer; tiles 6, 7 stage the NEXT layer's slabs (NBB_); the last tile's epilogue is not deferred
#define SNY_LAYER(NK_, SET_, NBA_, NBB_, EPI_, W_)                              \
  do {                                                                          \
    SNY_SLAB(0, NK_, SET_, NBA_, EPI_, W_);                                     \
    SNY_SLAB(1, NK_, SET_, NBA_, EPI_, W_);                                     \
    SNY_SLAB(2, NK_, SET_, NBA_, EPI_, W_);                                     \
    SNY_SLAB(3, NK_, SET_, NBA_, EPI_, W_);                                     \
    SNY_SLAB(4, NK_, SET_, NBA_, EPI_, W_);                                     \
    SNY_SLAB(5, NK_, SET_, NBA_, EPI_, W_);                                     \
    SNY_SLAB(6, NK_, SET_, NBB_, EPI_, W_);                                     \
    SNY_SLAB(7, NK_, SET_, NBB_, EPI_, W_);                                     \
    x3_result_fence(a1, b1);                                                    \
    EPI_(SNY_W(W_), 7, a1, b1, 0); EPI_(SNY_W(W_), 7, a1, b1, 1);               \
    EPI_(SNY_W(W_), 7, a1, b1, 2); EPI_(SNY_W(W_), 7, a1, b1, 3);               \
    store_rows(out_slot, 7, 0); store_rows(out_slot, 7, 1);                     \
    store_rows(out_slot, 7, 2); store_rows(out_slot, 7, 3);                     \
  } while (0)
    // the sign words of the NEXT masked layer (acts slot `slot`) are requested at the start of the running one: eight sync points
    // with counted waits lie between the request and the hand-over at the next layer's start
#define SNY_PREFETCH_SIGNS(slot_)                                               \
  do {                                                                          \
    unsigned so_ = s_off;                                                       \
    asm volatile("" : "+v"(so_));                                               \
    x3_load_b128(swn, so_, sign_base(slot_));                                   \
  } while (0)
#define SNY_TAKE_SIGNS()                                                        \
  do {                                                                          \
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(swn) :: "memory");                 \
    sw = swn;                                                                   \
  } while (0)

    // ---- dir_encoding.0^T (first 256 inputs): g_final = W_d[:, :256]^T g_y2; reads set 0 (8 k-steps), writes set 1
    out_slot = 8;
    SNY_LAYER(8, 0, B_D, B_H, copy_tile, 1);
    // ---- xyz_encoding_final^T (+ sigma^T on the VALU): g_y8 = (W_f^T g_final + w_sigma g_sigma) [h8 > 0]; set 1 -> set 0
    out_slot = 7;                                                          // (sw = the words of acts slot 7, loaded above)
    SNY_PREFETCH_SIGNS(6);
    SNY_LAYER(16, 1, B_H, B_H, mask_sigma_tile, 0);
    // ---- xyz_encoding_{li+1}^T, li = 7..1: g_y_{li-1} = (W^T g_y_li) [h_li > 0] (acts slot li - 1); odd li reads set 0, writes set 1
#pragma unroll 1
    for (int li = 7; li >= 1; --li) {
      out_slot = li - 1;
      SNY_TAKE_SIGNS();
      if (li >= 2) SNY_PREFETCH_SIGNS(li - 2);
      if (li == 1) SNY_LAYER(16, 0, B_H, B_D, mask_tile, 1);                // tiles 6, 7 stage the next point tile's DIRT slabs
      else if (li & 1) SNY_LAYER(16, 0, B_H, B_H, mask_tile, 1);
      else SNY_LAYER(16, 1, B_H, B_H, mask_tile, 0);
    }
#undef SNY_LW_CUR
#undef SNY_LW_NEXT
#undef SNY_SNEXT
#undef SNY_W
#undef SNY_SLAB
#undef SNY_LAYER
#undef SNY_PREFETCH_SIGNS
#undef SNY_TAKE_SIGNS
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace snk

extern "C" int SN_LAUNCH_NAME(sn_mlp_backward_chain_bf16x3)(const void* bblob, const float* acts, const float* out_raw,
                                                   const float* g_raw, long n_points, long slot_rows, float* G, float* g_out,
                                                   hipStream_t stream) {
  using namespace snk;
  if (n_points <= 0) return 0;
  const long tiles = (n_points + 127) / 128;
  if (slot_rows < tiles * 128) return -1;
  const int n_cu = snh::cu_count();
  SN_ENSURE_DYN_LDS(mlp_bwd_chain_bf16x3_kernel, BX3_LDS_BYTES);
  hipLaunchKernelGGL(mlp_bwd_chain_bf16x3_kernel, dim3((unsigned)(tiles < n_cu ? tiles : n_cu)), dim3(256), BX3_LDS_BYTES, stream,
                     reinterpret_cast<const char*>(bblob), acts, out_raw, g_raw, n_points, slot_rows, G, g_out);
  return (int)hipGetLastError();
}
