// sn_mlp_bwd.hip -- backward "chain" of the fused NeRF MLP for gfx950 (fp32): input-gradient propagation
//   g_x[in_feature, point] = W^T * g_y ,  g_y = g_h (.) act'(.)
// for every layer, what torch autograd derives from models/nerf.py:122-148 (+ models/activations.py).
// Same scheme and the same machinery as the forward kernel (sn_mlp_fwd.hip / sn_mlp_pipe.h): persistent workgroups, a wave
// owns 32 points, the gradient w.r.t. a layer's output leaves the MFMA accumulators masked by the activation derivative
// straight into one of the two hand-managed AGPR sets and is the B operand of the next transposed layer; transposed
// weights stream L2 -> LDS through the 3-slot ring as pre-packed A fragments (sn_layout.h, "Backward-chain blob"), one
// mid-slab barrier per slab, epilogue of tile t under the MFMAs of tile t+1.  The two narrow transposed heads run on the
// VALU: g_h2 = W_rgb^T g_y3 (3 FMAs per value) and the sigma term of g_h8 (1 FMA per value, after the MFMA sum).
// (First generation of this kernel: builtin MFMAs, double-buffered Stager, barrier + epilogue + store drain on the critical
//  path of every slab -- 123 TF; see DESIGN.md.)
//
// The kernel WRITES the per-layer pre-activation gradients g_y (row-major [P][256]) -- the left operands of the
// weight-gradient contractions  dW_l = g_y_l^T X_l  over all points, which run as plain big-K GEMMs afterwards.
// Activation derivatives come from the stored forward activations:
//   ReLU (nerf.py:73)            : [h > 0]
//   ShiftedSoftplus (act.py:33)  : sigmoid(y-1) = 1 - exp(-softplus(y-1)) = 1 - exp(-h2)
//   WidenedSigmoid (act.py:18)   : .2505 * (1 - t^2),  t = tanh(.5 y) = (2*rgb - 1)/1.002
#include "sn_mlp_pipe.h"
#include <type_traits>

namespace snk {

constexpr int BWD_RING_SLOT = snl::B_MAX_SLAB_K * 128;                    // 32768
constexpr int BWD_TAIL_BYTES = snl::B_TAIL_FLOATS * 4;                    // 2816: zero "bias" slot + aux table
constexpr int MLP_BWD_LDS_BYTES = BWD_TAIL_BYTES + 3 * BWD_RING_SLOT + XPOSE_LDS_BYTES;   // 119552
typedef RingT<128, BWD_RING_SLOT> RingBk;

// masked epilogue block: four accumulator values x, four forward activations a -> v = (a > 0 ? x : 0), also written to
// a[reg..reg+3] (the next transposed layer's B operands)
SN_DEV void epi32_mask(int reg, float x0, float x1, float x2, float x3, float a0, float a1, float a2, float a3,
                       float (&v)[4]) {
  asm volatile("v_cmp_lt_f32 vcc, 0, %8\n\tv_cndmask_b32 %0, 0, %4, vcc\n\t"
               "v_cmp_lt_f32 vcc, 0, %9\n\tv_cndmask_b32 %1, 0, %5, vcc\n\t"
               "v_cmp_lt_f32 vcc, 0, %10\n\tv_cndmask_b32 %2, 0, %6, vcc\n\t"
               "v_cmp_lt_f32 vcc, 0, %11\n\tv_cndmask_b32 %3, 0, %7, vcc\n\t"
               "v_accvgpr_write_b32 a[%12], %0\n\tv_accvgpr_write_b32 a[%13], %1\n\t"
               "v_accvgpr_write_b32 a[%14], %2\n\tv_accvgpr_write_b32 a[%15], %3"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(reg), "n"(reg + 1),
                 "n"(reg + 2), "n"(reg + 3)
               : "vcc");
}

// acts / G slots: 0..7 = h1..h8 (resp. g_y of xyz_encoding_1..8), 8 = final, 9 = h2 / g_y2 (128 wide, ld 256)
__global__ void __launch_bounds__(256)
mlp_bwd_chain_f32_kernel(const char* __restrict__ bblob, const float* __restrict__ acts, const float* __restrict__ out_raw,
                         const float* __restrict__ g_raw, long P, long slot_rows, float* __restrict__ G,
                         float* __restrict__ g_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_zero = reinterpret_cast<float*>(smem);                       // the slab loop's "bias" slot: all zero
  const float* lds_aux = lds_zero + snl::B_ZERO_FLOATS;
  asm volatile("" ::: "a0", "a255");             // size the kernel for the whole hand-managed AGPR file (sn_mlp_pipe.h)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  const long n_tiles = (P + 127) / 128;

  RingBk ring;
  ring.blob = bblob;
  ring.gnext = bblob;
  ring.base = smem + BWD_TAIL_BYTES;
  ring.n_used = snl::NB_SLABS;
  ring.stage_id = 0;
  ring.stage_slot = 0;
  ring.remaining = 0;
  ring.tid = tid;
  ring.wbase = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
  ring.pieces = 0; ring.piece = 0; ring.slab_bytes = 0;
  constexpr int NP_D = 4, NP_H = 8;              // 4 KB pieces per slab: DIRT (K = 128), FINT / LT (K = 256)
#pragma unroll
  for (int i = 0; i < 2; ++i) {                  // slabs 0, 1 (both DIRT)
    ring.begin_static();
#pragma unroll
    for (int k = 0; k < NP_D; ++k) ring.piece_static();
    ring.template end_static<NP_D>();
  }
  {
    const float4* gb = reinterpret_cast<const float4*>(bblob + snl::b_tail_byte_offset());
    float4* lb = reinterpret_cast<float4*>(lds_zero);
    for (int i = tid; i < snl::B_TAIL_FLOATS / 4; i += 256) lb[i] = gb[i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                               // slabs 0,1 + zero/aux table visible

  int cslot = 0;                                 // ring slot of the slab being consumed
  f32x4 af[2];
  af[0] = *reinterpret_cast<const f32x4*>(ring.slot(0) + lane * 16);
  af[1] = *reinterpret_cast<const f32x4*>(ring.slot(0) + lane * 16 + 1024);
  f32x16 acc0 = load_bias(lds_zero, 0, h), acc1; // the two accumulator sets (VGPRs)
  // per-wave staging tile of the g_y row stores (sn_mlp_pipe.h XPOSE_*)
  char* const xp = smem + BWD_TAIL_BYTES + 3 * BWD_RING_SLOT + wave * XPOSE_WAVE_BYTES;
  // (36-float pitch: conflict-free for the 8-lane write groups, two-way for part of the 16-lane read groups -- SQ_LDS_BANK_CONFLICT 7.7-8.1 %.
  // The conflict-free tile of the bf16x3 streams, 128-byte rows with chunk ^= row & 7, was measured here in round 5: conflict fraction 0.000
  // and 0.4-1 % SLOWER, one more VALU per staging write: profiles/r05_f32_staging_ab.txt.  The conflicts are not what this kernel waits for.)
  const unsigned xp_w = (unsigned)(j * XPOSE_PITCH + 4 * h) * 4u;                       // this lane's register quads
  const unsigned xp_r = (unsigned)((lane >> 3) * XPOSE_PITCH + 4 * (lane & 7)) * 4u;    // row lane>>3, 16-byte chunk lane&7
  const unsigned g_off = (unsigned)((lane >> 3) * 256 + 4 * (lane & 7)) * 4u;

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p_wave = (tile * 4 + __builtin_amdgcn_readfirstlane(wave)) * 32;       // wave-uniform, in SGPRs
    const long p_raw = p_wave + j;
    const bool valid = p_raw < P;
    const long p = valid ? p_raw : P - 1;

    // ---- heads: g_y3 = g_rgb * d/dy WidenedSigmoid, g_sigma (both lane halves hold their point's four values)
    float gy3[3], gsig;
    f32x4 h2[16];                                // the h2 tile (slot 9) in the accumulator layout, all four feature tiles
    {
      const float* src = acts + ((long)9 * slot_rows + p) * 256 + 4 * h;
#pragma unroll
      for (int i = 0; i < 16; ++i) h2[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 8 * i));
      const float4 g = reinterpret_cast<const float4*>(g_raw)[p];
      const float4 o = reinterpret_cast<const float4*>(out_raw)[p];
      const float k = 0.5f * 1.002f * 0.5f;
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
        reinterpret_cast<float4*>(g_out)[p_raw] = gy;               // g_y of rgb.0 (3) and of sigma (1)
        // the same 4 values as a zero-padded 32-wide block in the unused half of slot 9 (columns 128..159): the A operand
        // of the rgb / sigma weight-gradient contractions (sn_dw.hip variants 4/5)
        float4* row = reinterpret_cast<float4*>(G + ((long)9 * slot_rows + p_raw) * 256 + 128);
        row[0] = gy;
#pragma unroll
        for (int q = 1; q < 8; ++q) row[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }

    // g_y slice (4 values per lane) -> the wave's staging tile; store_rows(i) later writes row group i (8 points x 128 B) of
    // the staged 32-point x 32-feature tile to G[slot][point][32t .. 32t+31] with non-temporal stores: 5 GB of write-once
    // data otherwise evict the L2-resident weight blob every workgroup streams.  Rows are allocated for whole 128-point
    // tiles (slot_rows), rows >= P receive the exact zeros their lanes computed: no predicate.
    auto stage = [&](int q, const float (&v)[4]) __attribute__((always_inline)) {
      f32x4 o;
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
      *reinterpret_cast<f32x4*>(xp + xp_w + 32 * q) = o;
    };
    auto store_rows = [&](int slot, int t, int i) __attribute__((always_inline)) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(xp + xp_r + 8 * i * XPOSE_PITCH * 4);
      const char* base = reinterpret_cast<const char*>(G) + (((long)slot * slot_rows + p_wave + 8 * i) * 256 + 32 * t) * 4;
      unsigned go = g_off;
      asm volatile("" : "+v"(go));               // opaque per store: no hoisted per-slot address registers
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(const_cast<char*>(base) + go));
    };
    // forward activation tile for the derivative mask of output tile t, accumulator layout (quad i of 4), requested one
    // slab ahead of the epilogue that consumes it
    f32x4 av[4] = {};
    auto load_act = [&](int slot, int t, int i) __attribute__((always_inline)) {
      const char* base = reinterpret_cast<const char*>(acts) + (((long)slot * slot_rows + p_wave) * 256 + 32 * t + 8 * i) * 4;
      unsigned ao = (unsigned)(j * 256 + 4 * h) * 4u;
      asm volatile("" : "+v"(ao));
      av[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + ao));
    };

    // ---- rgb.0^T on the VALU: g_h2 = W_r^T g_y3 ; g_y2 = g_h2 (1 - exp(-h2)); written to set 0 (K-slots 16t + 4q + i)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(lds_aux + snl::B_AUX_RGBT + 0 * 128 + h * 64 + 16 * t + 4 * q);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(lds_aux + snl::B_AUX_RGBT + 1 * 128 + h * 64 + 16 * t + 4 * q);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(lds_aux + snl::B_AUX_RGBT + 2 * 128 + h * 64 + 16 * t + 4 * q);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gh = __builtin_fmaf(w2[i], gy3[2], __builtin_fmaf(w1[i], gy3[1], w0[i] * gy3[0]));
          v[i] = SN_NEWACT ? gh * (1.0f - expf(-h2[4 * t + q][i])) : (h2[4 * t + q][i] > 0.0f ? gh : 0.0f);   // ReLU (nerf.py:94)
        }
        epi32_copy(16 * t + 4 * q, v[0], v[1], v[2], v[3]);
        stage(q, v);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) store_rows(9, t, i);
    }

    int mask_slot = 0;                           // acts slot of the ReLU mask of the running layer
    int out_slot = 0;                            // G slot the running layer writes
    // ---- epilogue slices.  Slice q (0..3) finalises accumulator registers 4q..4q+3 of output tile t (input features
    // 32t+8q+4h+(0..3) of the transposed layer) and writes them as K-slots 16t+4q+(0..3) of activation set W.
    auto copy_slice = [&](auto wset, int t, int q, const f32x16& r) __attribute__((always_inline)) {   // g_final: no activation
      constexpr int W = decltype(wset)::value;
      epi32_copy(W * 128 + 16 * t + 4 * q, r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
      const float v[4] = {r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
      stage(q, v);
    };
    // g_y = g_h [h > 0]; with_sigma: g_h8 also gets the sigma head's term  sigma.weight[f] g_sigma  (nerf.py:136)
    auto mask_slice_impl = [&](auto wset, auto with_sigma, int t, int q, const f32x16& r) __attribute__((always_inline)) {
      constexpr int W = decltype(wset)::value;
      constexpr bool SIG = decltype(with_sigma)::value;
      float x[4] = {r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
      if (SIG) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(lds_aux + snl::B_AUX_SIGT + h * 128 + 16 * t + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = __builtin_fmaf(w[i], gsig, x[i]);
      }
      float v[4];
      epi32_mask(W * 128 + 16 * t + 4 * q, x[0], x[1], x[2], x[3], av[q][0], av[q][1], av[q][2], av[q][3], v);
      stage(q, v);
    };
    auto mask_slice = [&](auto wset, int t, int q, const f32x16& r) __attribute__((always_inline)) {
      mask_slice_impl(wset, std::false_type{}, t, q, r);
    };
    auto mask_sigma_slice = [&](auto wset, int t, int q, const f32x16& r) __attribute__((always_inline)) {
      mask_slice_impl(wset, std::true_type{}, t, q, r);
    };
#define SNB_LW_CUR (ring.slot(cslot) + lane * 16)
#define SNB_LW_NEXT (ring.slot(cslot == 2 ? 0 : cslot + 1) + lane * 16)
#define SNB_W(W_) std::integral_constant<int, W_>{}
    // slab of output tile T_ (literal: it ends up in asm immediates and selects the accumulator set); EPI_ = the previous
    // tile's epilogue into activation set W_; its row stores and the loads of the activation tile THIS slab's epilogue needs
    // (MASK_) are the slab's memory steps.  NP_ = 4 KB pieces of the slab staged at its sync point (the slab TWO ahead).
#define SNB_SLAB(T_, NG_, SET_, NP_, EPI_, W_, MASK_)                                                              \
  do {                                                                                                             \
    /* memory steps (sn_mlp_pipe.h): DMA pieces in [4, LS), then the four row stores, then the four activation-    \
       tile loads (the scattered loads last: nothing queues up behind them) */                                     \
    constexpr int LS_ = ((NG_) == 16) ? 8 : 4 + (NP_);                                                             \
    if (((T_) & 1) == 0)                                                                                           \
      slab_f32a<NG_, 0, SET_, SET_, 4, NP_, LS_>(acc0, acc1, af, SNB_LW_CUR, static_cast<const float*>(nullptr), SNB_LW_NEXT, lds_zero, 0, h, ring, \
          [&](int q) __attribute__((always_inline)) { if ((T_) > 0) EPI_(SNB_W(W_), (T_) - 1, q, acc1); },         \
          [&](int i) __attribute__((always_inline)) {                                                              \
            if ((T_) > 0 && i < 4) store_rows(out_slot, (T_) - 1, i);                                              \
            if (MASK_ && i >= 4) load_act(mask_slot, T_, i - 4); });                                               \
    else                                                                                                           \
      slab_f32a<NG_, 0, SET_, SET_, 4, NP_, LS_>(acc1, acc0, af, SNB_LW_CUR, static_cast<const float*>(nullptr), SNB_LW_NEXT, lds_zero, 0, h, ring, \
          [&](int q) __attribute__((always_inline)) { EPI_(SNB_W(W_), (T_) - 1, q, acc0); },                       \
          [&](int i) __attribute__((always_inline)) {                                                              \
            if (i < 4) store_rows(out_slot, (T_) - 1, i);                                                          \
            if (MASK_ && i >= 4) load_act(mask_slot, T_, i - 4); });                                               \
    cslot = (cslot == 2) ? 0 : cslot + 1;                                                                          \
  } while (0)
    // the 8 output tiles of a transposed layer; tiles 6,7 stage the NEXT layer's slabs (NPB_); the last tile's epilogue is
    // not deferred
#define SNB_LAYER(NG_, SET_, NPA_, NPB_, EPI_, W_, MASK_)                       \
  do {                                                                          \
    SNB_SLAB(0, NG_, SET_, NPA_, EPI_, W_, MASK_);                              \
    SNB_SLAB(1, NG_, SET_, NPA_, EPI_, W_, MASK_);                              \
    SNB_SLAB(2, NG_, SET_, NPA_, EPI_, W_, MASK_);                              \
    SNB_SLAB(3, NG_, SET_, NPA_, EPI_, W_, MASK_);                              \
    SNB_SLAB(4, NG_, SET_, NPA_, EPI_, W_, MASK_);                              \
    SNB_SLAB(5, NG_, SET_, NPA_, EPI_, W_, MASK_);                              \
    SNB_SLAB(6, NG_, SET_, NPB_, EPI_, W_, MASK_);                              \
    SNB_SLAB(7, NG_, SET_, NPB_, EPI_, W_, MASK_);                              \
    mfma32_result_fence(acc1);                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) EPI_(SNB_W(W_), 7, q_, acc1);   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) store_rows(out_slot, 7, i_);    \
  } while (0)

    // ---- dir_encoding.0^T (first 256 inputs): g_final = W_d[:, :256]^T g_y2   (xyz_encoding_final has no activation);
    //      reads set 0 (64 K-slots), writes set 1
    out_slot = 8;
    SNB_LAYER(16, 0, NP_D, NP_H, copy_slice, 1, false);
    // ---- xyz_encoding_final^T (+ sigma^T on the VALU): g_y8 = (W_f^T g_final + w_sigma g_sigma) [h8 > 0]; set 1 -> set 0
    mask_slot = 7; out_slot = 7;
    SNB_LAYER(32, 1, NP_H, NP_H, mask_sigma_slice, 0, true);
    // ---- xyz_encoding_{li+1}^T, li = 7..1: g_y_{li-1} = (W^T g_y_li) [h_li > 0]; odd li reads set 0 and writes set 1
#pragma unroll 1
    for (int li = 7; li >= 1; --li) {
      mask_slot = li - 1; out_slot = li - 1;
      if (li == 1) SNB_LAYER(32, 0, NP_H, NP_D, mask_slice, 1, true);       // tiles 6,7 stage the next point tile's DIRT slabs
      else if (li & 1) SNB_LAYER(32, 0, NP_H, NP_H, mask_slice, 1, true);
      else SNB_LAYER(32, 1, NP_H, NP_H, mask_slice, 0, true);
    }
#undef SNB_LW_CUR
#undef SNB_LW_NEXT
#undef SNB_W
#undef SNB_SLAB
#undef SNB_LAYER
  }  // persistent tile loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // nothing may still be landing in LDS when the workgroup retires
}

}  // namespace snk

extern "C" int SN_LAUNCH_NAME(sn_mlp_backward_chain_f32)(const void* bblob, const float* acts, const float* out_raw,
                                                const float* g_raw, long n_points, long slot_rows, float* G,
                                                float* g_out, hipStream_t stream) {
  using namespace snk;
  if (n_points <= 0) return 0;
  if (slot_rows < (n_points + 127) / 128 * 128) return -1;      // whole 128-point tiles of G are written
  const long tiles = (n_points + 127) / 128;
  if (tiles > 0x7fffffffL) return -2;
  // persistent launch: one workgroup per CU (the 117 KB LDS footprint admits exactly one), each walks tiles b, b+grid, ...
  const int n_cu = snh::cu_count();
  auto kfn = mlp_bwd_chain_f32_kernel;
  SN_ENSURE_DYN_LDS(kfn, MLP_BWD_LDS_BYTES);
  hipLaunchKernelGGL(kfn, dim3((unsigned)(tiles < n_cu ? tiles : n_cu)), dim3(256), MLP_BWD_LDS_BYTES, stream,
                     reinterpret_cast<const char*>(bblob), acts, out_raw, g_raw, n_points, slot_rows, G, g_out);
  return (int)hipGetLastError();
}
