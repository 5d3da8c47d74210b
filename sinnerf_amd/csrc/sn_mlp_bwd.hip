// sn_mlp_bwd.hip -- backward "chain" of the fused NeRF MLP for gfx950 (fp32): input-gradient propagation
//   g_x[in_feature, point] = W^T * g_y ,  g_y = g_h (.) act'(.)
// for every layer, what torch autograd derives from models/nerf.py:122-148 (+ models/activations.py).
// Same scheme as the forward kernel (sn_mlp_fwd.hip): the gradient w.r.t. a layer's output lives in the MFMA
// accumulator layout and is fed unchanged as the B operand of the next transposed layer; transposed weights
// stream L2 -> LDS as pre-packed A fragments (sn_layout.h, "backward-chain blob").
//
// The kernel WRITES the per-layer pre-activation gradients g_y (row-major [P][256]) -- the left operands of the
// weight-gradient contractions  dW_l = g_y_l^T X_l  over all points, which run as plain big-K GEMMs afterwards.
// Activation derivatives come from the stored forward activations:
//   ReLU (nerf.py:73)            : [h > 0]
//   ShiftedSoftplus (act.py:33)  : sigmoid(y-1) = 1 - exp(-softplus(y-1)) = 1 - exp(-h2)
//   WidenedSigmoid (act.py:18)   : .2505 * (1 - t^2),  t = tanh(.5 y) = (2*rgb - 1)/1.002
#include "sn_mlp_common.h"

namespace snk {

constexpr int BWD_SLAB_LDS_BYTES = snl::B_MAX_SLAB_K * 128;      // 36864
// + a per-wave 32-point x 32-feature staging tile: the g_y tiles leave the accumulator layout (lane = point: a 16-byte
// piece per lane with a 1 KB lane stride, 64 cache lines per store instruction -- measured 0.8 ms of a 5.4 ms launch) as
// whole 128-byte rows (8 lanes x 16 B per point row, 8 rows per instruction)
constexpr int BWD_XP_PITCH = 36;                                 // floats per staged row: conflict-free b128 both ways
constexpr int BWD_XP_WAVE_BYTES = 32 * BWD_XP_PITCH * 4;         // 4608
constexpr int MLP_BWD_LDS_BYTES = 2 * BWD_SLAB_LDS_BYTES + 4 * BWD_XP_WAVE_BYTES;   // 92160

__device__ __forceinline__ int bslab_k_rt(int s) { return s < 4 ? 32 : s < 12 ? 128 : s < 20 ? 288 : 256; }

SN_DEV f32x16 zero_acc() {
  f32x16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.0f;
  return a;
}

// acts / G slots: 0..7 = h1..h8 (resp. g_y of xyz_encoding_1..8), 8 = final, 9 = h2 / g_y2 (128 wide, ld 256)
__global__ void __launch_bounds__(256)
mlp_bwd_chain_f32_kernel(const char* __restrict__ bblob, const float* __restrict__ acts, const float* __restrict__ out_raw,
                         const float* __restrict__ g_raw, long P, long slot_rows, float* __restrict__ G,
                         float* __restrict__ g_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const buf0 = smem;
  char* const buf1 = smem + BWD_SLAB_LDS_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  char* const xp = smem + 2 * BWD_SLAB_LDS_BYTES + wave * BWD_XP_WAVE_BYTES;
  const unsigned xp_w = (unsigned)(j * BWD_XP_PITCH + 4 * h) * 4u;                          // this lane's register quads
  const unsigned xp_r = (unsigned)((lane >> 3) * BWD_XP_PITCH + 4 * (lane & 7)) * 4u;       // row lane>>3, 16-byte chunk lane&7
  const long p_wave = ((long)blockIdx.x * 4 + wave) * 32;
  const long p_raw = p_wave + j;
  const bool valid = p_raw < P;
  const long p = valid ? p_raw : P - 1;

  Stager st;
  const char* gnext = bblob;
  st.issue(gnext, buf0, snl::bslab_k(0) / 32, tid);
  gnext += snl::bslab_k(0) * 128;

  float b_rgb[16], b_sig[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { b_rgb[i] = 0.0f; b_sig[i] = 0.0f; }
  if (h == 0) {
    const float4 g = reinterpret_cast<const float4*>(g_raw)[p];
    const float4 o = reinterpret_cast<const float4*>(out_raw)[p];
    const float k = 0.5f * 1.002f * 0.5f;
    const float tx = (2.0f * o.x - 1.0f) * (1.0f / 1.002f), ty = (2.0f * o.y - 1.0f) * (1.0f / 1.002f),
                tz = (2.0f * o.z - 1.0f) * (1.0f / 1.002f);
    float4 gy;
    gy.x = valid ? g.x * k * (1.0f - tx * tx) : 0.0f;
    gy.y = valid ? g.y * k * (1.0f - ty * ty) : 0.0f;
    gy.z = valid ? g.z * k * (1.0f - tz * tz) : 0.0f;
    gy.w = valid ? g.w : 0.0f;
    b_rgb[0] = gy.x; b_rgb[1] = gy.y; b_rgb[2] = gy.z;
    b_sig[0] = gy.w;
    if (valid) {
      reinterpret_cast<float4*>(g_out)[p_raw] = gy;               // g_y of rgb.0 (3) and of sigma (1)
      // the same 4 values as a zero-padded 32-wide block in the unused half of slot 9 (columns 128..159): the A operand
      // of the rgb / sigma weight-gradient contractions (sn_dw.hip variants 4/5)
      float4* row = reinterpret_cast<float4*>(G + ((long)9 * slot_rows + p_raw) * 256 + 128);
      row[0] = gy;
#pragma unroll
      for (int q = 1; q < 8; ++q) row[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int s = 0;
  const int last_slab = snl::NB_SLABS - 1;

#define SNB_BEGIN(cur, oth)                                                  \
  {                                                                          \
    if (s < last_slab) {                                                     \
      const int kn = bslab_k_rt(s + 1);                                      \
      st.issue(gnext, (oth), kn >> 5, tid);                                  \
      gnext += kn * 128;                                                     \
    }                                                                        \
  }                                                                          \
  f32x16 acc = zero_acc();                                                   \
  const char* lw = (cur) + lane * 16;
#define SNB_END()                                                            \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           \
  __syncthreads();                                                           \
  ++s;
  // forward activation tile for the derivative mask (same 4 x float4 pattern as the forward's store).
  // (Measured and dropped for this fp32 kernel: reading the masks as per-tile sign words written by the training forward, as
  //  the bf16-state chain does.  These accumulator-layout loads cost 0.39 of 5.05 ms (timing build without them), but applying a
  //  mask bit to an fp32 value takes 4 VALU instructions against the 2 of compare + select on the loaded activation, and
  //  the forward pays 2 more per value to collect the bits: chain 4.86 -> 4.80, forward 5.11 -> 5.27 ms -- a net loss.
  //  A bit-cast AND form of the mask (2 instructions) returned wrong values from the builtin-MFMA accumulators.)
#define SNB_LOAD_ACT(slot, t)                                                                          \
  f32x4 av[4];                                                                                         \
  {                                                                                                    \
    const float* src = acts + ((long)(slot) * slot_rows + p) * 256 + 32 * (t) + 4 * h;                 \
    _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) av[q4] = *reinterpret_cast<const f32x4*>(src + 8 * q4); \
  }
  // g_y tile (16 values per lane) -> G[slot][point][32t .. 32t+31] through the wave's staging tile, with non-temporal
  // stores: 5 GB of write-once data otherwise evict the L2-resident weight blob every workgroup streams (measured -5 %).  Rows are allocated for
  // whole 128-point tiles (slot_rows), rows >= P receive the exact zeros their lanes computed: no predicate.
#define SNB_STORE_G(slot, t, arr, off)                                                                 \
  {                                                                                                    \
    _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) {                                                 \
      f32x4 v;                                                                                         \
      v[0] = arr[(off) + 4 * q4 + 0]; v[1] = arr[(off) + 4 * q4 + 1];                                  \
      v[2] = arr[(off) + 4 * q4 + 2]; v[3] = arr[(off) + 4 * q4 + 3];                                  \
      *reinterpret_cast<f32x4*>(xp + xp_w + 32 * q4) = v;                                              \
    }                                                                                                  \
    char* gb = reinterpret_cast<char*>(G) + (((long)(slot) * slot_rows + p_wave) * 256 + 32 * (t)) * 4 \
               + ((lane >> 3) * 256 + 4 * (lane & 7)) * 4;                                             \
    _Pragma("unroll") for (int i4 = 0; i4 < 4; ++i4) {                                                 \
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + xp_r + i4 * 8 * BWD_XP_PITCH * 4);          \
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(gb + i4 * 8 * 1024));   /* streaming: */ \
    }                                                                                                  \
  }

  // ---- rgb.0^T : g_h2 = W_r^T g_y3 ; g_y2 = g_h2 * (1 - exp(-h2))
  float g2[64];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    SNB_BEGIN((t & 1) ? buf1 : buf0, (t & 1) ? buf0 : buf1)
    SNB_LOAD_ACT(9, t)
    mma_f32<4>(acc, lw, b_rgb);
#pragma unroll
    for (int r = 0; r < 16; ++r) g2[16 * t + r] = acc[r] * (1.0f - expf(-av[r >> 2][r & 3]));
    SNB_STORE_G(9, t, g2, 16 * t)
    SNB_END()
  }
  // ---- dir_encoding.0^T (first 256 inputs): g_final = W_d[:, :256]^T g_y2   (xyz_encoding_final has no activation)
  float gh[128], nxt[128];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    SNB_BEGIN((t & 1) ? buf1 : buf0, (t & 1) ? buf0 : buf1)
    mma_f32<16>(acc, lw, g2);
#pragma unroll
    for (int r = 0; r < 16; ++r) nxt[16 * t + r] = acc[r];
    SNB_STORE_G(8, t, nxt, 16 * t)
    SNB_END()
  }
#pragma unroll
  for (int i = 0; i < 128; ++i) gh[i] = nxt[i];
  // ---- [xyz_encoding_final ; sigma]^T : g_h8 = W_f^T g_final + W_sigma^T g_sigma ; g_y8 = g_h8 * [h8 > 0]
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    SNB_BEGIN((t & 1) ? buf1 : buf0, (t & 1) ? buf0 : buf1)
    SNB_LOAD_ACT(7, t)
    mma_f32<32>(acc, lw, gh);
    mma_f32<4>(acc, lw + 32 * 1024, b_sig);
#pragma unroll
    for (int r = 0; r < 16; ++r) nxt[16 * t + r] = (av[r >> 2][r & 3] > 0.0f) ? acc[r] : 0.0f;
    SNB_STORE_G(7, t, nxt, 16 * t)
    SNB_END()
  }
#pragma unroll
  for (int i = 0; i < 128; ++i) gh[i] = nxt[i];
  // ---- xyz_encoding_{li+1}^T for li = 7..1 : g_h_li = W^T g_y ; g_y_{li-1} = g_h_li * [h_li > 0]
#pragma unroll 1
  for (int li = 7; li >= 1; --li) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      SNB_BEGIN((t & 1) ? buf1 : buf0, (t & 1) ? buf0 : buf1)
      SNB_LOAD_ACT(li - 1, t)
      mma_f32<32>(acc, lw, gh);
#pragma unroll
      for (int r = 0; r < 16; ++r) nxt[16 * t + r] = (av[r >> 2][r & 3] > 0.0f) ? acc[r] : 0.0f;
      SNB_STORE_G(li - 1, t, nxt, 16 * t)
      SNB_END()
    }
#pragma unroll
    for (int i = 0; i < 128; ++i) gh[i] = nxt[i];
  }
#undef SNB_BEGIN
#undef SNB_END
#undef SNB_LOAD_ACT
#undef SNB_STORE_G
}

}  // namespace snk

extern "C" int sn_mlp_backward_chain_f32_launch(const void* bblob, const float* acts, const float* out_raw,
                                                const float* g_raw, long n_points, long slot_rows, float* G,
                                                float* g_out, hipStream_t stream) {
  using namespace snk;
  if (n_points <= 0) return 0;
  if (slot_rows < (n_points + 127) / 128 * 128) return -1;      // whole 128-point tiles of G are written
  const long tiles = (n_points + 127) / 128;
  if (tiles > 0x7fffffffL) return -2;
  auto kfn = mlp_bwd_chain_f32_kernel;
  SN_ENSURE_DYN_LDS(kfn, MLP_BWD_LDS_BYTES);
  hipLaunchKernelGGL(kfn, dim3((unsigned)tiles), dim3(256), MLP_BWD_LDS_BYTES, stream,
                     reinterpret_cast<const char*>(bblob), acts, out_raw, g_raw, n_points, slot_rows, G, g_out);
  return (int)hipGetLastError();
}
