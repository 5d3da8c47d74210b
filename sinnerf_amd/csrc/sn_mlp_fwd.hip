// sn_mlp_fwd.hip -- fused NeRF MLP forward for gfx950 (MI355X), fp32 path.
//
// Replaces, for one batch of sample points, the reference's per-chunk sequence
//   xyz = o + d*z                     models/rendering.py:284-285 / :317-318
//   Embedding(xyz), Embedding(dir)    models/nerf.py:36-41  (rendering.py:198, :261)
//   cat / repeat_interleave           models/rendering.py:189-201
//   NeRF.forward                      models/nerf.py:122-148 (8x Linear+ReLU, skip, sigma, final, dir, rgb)
// with ONE kernel: every activation stays in registers, weights stream L2 -> LDS as pre-packed MFMA
// A fragments (sn_layout.h), contractions run on v_mfma_f32_32x32x2_f32 (exact fp32, fmaf-chain order).
//
// Work decomposition: a wave owns 32 consecutive points (MFMA columns); a 256-thread workgroup = 4 waves
// = 128 points shares each weight slab through LDS (double buffered, one barrier per slab).
// Per point tile: 296 x 32 MFMAs of 64 cycles -> MFMA-bound by construction (fp32 roofline 157.3 TF).
#include "sn_mlp_common.h"

namespace snk {

// Slots this lane half computes for the first layer / skip layer (sn_layout.h: xyz_slot_col).
SN_DEV void embed_xyz(float x, float y, float z, int h, float* xe) {
  const Rev2 px = to_revolutions(x), py = to_revolutions(y), pz = to_revolutions(z);
  const float hs = h ? 32.0f : 1.0f;            // bands 5..9 on the upper lane half
#pragma unroll
  for (int p = 0; p < 15; ++p) {
    const Rev2 pc = (p % 3 == 0) ? px : (p % 3 == 1) ? py : pz;
    const float scale = hs * (float)(1 << (p / 3));
    sincos_rev(pc, scale, xe[2 * p], xe[2 * p + 1]);
  }
  xe[30] = h ? z : x;
  xe[31] = h ? 0.0f : y;
}
SN_DEV void embed_dir(float x, float y, float z, int h, float* de) {
  const Rev2 px = to_revolutions(x), py = to_revolutions(y), pz = to_revolutions(z);
  const float hs = h ? 4.0f : 1.0f;             // bands 2,3 on the upper lane half
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const Rev2 pc = (p % 3 == 0) ? px : (p % 3 == 1) ? py : pz;
    const float scale = hs * (float)(1 << (p / 3));
    sincos_rev(pc, scale, de[2 * p], de[2 * p + 1]);
  }
  de[12] = h ? z : x;
  de[13] = h ? 0.0f : y;
  de[14] = 0.0f;
  de[15] = 0.0f;
}

// INPUT_MODE 0: points from (rays, z_vals):  p -> ray = p / S, xyz = o + d*z     (render_rays path)
// INPUT_MODE 1: pre-embedded rows x[p, 0:63(+27)] with leading dimension ld       (NeRF.forward path)
// STORE: training forward -- additionally writes every layer's activations (acts[10][P][256]: h1..h8, final, h2) and the
// embedded inputs (emb[P][128], zero-filled by the caller: xyz columns 0..62, dir columns 64..90, reference column order) for the backward pass.
template <bool DMA, bool SIGMA_ONLY, int INPUT_MODE, bool STORE>
__global__ void __launch_bounds__(256)
mlp_fwd_f32_kernel(const char* __restrict__ blob, const float* __restrict__ in0, const float* __restrict__ in1,
                   long P, int S, float* __restrict__ out, float* __restrict__ acts, float* __restrict__ emb, long slot_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_bias = reinterpret_cast<float*>(smem);
  char* const buf0 = smem + BIAS_LDS_BYTES;
  char* const buf1 = buf0 + SLAB_LDS_BYTES_F32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  const long p_raw = ((long)blockIdx.x * 4 + wave) * 32 + j;
  const bool valid = p_raw < P;
  const long p = valid ? p_raw : P - 1;

  Stager<DMA> st;
  const char* gnext = blob;
  // stage slab 0 and the bias table while the embeddings are computed
  st.issue(gnext, buf0, snl::slab_k(0) / 32, tid);
  gnext += snl::slab_k(0) * 128;
  {
    const float4* gb = reinterpret_cast<const float4*>(blob + snl::bias_byte_offset(snl::DT_F32));
    float4* lb = reinterpret_cast<float4*>(lds_bias);
    for (int i = tid; i < snl::BIAS_FLOATS / 4; i += 256) lb[i] = gb[i];
  }

  float xe[32], de[16];
  if (INPUT_MODE == 0) {
    const long ray = p / S;
    const float* rp = in0 + ray * 8;
    const float ox = rp[0], oy = rp[1], oz = rp[2], dx = rp[3], dy = rp[4], dz = rp[5];
    const float zz = in1[p];
    // xyz = o + d*z with separate roundings (torch: mul then add, rendering.py:284-285)
    const float x = __fadd_rn(ox, __fmul_rn(dx, zz));
    const float y = __fadd_rn(oy, __fmul_rn(dy, zz));
    const float z = __fadd_rn(oz, __fmul_rn(dz, zz));
    embed_xyz(x, y, z, h, xe);
    if (!SIGMA_ONLY) embed_dir(dx, dy, dz, h, de);
  } else {
    const float* row = in0 + p * (long)S;        // S = leading dimension here
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int c0 = snl::xyz_slot_col(0, e), c1 = snl::xyz_slot_col(1, e);
      const int c = h ? c1 : c0;
      xe[e] = (c >= 0) ? row[c < 0 ? 0 : c] : 0.0f;
    }
    if (!SIGMA_ONLY) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int c0 = snl::dir_slot_col(0, e), c1 = snl::dir_slot_col(1, e);
        const int c = h ? c1 : c0;
        de[e] = (c >= 0) ? row[63 + (c < 0 ? 0 : c)] : 0.0f;
      }
    }
  }
  st.commit(buf0, tid);
  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int s = 0;                                     // index of the slab being consumed
  const int last_slab = SIGMA_ONLY ? snl::SLAB_SIG : snl::N_SLABS - 1;
  float hid[128], nxt[128];
  // activation tile store (accumulator layout -> row-major [P][256]): 4 x 16 B per lane per 32-feature tile
#define SN_STORE_TILE(slot, t, arr, off)                                                               \
  if (STORE && valid) {                                                                                \
    float* dst = acts + ((long)(slot) * slot_rows + p_raw) * 256 + 32 * (t) + 4 * h;                   \
    _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) {                                                 \
      float4 v;                                                                                        \
      v.x = arr[(off) + 4 * q4 + 0]; v.y = arr[(off) + 4 * q4 + 1];                                    \
      v.z = arr[(off) + 4 * q4 + 2]; v.w = arr[(off) + 4 * q4 + 3];                                    \
      *reinterpret_cast<float4*>(dst + 8 * q4) = v;                                                    \
    }                                                                                                  \
  }
  if (STORE && valid) {
    float* er = emb + p_raw * 128;        // caller zero-fills emb: pad columns 63, 91..127 stay 0
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int c0 = snl::xyz_slot_col(0, e), c1 = snl::xyz_slot_col(1, e);
      const int c = h ? c1 : c0;
      if (c >= 0) er[c] = xe[e];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c0 = snl::dir_slot_col(0, e), c1 = snl::dir_slot_col(1, e);
      const int c = h ? c1 : c0;
      if (c >= 0) er[64 + c] = de[e];
    }
  }

  // Common per-slab prologue / epilogue.  `cur`/`oth` are compile-time buffer choices.
#define SN_SLAB_BEGIN(cur, oth)                                             \
  {                                                                         \
    if (s < last_slab) {                                                    \
      const int kn = slab_k_rt(s + 1);                                      \
      st.issue(gnext, (oth), kn >> 5, tid);                                 \
      gnext += kn * 128;                                                    \
    }                                                                       \
  }                                                                         \
  f32x16 acc = load_bias(lds_bias, s, h);                                   \
  const char* lw = (cur) + lane * 16;
#define SN_SLAB_END(oth)                                                    \
  if (s < last_slab) st.commit((oth), tid);                                 \
  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 \
  __syncthreads();                                                          \
  ++s;

  // ---- layer 0: xyz_encoding_1  (nerf.py:68)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    SN_SLAB_BEGIN((t & 1) ? buf1 : buf0, (t & 1) ? buf0 : buf1)
    mma_f32<8>(acc, lw, xe);
#pragma unroll
    for (int r = 0; r < 16; ++r) nxt[16 * t + r] = fmaxf(acc[r], 0.0f);
    SN_STORE_TILE(0, t, nxt, 16 * t)
    SN_SLAB_END((t & 1) ? buf0 : buf1)
  }
#pragma unroll
  for (int i = 0; i < 128; ++i) hid[i] = nxt[i];

  // ---- layers 1..7: xyz_encoding_2..8, skip concat at layer 4 (nerf.py:70,132-134)
#pragma unroll 1
  for (int l = 1; l < 8; ++l) {
    if (l == 4) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        SN_SLAB_BEGIN((t & 1) ? buf1 : buf0, (t & 1) ? buf0 : buf1)
        mma_f32<8>(acc, lw, xe);
        mma_f32<32>(acc, lw + 8 * 1024, hid);
#pragma unroll
        for (int r = 0; r < 16; ++r) nxt[16 * t + r] = fmaxf(acc[r], 0.0f);
        SN_STORE_TILE(l, t, nxt, 16 * t)
        SN_SLAB_END((t & 1) ? buf0 : buf1)
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        SN_SLAB_BEGIN((t & 1) ? buf1 : buf0, (t & 1) ? buf0 : buf1)
        mma_f32<32>(acc, lw, hid);
#pragma unroll
        for (int r = 0; r < 16; ++r) nxt[16 * t + r] = fmaxf(acc[r], 0.0f);
        SN_STORE_TILE(l, t, nxt, 16 * t)
        SN_SLAB_END((t & 1) ? buf0 : buf1)
      }
    }
#pragma unroll
    for (int i = 0; i < 128; ++i) hid[i] = nxt[i];
  }

  // ---- sigma head (nerf.py:136): row 0 of its tile -> accumulator register 0 of lanes 0..31
  float sigma;
  {
    SN_SLAB_BEGIN(buf0, buf1)
    mma_f32<32>(acc, lw, hid);
    sigma = acc[0];
    SN_SLAB_END(buf1)
  }
  if (SIGMA_ONLY) {
    if (valid && h == 0) out[p_raw] = sigma;
    return;
  }

  // ---- xyz_encoding_final (nerf.py:140), no activation.  slabs 65..72 start on buf1.
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    SN_SLAB_BEGIN((t & 1) ? buf0 : buf1, (t & 1) ? buf1 : buf0)
    mma_f32<32>(acc, lw, hid);
#pragma unroll
    for (int r = 0; r < 16; ++r) nxt[16 * t + r] = acc[r];
    SN_STORE_TILE(8, t, nxt, 16 * t)
    SN_SLAB_END((t & 1) ? buf1 : buf0)
  }
#pragma unroll
  for (int i = 0; i < 128; ++i) hid[i] = nxt[i];

  // ---- dir_encoding + ShiftedSoftplus (nerf.py:142-143).  slabs 73..76 start on buf1.
  float h2[64];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    SN_SLAB_BEGIN((t & 1) ? buf0 : buf1, (t & 1) ? buf1 : buf0)
    mma_f32<32>(acc, lw, hid);
    mma_f32<4>(acc, lw + 32 * 1024, de);
#pragma unroll
    for (int r = 0; r < 16; ++r) h2[16 * t + r] = shifted_softplus(acc[r]);
    SN_STORE_TILE(9, t, h2, 16 * t)
    SN_SLAB_END((t & 1) ? buf1 : buf0)
  }

  // ---- rgb + WidenedSigmoid (nerf.py:144): rows 0..2 -> registers 0..2 of lanes 0..31.  slab 77 on buf1.
  {
    SN_SLAB_BEGIN(buf1, buf0)
    mma_f32<16>(acc, lw, h2);
    if (valid && h == 0) {
      float4 o;
      o.x = widened_sigmoid(acc[0]);
      o.y = widened_sigmoid(acc[1]);
      o.z = widened_sigmoid(acc[2]);
      o.w = sigma;                               // cat([rgb, sigma]) nerf.py:146
      reinterpret_cast<float4*>(out)[p_raw] = o;
    }
  }
#undef SN_SLAB_BEGIN
#undef SN_SLAB_END
#undef SN_STORE_TILE
}

}  // namespace snk

// ---------------------------------------------------------------------------------------------------
extern "C" int sn_mlp_forward_f32_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                                         int sigma_only, int input_mode, int use_dma, float* out, float* acts, float* emb,
                                         long slot_rows, hipStream_t stream) {
  using namespace snk;
  if (n_points <= 0) return 0;
  const long tiles = (n_points + 127) / 128;
  if (tiles > 0x7fffffffL) return -2;
  const bool store = acts != nullptr;
  if (store && (sigma_only || input_mode != 0 || emb == nullptr || slot_rows < n_points)) return -1;
  dim3 grid((unsigned)tiles), block(256);
  const size_t lds = MLP_F32_LDS_BYTES;
  const char* b = reinterpret_cast<const char*>(blob);
#define SN_LAUNCH(DMA, SO, IM, ST)                                                                               \
  do {                                                                                                           \
    auto kfn = mlp_fwd_f32_kernel<DMA, SO, IM, ST>;                                                              \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return (int)e;                                                                          \
    hipLaunchKernelGGL(kfn, grid, block, lds, stream, b, in0, in1, n_points, s_or_ld, out, acts, emb, slot_rows); \
  } while (0)
  if (store) {
    SN_LAUNCH(true, false, 0, true);
  } else if (input_mode == 0) {
    if (use_dma) { if (sigma_only) SN_LAUNCH(true, true, 0, false); else SN_LAUNCH(true, false, 0, false); }
    else         { if (sigma_only) SN_LAUNCH(false, true, 0, false); else SN_LAUNCH(false, false, 0, false); }
  } else {
    if (sigma_only) SN_LAUNCH(true, true, 1, false); else SN_LAUNCH(true, false, 1, false);
  }
#undef SN_LAUNCH
  return (int)hipGetLastError();
}
