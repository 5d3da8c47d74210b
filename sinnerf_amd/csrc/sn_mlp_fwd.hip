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
// = 128 points shares each weight slab through LDS (3-slot ring, one mid-slab barrier per slab: sn_mlp_pipe.h).
// Per point tile: 290 x 32 MFMAs of 64 cycles (289.75 algorithmic: only the 63->64 input pad) -> MFMA-bound by
// construction (fp32 roofline 157.3 TF).  The 1-row sigma and 3-row rgb heads run on the VALU from registers.
#include "sn_mlp_pipe.h"

namespace snk {

// INPUT_MODE 0: points from (rays, z_vals):  p -> ray = p / S, xyz = o + d*z     (render_rays path)
// INPUT_MODE 1: pre-embedded rows x[p, 0:63(+27)] with leading dimension ld       (NeRF.forward path)
// STORE: training forward -- additionally writes every layer's activations (acts[10][slot_rows][256]: h1..h8, final, h2)
// and the embedded inputs (emb[slot_rows][128], zero-filled by the caller: xyz columns 0..62, dir columns 64..90,
// reference column order) for the backward pass.
template <bool SIGMA_ONLY, int INPUT_MODE, bool STORE>
__global__ void __launch_bounds__(256)
mlp_fwd_f32_kernel(const char* __restrict__ blob, const float* __restrict__ in0, const float* __restrict__ in1,
                   long P, int S, float* __restrict__ out, float* __restrict__ acts, float* __restrict__ emb, long slot_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_bias = reinterpret_cast<float*>(smem);
  const float* lds_aux = lds_bias + snl::BIAS_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  const long n_tiles = (P + 127) / 128;
  const long my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;     // persistent: tiles b, b+grid, ...

  Ring ring;
  ring.blob = blob;
  ring.gnext = blob;
  ring.base = smem + TAIL_LDS_BYTES;
  ring.n_used = SIGMA_ONLY ? snl::SLAB_FIN : snl::N_SLABS;
  ring.stage_id = 0;
  ring.stage_slot = 0;
  ring.remaining = my_tiles * ring.n_used;      // (only the prologue staging checks it)
  ring.tid = tid;
  ring.wbase = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
  ring.pieces = 0; ring.piece = 0;
  ring.stage_whole();                            // slab 0
  ring.stage_whole();                            // slab 1
  {
    const float4* gb = reinterpret_cast<const float4*>(blob + snl::bias_byte_offset(snl::DT_F32));
    float4* lb = reinterpret_cast<float4*>(lds_bias);
    for (int i = tid; i < snl::TAIL_FLOATS / 4; i += 256) lb[i] = gb[i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                               // slabs 0,1 + bias/aux table visible

  int cslot = 0;                                 // ring slot of the slab being consumed
  f32x4 af[2];
  af[0] = *reinterpret_cast<const f32x4*>(ring.slot(0) + lane * 16);
  af[1] = *reinterpret_cast<const f32x4*>(ring.slot(0) + lane * 16 + 1024);
  f32x16 acc = load_bias(lds_bias, 0, h);
  const int n_used = ring.n_used;

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  const long p_raw = (tile * 4 + wave) * 32 + j;
  const bool valid = p_raw < P;
  const long p = valid ? p_raw : P - 1;

  float xe[32];
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
  } else {
    const float* row = in0 + p * (long)S;        // S = leading dimension here
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int c0 = snl::xyz_slot_col(0, e), c1 = snl::xyz_slot_col(1, e);
      const int c = h ? c1 : c0;
      xe[e] = (c >= 0) ? row[c < 0 ? 0 : c] : 0.0f;
    }
  }
  if (STORE && valid) {
    float* er = emb + p_raw * 128;               // caller zero-fills emb: pad columns 63, 91..127 stay 0
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int c0 = snl::xyz_slot_col(0, e), c1 = snl::xyz_slot_col(1, e);
      const int c = h ? c1 : c0;
      if (c >= 0) er[c] = xe[e];
    }
  }

  int s = 0;                                     // slab id being consumed
  float hid[128], nxt[128];
  f32x16 pacc, acc_pre;

  // Epilogue slices: slice q (0..3) finalises accumulator registers 4q..4q+3 of an output tile = features 32t+8q+4h+(0..3),
  // i.e. exactly one 16-byte store of the row-major activation matrix in the training variant.
  auto store_slice = [&](int slot, int t, int q, const float* v) {
    if (STORE && valid) {
      float4 o;
      o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
      *reinterpret_cast<float4*>(acts + ((long)slot * slot_rows + p_raw) * 256 + 32 * t + 4 * h + 8 * q) = o;
    }
  };
  auto relu_slice = [&](int slot, int t, int q, const f32x16& a) {
#pragma unroll
    for (int r = 4 * q; r < 4 * q + 4; ++r) nxt[16 * t + r] = relu1(a[r]);   // one v_max_f32; the asm also pins it here
    store_slice(slot, t, q, nxt + 16 * t + 4 * q);
  };
  auto relu_tile = [&](int slot, int t, const f32x16& a) {                    // un-overlapped form (last tile of a layer)
#pragma unroll
    for (int q = 0; q < 4; ++q) relu_slice(slot, t, q, a);
  };
#define SN_LW_CUR (ring.slot(cslot) + lane * 16)
#define SN_LW_NEXT (ring.slot(cslot == 2 ? 0 : cslot + 1) + lane * 16)
#define SN_ADVANCE() do { pacc = acc; acc = acc_pre; ++s; cslot = (cslot == 2) ? 0 : cslot + 1; } while (0)

  // NP argument of each slab = 4 KB pieces of the slab staged at its sync point = the slab TWO ahead in the stream:
  // K/32 -> 2 (xyz_encoding_1), 8 (256-wide), 10 (skip), 9 (dir_encoding).
#define SN_SLAB(NG0_, NG1_, GB_, NP_, B0_, B1_, PEND_)                                                               \
  do {                                                                                                              \
    slab_f32<NG0_, NG1_, GB_, NP_>(acc, af, acc_pre, SN_LW_CUR, B0_, B1_, SN_LW_NEXT, lds_bias,                      \
                                   (s + 1 == n_used ? 0 : s + 1), h, ring, PEND_);                                   \
    SN_ADVANCE();                                                                                                   \
  } while (0)

  // ---- layer 0: xyz_encoding_1  (nerf.py:68)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    auto pend = [&](int q) { if (t > 0) relu_slice(0, t - 1, q, pacc); };
    if (t < 6) SN_SLAB(8, 0, 2, 2, xe, xe, pend); else SN_SLAB(8, 0, 2, 8, xe, xe, pend);
  }
  relu_tile(0, 7, pacc);
#pragma unroll
  for (int i = 0; i < 128; ++i) hid[i] = nxt[i];

  // ---- layers 1..7: xyz_encoding_2..8, skip concat at layer 4 (nerf.py:70,132-134)
#pragma unroll 1
  for (int l = 1; l < 8; ++l) {
    if (l == 4) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        auto pend = [&](int q) { if (t > 0) relu_slice(l, t - 1, q, pacc); };
        if (t < 6) SN_SLAB(8, 32, 4, 10, xe, hid, pend); else SN_SLAB(8, 32, 4, 8, xe, hid, pend);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        auto pend = [&](int q) { if (t > 0) relu_slice(l, t - 1, q, pacc); };
        if (t < 6) {
          SN_SLAB(32, 0, 4, 8, hid, hid, pend);
        } else if (l == 3) {                      // the slab two ahead belongs to the skip layer
          SN_SLAB(32, 0, 4, 10, hid, hid, pend);
        } else if (SIGMA_ONLY && l == 7) {        // ... or to xyz_encoding_1 of the next tile
          SN_SLAB(32, 0, 4, 2, hid, hid, pend);
        } else {
          SN_SLAB(32, 0, 4, 8, hid, hid, pend);
        }
      }
    }
    relu_tile(l, 7, pacc);
#pragma unroll
    for (int i = 0; i < 128; ++i) hid[i] = nxt[i];
  }

  // ---- sigma head (nerf.py:136) on the VALU: this lane half's 128 K-slots, then one cross-half add
  float sigma;
  {
    const f32x4* ws = reinterpret_cast<const f32x4*>(lds_aux + snl::AUX_SIGW + h * 128);
    float sg = 0.0f;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const f32x4 w = ws[q];
      sg = __builtin_fmaf(w[0], hid[4 * q + 0], sg);
      sg = __builtin_fmaf(w[1], hid[4 * q + 1], sg);
      sg = __builtin_fmaf(w[2], hid[4 * q + 2], sg);
      sg = __builtin_fmaf(w[3], hid[4 * q + 3], sg);
    }
    sigma = sg + __shfl_xor(sg, 32, 64) + lds_aux[snl::AUX_HEADB];
  }
  if (SIGMA_ONLY) {
    if (valid && h == 0) out[p_raw] = sigma;
    continue;
  }

  // ---- xyz_encoding_final (nerf.py:140), no activation
  auto copy_slice = [&](int slot, int t, int q, const f32x16& a) {
#pragma unroll
    for (int r = 4 * q; r < 4 * q + 4; ++r) {
      float v = a[r];
      asm volatile("" : "+v"(v));
      nxt[16 * t + r] = v;
    }
    store_slice(slot, t, q, nxt + 16 * t + 4 * q);
  };
  auto copy_tile = [&](int slot, int t, const f32x16& a) {
#pragma unroll
    for (int q = 0; q < 4; ++q) copy_slice(slot, t, q, a);
  };
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    auto pend = [&](int q) { if (t > 0) copy_slice(8, t - 1, q, pacc); };
    if (t < 6) SN_SLAB(32, 0, 4, 8, hid, hid, pend); else SN_SLAB(32, 0, 4, 9, hid, hid, pend);    // tiles 6,7 stage dir_encoding
  }
  copy_tile(8, 7, pacc);
#pragma unroll
  for (int i = 0; i < 128; ++i) hid[i] = nxt[i];

  // ---- dir_encoding + ShiftedSoftplus (nerf.py:142-143).  The 32-slot direction embedding is built here, not in the
  // prologue, so that it does not occupy 16 registers through the trunk.
  float de[16];
  if (INPUT_MODE == 0) {
    const float* rp = in0 + (p / S) * 8;
    embed_dir(rp[3], rp[4], rp[5], h, de);
  } else {
    const float* row = in0 + p * (long)S;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c0 = snl::dir_slot_col(0, e), c1 = snl::dir_slot_col(1, e);
      const int c = h ? c1 : c0;
      de[e] = (c >= 0) ? row[63 + (c < 0 ? 0 : c)] : 0.0f;
    }
  }
  if (STORE && valid) {
    float* er = emb + p_raw * 128;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c0 = snl::dir_slot_col(0, e), c1 = snl::dir_slot_col(1, e);
      const int c = h ? c1 : c0;
      if (c >= 0) er[64 + c] = de[e];
    }
  }
  float h2[64];
  auto ssp_slice = [&](int t, int q, const f32x16& a) {
#pragma unroll
    for (int r = 4 * q; r < 4 * q + 4; ++r) h2[16 * t + r] = shifted_softplus_fast(a[r]);
    store_slice(9, t, q, h2 + 16 * t + 4 * q);
  };
  auto ssp_tile = [&](int t, const f32x16& a) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ssp_slice(t, q, a);
  };
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    auto pend = [&](int q) { if (t > 0) ssp_slice(t - 1, q, pacc); };
    if (t < 2) SN_SLAB(32, 4, 4, 9, hid, de, pend); else SN_SLAB(32, 4, 4, 2, hid, de, pend);       // tiles 2,3 stage the next tile's layer 0
  }
  ssp_tile(3, pacc);

  // ---- rgb + WidenedSigmoid (nerf.py:144) on the VALU: 3 rows x this half's 64 K-slots, cross-half add
  {
    float c3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x4* wr = reinterpret_cast<const f32x4*>(lds_aux + snl::AUX_RGBW + c * 128 + h * 64);
      float a = 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const f32x4 w = wr[q];
        a = __builtin_fmaf(w[0], h2[4 * q + 0], a);
        a = __builtin_fmaf(w[1], h2[4 * q + 1], a);
        a = __builtin_fmaf(w[2], h2[4 * q + 2], a);
        a = __builtin_fmaf(w[3], h2[4 * q + 3], a);
      }
      c3[c] = a + __shfl_xor(a, 32, 64) + lds_aux[snl::AUX_HEADB + 1 + c];
    }
    if (valid && h == 0) {
      float4 o;
      o.x = widened_sigmoid(c3[0]);
      o.y = widened_sigmoid(c3[1]);
      o.z = widened_sigmoid(c3[2]);
      o.w = sigma;                               // cat([rgb, sigma]) nerf.py:146
      reinterpret_cast<float4*>(out)[p_raw] = o;
    }
  }
  }  // persistent tile loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // nothing may still be landing in LDS when the workgroup retires
#undef SN_LW_CUR
#undef SN_LW_NEXT
#undef SN_ADVANCE
#undef SN_SLAB
}

}  // namespace snk

// ---------------------------------------------------------------------------------------------------
extern "C" int sn_mlp_forward_f32_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                                         int sigma_only, int input_mode, int use_dma, float* out, float* acts, float* emb,
                                         long slot_rows, hipStream_t stream) {
  using namespace snk;
  (void)use_dma;                                 // the register-staged ablation path was retired with the v2 pipeline
  if (n_points <= 0) return 0;
  const long tiles = (n_points + 127) / 128;
  const bool store = acts != nullptr;
  if (store && (sigma_only || input_mode != 0 || emb == nullptr || slot_rows < n_points)) return -1;
  // persistent launch: one workgroup per CU (the 135 KB LDS ring admits exactly one), each walks tiles b, b+grid, ...
  int dev = 0, n_cu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
  dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu)), block(256);
  const size_t lds = MLP_F32_LDS_BYTES_V2;
  const char* b = reinterpret_cast<const char*>(blob);
#define SN_LAUNCH(SO, IM, ST)                                                                                    \
  do {                                                                                                           \
    auto kfn = mlp_fwd_f32_kernel<SO, IM, ST>;                                                                   \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return (int)e;                                                                          \
    hipLaunchKernelGGL(kfn, grid, block, lds, stream, b, in0, in1, n_points, s_or_ld, out, acts, emb, slot_rows); \
  } while (0)
  if (store) {
    SN_LAUNCH(false, 0, true);
  } else if (input_mode == 0) {
    if (sigma_only) SN_LAUNCH(true, 0, false); else SN_LAUNCH(false, 0, false);
  } else {
    if (sigma_only) SN_LAUNCH(true, 1, false); else SN_LAUNCH(false, 1, false);
  }
#undef SN_LAUNCH
  return (int)hipGetLastError();
}
