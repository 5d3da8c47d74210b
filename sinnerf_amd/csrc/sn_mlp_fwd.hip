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
#include <type_traits>

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

  asm volatile("" ::: "a0", "a255");             // size the kernel for the whole hand-managed AGPR file (sn_mlp_pipe.h)
  int cslot = 0;                                 // ring slot of the slab being consumed
  f32x4 af[2];
  af[0] = *reinterpret_cast<const f32x4*>(ring.slot(0) + lane * 16);
  af[1] = *reinterpret_cast<const f32x4*>(ring.slot(0) + lane * 16 + 1024);
  f32x16 acc0 = load_bias(lds_bias, 0, h), acc1;            // the two accumulator sets (VGPRs)
  const int n_used = ring.n_used;
  // training forward: per-wave staging tile of the activation stores (sn_mlp_pipe.h XPOSE_*)
  char* const xp = smem + MLP_F32_LDS_BYTES_V2 + wave * XPOSE_WAVE_BYTES;
  // (36-float pitch: conflict-free for the 8-lane write groups, two-way for part of the 16-lane read groups -- SQ_LDS_BANK_CONFLICT 7.7-8.1 %.
  // The conflict-free tile of the bf16x3 streams, 128-byte rows with chunk ^= row & 7, was measured here in round 5: conflict fraction 0.000
  // and 0.4-1 % SLOWER, one more VALU per staging write: profiles/r05_f32_staging_ab.txt.  The conflicts are not what this kernel waits for.)
  const unsigned xp_w = (unsigned)(j * XPOSE_PITCH + 4 * h) * 4u;                       // this lane's register quads
  const unsigned xp_r = (unsigned)((lane >> 3) * XPOSE_PITCH + 4 * (lane & 7)) * 4u;    // row lane>>3, 16-byte chunk lane&7
  const unsigned g_off = (unsigned)((lane >> 3) * 256 + 4 * (lane & 7)) * 4u;
  // epilogue staging (epi32_*_lds, sn_mlp_pipe.h): where this lane's accumulator quad q goes.  Training forward: the quad's place in the
  // staging tile of the activation stores (row j, floats 8 q + 4 h ..), which the row stores read afterwards; inference: a tile of its own.
  unsigned epi_a = STORE ? (unsigned)(size_t)xp + xp_w : (unsigned)(size_t)(smem + MLP_F32_LDS_BYTES_V2 + wave * EPI_WAVE_BYTES) + lane * 16;
  constexpr int EPI_Q = STORE ? 32 : 1024;                   // byte step between the quads of a lane
  unsigned vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));             // opaque: stays ONE live register (an immediate would be re-materialised per block)
  asm volatile("" : "+v"(epi_a));

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  const long p_wave = (tile * 4 + __builtin_amdgcn_readfirstlane(wave)) * 32;       // wave-uniform, in SGPRs
  const long p_raw = p_wave + j;
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
    int hh = h;
    asm volatile("" : "+v"(hh));                 // keep the 32 column selects inside the tile loop (else hoisted: +32 VGPRs)
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int c0 = snl::xyz_slot_col(0, e), c1 = snl::xyz_slot_col(1, e);
      const int c = hh ? c1 : c0;
      xe[e] = (c >= 0) ? row[c < 0 ? 0 : c] : 0.0f;
    }
  }
  if (STORE && INPUT_MODE == 0) {                // rows are allocated for whole 128-point tiles: no predicate.  (Pre-embedded
                                                 // rows, INPUT_MODE 1: the caller builds emb itself, it is a column re-layout of x)
    int hh = h;
    asm volatile("" : "+v"(hh));                 // the half-dependent offsets stay inside the tile loop
    store_emb_xyz(emb + p_raw * 128, xe, hh);    // columns [0, 63); the pad columns 63, 91..127 are never read back
  }

  int s = 0;                                     // slab id being consumed
  float sg = 0.0f;                               // sigma head partial of this lane half (nerf.py:136), K-slot order

  // ---- epilogue slices.  Slice q (0..3) finalises accumulator registers 4q..4q+3 of output tile t (features
  // 32t+8q+4h+(0..3)) and writes them as K-slots 16t+4q+(0..3) of activation set W.
  // training forward: the slice's four values also go to the wave's staging tile; store_rows(i) later writes row group i
  // (8 points x 128 B) of the staged 32-point x 32-feature tile to acts[slot][point][32t..32t+31], non-temporal (5 GB of
  // write-once data must not evict the L2-resident weight blob).
  auto stage = [&](int q, const float (&v)[4]) __attribute__((always_inline)) {
    if (STORE) {
      f32x4 o;
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
      *reinterpret_cast<f32x4*>(xp + xp_w + 32 * q) = o;
    }
  };
  auto store_rows = [&](int slot, int t, int i) __attribute__((always_inline)) {
    if (STORE) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(xp + xp_r + 8 * i * XPOSE_PITCH * 4);
      // wave-uniform 64-bit base (SALU) + one 32-bit per-lane offset, kept opaque: hipcc otherwise precomputes a 64-bit
      // VGPR address per slot and runs out of registers
      const char* base = reinterpret_cast<const char*>(acts) + (((long)slot * slot_rows + p_wave + 8 * i) * 256 + 32 * t) * 4;
      unsigned go = g_off;
      asm volatile("" : "+v"(go));               // opaque per store: no hoisted per-slot address registers
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(const_cast<char*>(base) + go));
    }
  };
  auto quad = [](const f32x16& r, int q) __attribute__((always_inline)) {       // registers 4q..4q+3 of an accumulator set, as they lie
    f32x4 x;
    x[0] = r[4 * q]; x[1] = r[4 * q + 1]; x[2] = r[4 * q + 2]; x[3] = r[4 * q + 3];
    return x;
  };
  auto relu_slice = [&](auto wset, int slot, int t, int q, const f32x16& r) __attribute__((always_inline)) {
    constexpr int W = decltype(wset)::value;
    if (slot == 7) {                             // layer 8 feeds the sigma head: same accumulation order as a K-slot sweep
      f32x4 v;                                   // ... which needs the activated values in VGPRs: ReLU on the VALU here
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = relu1(r[4 * q + i]);
      const f32x4 w = *reinterpret_cast<const f32x4*>(lds_aux + snl::AUX_SIGW + h * 128 + 16 * t + 4 * q);
      sg = __builtin_fmaf(w[0], v[0], sg);
      sg = __builtin_fmaf(w[1], v[1], sg);
      sg = __builtin_fmaf(w[2], v[2], sg);
      sg = __builtin_fmaf(w[3], v[3], sg);
      asm volatile("" : "+v"(sg));
      epi32_copy_lds(W * 128 + 16 * t + 4 * q, epi_a, EPI_Q * q, v);
    } else {                                     // no VALU: ds_write_b128, ReLU by LDS integer max, ds_read_b128 into the AGPRs
      epi32_relu_lds(W * 128 + 16 * t + 4 * q, epi_a, EPI_Q * q, quad(r, q), vzero);
    }
  };
  auto copy_slice = [&](auto wset, int slot, int t, int q, const f32x16& r) __attribute__((always_inline)) {   // xyz_encoding_final
    constexpr int W = decltype(wset)::value;
    epi32_copy_lds(W * 128 + 16 * t + 4 * q, epi_a, EPI_Q * q, quad(r, q));
  };
#define SN_LW_CUR (ring.slot(cslot) + lane * 16)
#define SN_LW_NEXT (ring.slot(cslot == 2 ? 0 : cslot + 1) + lane * 16)
#define SN_W(W_) std::integral_constant<int, W_>{}
  // slab of output tile T_ (literal: it ends up in asm immediates and selects the accumulator set) of layer slot SLOT_;
  // EPI_ = the previous tile's epilogue into activation set W_.  NP_ = 4 KB pieces of the slab staged at its sync point =
  // the slab TWO ahead in the stream: K/32 -> 2 (xyz_encoding_1), 8 (256-wide), 10 (skip), 9 (dir_encoding).
#define SN_SLAB(T_, NG0_, NG1_, S0_, S1_, GB_, NP_, BV_, EPI_, W_, SLOT_)                                          \
  do {                                                                                                             \
    /* training forward: DMA pieces in [GB, LS), the four row stores in the groups behind them (sn_mlp_pipe.h) */  \
    constexpr int LS_ = !STORE ? -1 : ((NG0_) + (NG1_) == 8) ? 4 : (GB_) + (NP_);                                  \
    if (((T_) & 1) == 0)                                                                                           \
      slab_f32a<NG0_, NG1_, S0_, S1_, GB_, NP_, LS_>(acc0, acc1, af, SN_LW_CUR, BV_, SN_LW_NEXT, lds_bias,    \
          (s + 1 == n_used ? 0 : s + 1), h, ring,                                                                  \
          [&](int q) __attribute__((always_inline)) { if ((T_) > 0) EPI_(SN_W(W_), SLOT_, (T_) - 1, q, acc1); },   \
          [&](int i) __attribute__((always_inline)) { if ((T_) > 0 && i < 4) store_rows(SLOT_, (T_) - 1, i); });   \
    else                                                                                                           \
      slab_f32a<NG0_, NG1_, S0_, S1_, GB_, NP_, LS_>(acc1, acc0, af, SN_LW_CUR, BV_, SN_LW_NEXT, lds_bias,    \
          (s + 1 == n_used ? 0 : s + 1), h, ring,                                                                  \
          [&](int q) __attribute__((always_inline)) { EPI_(SN_W(W_), SLOT_, (T_) - 1, q, acc0); },                 \
          [&](int i) __attribute__((always_inline)) { if (i < 4) store_rows(SLOT_, (T_) - 1, i); });               \
    ++s; cslot = (cslot == 2) ? 0 : cslot + 1;                                                                     \
  } while (0)
  // the 8 output tiles of a layer; tiles 6,7 stage the NEXT layer's slabs (NPB_); the last tile's epilogue is not deferred
#define SN_LAYER(NG0_, NG1_, S0_, S1_, GB_, NPA_, NPB_, BV_, EPI_, W_, SLOT_)   \
  do {                                                                          \
    SN_SLAB(0, NG0_, NG1_, S0_, S1_, GB_, NPA_, BV_, EPI_, W_, SLOT_);          \
    SN_SLAB(1, NG0_, NG1_, S0_, S1_, GB_, NPA_, BV_, EPI_, W_, SLOT_);          \
    SN_SLAB(2, NG0_, NG1_, S0_, S1_, GB_, NPA_, BV_, EPI_, W_, SLOT_);          \
    SN_SLAB(3, NG0_, NG1_, S0_, S1_, GB_, NPA_, BV_, EPI_, W_, SLOT_);          \
    SN_SLAB(4, NG0_, NG1_, S0_, S1_, GB_, NPA_, BV_, EPI_, W_, SLOT_);          \
    SN_SLAB(5, NG0_, NG1_, S0_, S1_, GB_, NPA_, BV_, EPI_, W_, SLOT_);          \
    SN_SLAB(6, NG0_, NG1_, S0_, S1_, GB_, NPB_, BV_, EPI_, W_, SLOT_);          \
    SN_SLAB(7, NG0_, NG1_, S0_, S1_, GB_, NPB_, BV_, EPI_, W_, SLOT_);          \
    mfma32_result_fence(acc1);                                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) EPI_(SN_W(W_), SLOT_, 7, q_, acc1);   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) store_rows(SLOT_, 7, i_);  \
  } while (0)

  // ---- layer 0: xyz_encoding_1 (nerf.py:68): reads the xyz embedding (VGPRs), writes set 0
  SN_LAYER(8, 0, -1, -1, 2, 2, 8, xe, relu_slice, 0, 0);

  // ---- layers 1..7: xyz_encoding_2..8, skip concat at layer 4 (nerf.py:70,132-134).  Odd layers read set 0 and write
  //      set 1, even layers the reverse.
#pragma unroll 1
  for (int l = 1; l < 8; ++l) {
    if (l == 4) {
      SN_LAYER(8, 32, -1, 1, 4, 10, 8, xe, relu_slice, 0, 4);
    } else if (l == 7) {
      if (SIGMA_ONLY) SN_LAYER(32, 0, 0, 0, 4, 8, 2, xe, relu_slice, 1, 7);     // tiles 6,7 stage the next point tile's layer 0
      else SN_LAYER(32, 0, 0, 0, 4, 8, 8, xe, relu_slice, 1, 7);
    } else if (l == 3) {
      SN_LAYER(32, 0, 0, 0, 4, 8, 10, xe, relu_slice, 1, 3);                    // ... the skip layer's slabs
    } else if (l == 1) {
      SN_LAYER(32, 0, 0, 0, 4, 8, 8, xe, relu_slice, 1, 1);
    } else if (l == 5) {
      SN_LAYER(32, 0, 0, 0, 4, 8, 8, xe, relu_slice, 1, 5);
    } else if (l == 2) {
      SN_LAYER(32, 0, 1, 1, 4, 8, 8, xe, relu_slice, 0, 2);
    } else {
      SN_LAYER(32, 0, 1, 1, 4, 8, 8, xe, relu_slice, 0, 6);
    }
  }

  // ---- sigma head (nerf.py:136): accumulated in layer 8's epilogues; one cross-half add
  const float sigma = sg + __shfl_xor(sg, 32, 64) + lds_aux[snl::AUX_HEADB];
  if (SIGMA_ONLY) {
    if (valid && h == 0) out[p_raw] = sigma;
    continue;
  }

  // ---- xyz_encoding_final (nerf.py:140), no activation: reads set 1, writes set 0
  SN_LAYER(32, 0, 1, 1, 4, 8, 9, xe, copy_slice, 0, 8);

  // ---- dir_encoding + ShiftedSoftplus (nerf.py:142-143): reads set 0 and the dir embedding (VGPRs).  The 32-slot direction
  // embedding is built here, not in the prologue, so that it does not occupy 16 registers through the trunk.
  float de[16];
  if (INPUT_MODE == 0) {
    const float* rp = in0 + (p / S) * 8;
    embed_dir(rp[3], rp[4], rp[5], h, de);
  } else {
    const float* row = in0 + p * (long)S;
    int hh = h;
    asm volatile("" : "+v"(hh));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c0 = snl::dir_slot_col(0, e), c1 = snl::dir_slot_col(1, e);
      const int c = hh ? c1 : c0;
      de[e] = (c >= 0) ? row[63 + (c < 0 ? 0 : c)] : 0.0f;
    }
  }
  if (STORE && INPUT_MODE == 0) {
    int hh = h;
    asm volatile("" : "+v"(hh));
    store_emb_dir(emb + p_raw * 128 + 64, de, hh);                 // columns [64, 91)
  }
  // rgb head (nerf.py:144) accumulated from the softplus outputs while they are produced: 3 rows x this half's 64 K-slots
  float c3[3] = {0.0f, 0.0f, 0.0f};
  auto ssp_slice = [&](auto, int, int t, int q, const f32x16& r) __attribute__((always_inline)) {
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = SN_NEWACT ? shifted_softplus_fast(r[4 * q + i]) : relu1(r[4 * q + i]);   // nerf.py:84 / :94
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(lds_aux + snl::AUX_RGBW + c * 128 + h * 64 + 16 * t + 4 * q);
      c3[c] = __builtin_fmaf(w[0], v[0], c3[c]);
      c3[c] = __builtin_fmaf(w[1], v[1], c3[c]);
      c3[c] = __builtin_fmaf(w[2], v[2], c3[c]);
      c3[c] = __builtin_fmaf(w[3], v[3], c3[c]);
    }
    asm volatile("" : "+v"(c3[0]), "+v"(c3[1]), "+v"(c3[2]));
    stage(q, v);
  };
  SN_SLAB(0, 32, 4, 0, -1, 4, 9, de, ssp_slice, 0, 9);
  SN_SLAB(1, 32, 4, 0, -1, 4, 9, de, ssp_slice, 0, 9);
  SN_SLAB(2, 32, 4, 0, -1, 4, 2, de, ssp_slice, 0, 9);          // tiles 2,3 stage the next point tile's layer 0
  SN_SLAB(3, 32, 4, 0, -1, 4, 2, de, ssp_slice, 0, 9);
  mfma32_result_fence(acc1);
#pragma unroll
  for (int q = 0; q < 4; ++q) ssp_slice(SN_W(0), 9, 3, q, acc1);
#pragma unroll
  for (int i = 0; i < 4; ++i) store_rows(9, 3, i);

  // ---- WidenedSigmoid (resp. Sigmoid; nerf.py:144) of the three cross-half sums
  {
    float o3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) o3[c] = c3[c] + __shfl_xor(c3[c], 32, 64) + lds_aux[snl::AUX_HEADB + 1 + c];
    if (valid && h == 0) {
      float4 o;
      o.x = rgb_activation(o3[0]);
      o.y = rgb_activation(o3[1]);
      o.z = rgb_activation(o3[2]);
      o.w = sigma;                               // cat([rgb, sigma]) nerf.py:146
      reinterpret_cast<float4*>(out)[p_raw] = o;
    }
  }
  }  // persistent tile loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // nothing may still be landing in LDS when the workgroup retires
#undef SN_LW_CUR
#undef SN_LW_NEXT
#undef SN_W
#undef SN_SLAB
#undef SN_LAYER
}

}  // namespace snk

// ---------------------------------------------------------------------------------------------------
extern "C" int SN_LAUNCH_NAME(sn_mlp_forward_f32)(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                                         int sigma_only, int input_mode, float* out, float* acts, float* emb,
                                         long slot_rows, hipStream_t stream) {
  using namespace snk;
  if (n_points <= 0) return 0;
  const long tiles = (n_points + 127) / 128;
  const bool store = acts != nullptr;
  if (store && (sigma_only || emb == nullptr || slot_rows < tiles * 128)) return -1;
  // persistent launch: one workgroup per CU (the 135 KB LDS ring admits exactly one), each walks tiles b, b+grid, ...
  const int n_cu = snh::cu_count();
  dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu)), block(256);
  const size_t lds = MLP_F32_LDS_BYTES_V2 + (store ? XPOSE_LDS_BYTES : EPI_LDS_BYTES);
  const char* b = reinterpret_cast<const char*>(blob);
#define SN_LAUNCH(SO, IM, ST)                                                                                    \
  do {                                                                                                           \
    auto kfn = mlp_fwd_f32_kernel<SO, IM, ST>;                                                                   \
    SN_ENSURE_DYN_LDS(kfn, lds);                                                                                 \
    hipLaunchKernelGGL(kfn, grid, block, lds, stream, b, in0, in1, n_points, s_or_ld, out, acts, emb, slot_rows); \
  } while (0)
  if (store) {
    if (input_mode == 0) SN_LAUNCH(false, 0, true); else SN_LAUNCH(false, 1, true);
#ifdef SN_CLASSIC_HEADS                         // the sigma-only kernels never reach the heads: sn_api.hip routes them to the main pass
  } else if (sigma_only) {
    return -4;
  } else if (input_mode == 0) {
    SN_LAUNCH(false, 0, false);
  } else {
    SN_LAUNCH(false, 1, false);
  }
#else
  } else if (input_mode == 0) {
    if (sigma_only) SN_LAUNCH(true, 0, false); else SN_LAUNCH(false, 0, false);
  } else {
    if (sigma_only) SN_LAUNCH(true, 1, false); else SN_LAUNCH(false, 1, false);
  }
#endif
#undef SN_LAUNCH
  return (int)hipGetLastError();
}
