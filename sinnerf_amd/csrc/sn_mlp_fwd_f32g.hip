// sn_mlp_fwd_f32g.hip -- fused NeRF MLP forward for gfx950 (MI355X), fp32 INFERENCE path, third generation (round 6).
//
// Replaces the same reference sequence as sn_mlp_fwd.hip (xyz = o + d z, Embedding, cat, NeRF.forward: models/rendering.py:189-201,
// :284-285, models/nerf.py:36-41, :122-148) with the same arithmetic (v_mfma_f32_32x32x2_f32 in the same K order, the same VALU
// heads: bit-identical outputs), but a data flow built on what tools/ubench/f32_gap_cost.hip measured
// (profiles/r06_ubench_f32_gap_cost.txt):
//
//   * the f32-input MFMA runs on the f32 VECTOR pipe.  A VALU instruction between two MFMAs is not hidden in the MFMA's shadow
//     as it is beside the bf16 MFMAs: a gap with n VALU instructions costs 9.6 + 4 n cycles of matrix time (v_mov, v_max,
//     v_accvgpr_write alike) -- while s_nop, s_waitcnt, SALU, ds_read / ds_write, global loads cost NOTHING there.  (The round-1 law
//     "64.2 N_mfma + 4.6 N_other" had lumped the classes together.)
//   * so the trunk carries NO VALU instruction at all:
//       - A fragments come STRAIGHT FROM L2 into a register ring (buffer_load_dwordx4 with a constant per-lane offset and SGPR slab
//         offsets: no per-lane address arithmetic), eight groups = 32 MFMAs = 2 048 cycles ahead of their use.  No LDS ring, no
//         LDS-DMA, no per-slab barrier: the four waves of a workgroup never wait for each other (the packed weights, 2.4 MB, stay
//         resident in every XCD's 4 MB L2; one wave reads 1 KB per 256 cycles = 16 B/clk per CU, a quarter of the L1 path);
//       - the epilogue of a 32 x 32 output tile is an LDS round trip (sn_mlp_pipe.h epi32_relu_lds): ds_write_b128 of the
//         accumulators, ReLU as ds_max_i32 against 0, ds_read_b128 straight into the AGPRs of the next layer's B operand;
//       - it is deferred into the next slab's first groups ACROSS layer boundaries too (the next layer reads a tile's K-slots only
//         in its groups 4 t ..), so no layer end drains the matrix pipe.
//   * what VALU work is left sits where it cannot be avoided: the embeddings (exact range reduction), the sigma head's 256 FMAs,
//     ShiftedSoftplus and the rgb head -- 2.6 k instructions per 128-point tile beside 9 280 MFMAs per wave.
// The training forward (activation stores) and the backward chain stay on sn_mlp_fwd.hip / sn_mlp_bwd.hip (LDS ring, same epilogues).
#include "sn_mlp_f32g.h"

namespace snk {

constexpr unsigned slab_byte_offset(int s) { return (unsigned)(snl::slab_elem_offset(s) * 4); }

// INPUT_MODE 0: points from (rays, z_vals); 1: pre-embedded rows x[p, 0:63(+27)] with leading dimension ld (NeRF.forward path)
// STORE: training forward -- additionally writes every layer's activations (acts[10][slot_rows][256]: h1..h8, final, h2) and the embedded
// inputs (emb[slot_rows][128]: xyz columns 0..62, dir columns 64..90, reference column order; pre-embedded rows: the caller builds emb)
// for the backward pass, exactly as mlp_fwd_f32_kernel<.., STORE> did (sn_mlp_fwd.hip): same values, same layout.
template <bool SIGMA_ONLY, int INPUT_MODE, bool STORE>
__global__ void __launch_bounds__(256)
mlp_fwd_f32g_kernel(const char* __restrict__ blob, const float* __restrict__ in0, const float* __restrict__ in1, long P, int S,
                    float* __restrict__ out, float* __restrict__ acts, float* __restrict__ emb, long slot_rows) {
  constexpr int FD = STORE ? FD_STORE : FD_INFER;
  constexpr int NSTEPS = STORE ? EPI_STEPS_STORE : EPI_STEPS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_bias = reinterpret_cast<float*>(smem);
  const float* lds_aux = lds_bias + snl::BIAS_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  const long n_tiles = (P + 127) / 128;
  constexpr int N_USED = SIGMA_ONLY ? snl::SLAB_FIN : snl::N_SLABS;

  {
    const float4* gb = reinterpret_cast<const float4*>(blob + snl::bias_byte_offset(snl::DT_F32));
    float4* lb = reinterpret_cast<float4*>(lds_bias);
    for (int i = tid; i < snl::TAIL_FLOATS / 4; i += 256) lb[i] = gb[i];
  }
  __syncthreads();                               // bias / head table visible; the ONLY barrier of the kernel

  asm volatile("" ::: "a0", "a255");             // size the kernel for the whole hand-managed AGPR file (sn_mlp_pipe.h)
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(blob), 0, (int)snl::blob_bytes(snl::DT_F32), 0x00020000);
  unsigned voff = lane * 16;
  asm volatile("" : "+v"(voff));
  f32x4 fr[FD];
#pragma unroll
  for (int g = 0; g < FD - 1; ++g) fr[g] = load_frag(rs, voff, g * 1024);      // slab 0, groups 0 .. FD-2
  f32x16 acc0 = load_bias(lds_bias, 0, h), acc1;
  // epilogue staging: where this lane's accumulator quad q goes.  Inference: a tile of its own (quad q of lane l at 1024 q + 16 l).
  // Training forward: the quad's place in the [point row][feature] staging tile of the activation stores (row j, floats 8 q + 4 h ..;
  // 36-float pitch, sn_mlp_pipe.h XPOSE_*), which the row stores read afterwards.
  char* const xp = smem + TAIL_LDS_BYTES + wave * XPOSE_WAVE_BYTES;
  const unsigned xp_w = (unsigned)(j * XPOSE_PITCH + 4 * h) * 4u;
  const unsigned xp_r = (unsigned)((lane >> 3) * XPOSE_PITCH + 4 * (lane & 7)) * 4u;    // row lane>>3, 16-byte chunk lane&7
  unsigned g_off = (unsigned)((lane >> 3) * 256 + 4 * (lane & 7)) * 4u;
  constexpr int EQ = STORE ? 32 : 1024;          // byte step between the quads of a lane
  unsigned epi_a = STORE ? (unsigned)(size_t)xp + xp_w : (unsigned)(size_t)(smem + TAIL_LDS_BYTES + wave * EPI_WAVE_BYTES) + lane * 16;
  asm volatile("" : "+v"(g_off));
  f32x4 rowbuf[2];
  unsigned vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  asm volatile("" : "+v"(epi_a));

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  const long p_wave = (tile * 4 + __builtin_amdgcn_readfirstlane(wave)) * 32;
  const long p_raw = p_wave + j;
  const bool valid = p_raw < P;
  const long p = valid ? p_raw : P - 1;

  float xe[32];
  if (INPUT_MODE == 0) {
#ifdef SN_F32G_NO_RAY_LOADS
    const float ox = 0.1f, oy = 0.2f, oz = 0.3f, dx = 0.5f, dy = 0.4f, dz = -0.7f, zz = 2.0f + 1e-6f * (float)p;
#else
    const long ray = p / S;
    const float* rp = in0 + ray * 8;
    const float ox = rp[0], oy = rp[1], oz = rp[2], dx = rp[3], dy = rp[4], dz = rp[5];
    const float zz = in1[p];
#endif
    // xyz = o + d*z with separate roundings (torch: mul then add, rendering.py:284-285)
    const float x = __fadd_rn(ox, __fmul_rn(dx, zz));
    const float y = __fadd_rn(oy, __fmul_rn(dy, zz));
    const float z = __fadd_rn(oz, __fmul_rn(dz, zz));
#ifdef SN_F32G_NO_VALU_BLOCKS                    // timing build: the trunk alone (no embedding / sigma head / softplus arithmetic)
#pragma unroll
    for (int e = 0; e < 32; ++e) xe[e] = x + (float)e;
#else
    embed_xyz(x, y, z, h, xe);
#endif
  } else {
    const float* row = in0 + p * (long)S;        // S = leading dimension here
    int hh = h;
    asm volatile("" : "+v"(hh));
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int c0 = snl::xyz_slot_col(0, e), c1 = snl::xyz_slot_col(1, e);
      const int c = hh ? c1 : c0;
      xe[e] = (c >= 0) ? row[c < 0 ? 0 : c] : 0.0f;
    }
  }

  if (STORE && INPUT_MODE == 0) {                // rows are allocated for whole 128-point tiles: no predicate.  (Pre-embedded rows: the
    int hh = h;                                  // caller builds emb itself, it is a column re-layout of x)
    asm volatile("" : "+v"(hh));                 // the half-dependent offsets stay inside the tile loop
    store_emb_xyz(emb + p_raw * 128, xe, hh);    // columns [0, 63); the pad columns 63, 91..127 are never read back
  }

  float sg = 0.0f;                               // sigma head partial of this lane half (nerf.py:136), K-slot order
  float sv[16];                                  // layer 8: the activated values of a tile between the VALU step and their LDS writes

  auto quad = [](const f32x16& r, int q) __attribute__((always_inline)) {
    f32x4 x;
    x[0] = r[4 * q]; x[1] = r[4 * q + 1]; x[2] = r[4 * q + 2]; x[3] = r[4 * q + 3];
    return x;
  };
  // epilogue PROGRAMS of output tile t of a layer (accumulator registers 4q..4q+3 -> K-slots 16t+4q.. of activation set W), one
  // instruction per step: slice q = st / 6 -- st % 6 == 0: ds_write_b128 of the quad, 1..4: ReLU of one word, 5: ds_read_b128 -> AGPRs
  // training forward, steps 24..29: row group i (8 points x 128 B) of the staged tile -> acts[slot][point][32t..32t+31], non-temporal (5 GB
  // of write-once data must not evict the L2-resident weights); every read two steps ahead of its store
  const long p_wave_c = p_wave;
  auto row_read = [&](int i) __attribute__((always_inline)) {         // (compiler-tracked LDS load: it places the lgkmcnt wait of the store)
    return *reinterpret_cast<const f32x4*>(xp + xp_r + 8 * i * XPOSE_PITCH * 4);
  };
  auto row_store = [&](int slot, int t, int i, const f32x4& o) __attribute__((always_inline)) {
    // wave-uniform 64-bit base (SALU) + one 32-bit per-lane offset (made opaque once, outside the tile loop)
    const char* base = reinterpret_cast<const char*>(acts) + (((long)slot * slot_rows + p_wave_c + 8 * i) * 256 + 32 * t) * 4;
    // (no per-store opaque copy of the offset: a v_mov is a VALU gap of 13.6 cycles next to the f32-input MFMA; hipcc selects the saddr form)
#ifdef SN_F32G_NO_ROW_STORES                     // timing build: the staging round trip without the global stores
    asm volatile("" :: "v"(o), "s"(base), "v"(g_off));
#else
    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(const_cast<char*>(base) + g_off));
#endif
  };
  auto row_step = [&](int slot, int t, int k) __attribute__((always_inline)) {
    if (!STORE) return;
    if (k == 0) rowbuf[0] = row_read(0);
    else if (k == 1) rowbuf[1] = row_read(1);
    else if (k == 2) { row_store(slot, t, 0, rowbuf[0]); rowbuf[0] = row_read(2); }
    else if (k == 3) { row_store(slot, t, 1, rowbuf[1]); rowbuf[1] = row_read(3); }
    else if (k == 4) row_store(slot, t, 2, rowbuf[0]);
    else if (k == 5) row_store(slot, t, 3, rowbuf[1]);
  };
  f32x4 sw[4];                                   // layer 8: the sigma head's weights of a tile (requested a step ahead of their use)
  auto relu_prog = [&](auto wset, auto slot_c, int t, int st, const f32x16& r) __attribute__((always_inline)) {
    constexpr int W = decltype(wset)::value;
    constexpr int slot = decltype(slot_c)::value;
    const int q = st / 6, k = st % 6;
    if (slot == 7) {                             // layer 8 feeds the sigma head (nerf.py:136), which needs the activated values in VGPRs:
      if (st == 0) {                             // ReLU and the 16 FMAs of a tile on the VALU, in ONE gap (9.6 + 4 n cycles per gap)
#pragma unroll
        for (int i = 0; i < 4; ++i) sw[i] = *reinterpret_cast<const f32x4*>(lds_aux + snl::AUX_SIGW + h * 128 + 16 * t + 4 * i);
      } else if (st == 3) {
#ifdef SN_F32G_NO_VALU_BLOCKS
#pragma unroll
        for (int i = 0; i < 16; ++i) sv[i] = r[i];
        if (false)
#endif
#pragma unroll
        for (int i = 0; i < 16; ++i) sv[i] = relu1(r[i]);
#pragma unroll
#ifndef SN_F32G_NO_VALU_BLOCKS
        for (int i = 0; i < 16; ++i) sg = __builtin_fmaf(sw[i / 4][i % 4], sv[i], sg);        // same order as a K-slot sweep
#else
        for (int i = 0; i < 1; ++i) sg += sv[0];
#endif
        asm volatile("" : "+v"(sg));
      } else if (st >= 4 && st < 8) {
        f32x4 x;
        x[0] = sv[4 * (st - 4)]; x[1] = sv[4 * (st - 4) + 1]; x[2] = sv[4 * (st - 4) + 2]; x[3] = sv[4 * (st - 4) + 3];
        lds_put_quad(epi_a, EQ * (st - 4), x);
      } else if (st >= 8 && st < 12) {
        lds_get_quad_agpr(W * 128 + 16 * t + 4 * (st - 8), epi_a, EQ * (st - 8));
      } else if (st >= 24) {
        row_step(7, t, st - 24);
      }
      return;
    }
    if (st >= 24) { row_step(slot, t, st - 24); return; }
    if (k == 0) lds_put_quad(epi_a, EQ * q, quad(r, q));
#ifndef SN_F32G_NO_ATOMICS
    else if (k < 5) lds_relu_word(epi_a, EQ * q + 4 * (k - 1), vzero);
#endif
    else if (k == 5) lds_get_quad_agpr(W * 128 + 16 * t + 4 * q, epi_a, EQ * q);
  };
  auto copy_prog = [&](auto wset, auto, int t, int st, const f32x16& r) __attribute__((always_inline)) {       // xyz_encoding_final
    constexpr int W = decltype(wset)::value;
    const int q = st / 6, k = st % 6;
    if (st >= 24) { row_step(8, t, st - 24); return; }
    if (k == 0) lds_put_quad(epi_a, EQ * q, quad(r, q));
    else if (k == 5) lds_get_quad_agpr(W * 128 + 16 * t + 4 * q, epi_a, EQ * q);
  };
  auto no_prog = [&](auto, auto, int, int, const f32x16&) __attribute__((always_inline)) {};

#define SN_C(V_) std::integral_constant<int, V_>{}
  // slab S_ (stream index, literal) = output tile S_ % 8 of its layer; its first groups run the epilogue PEPI_ of the previous slab
  // (tile PT_ of the layer that writes set PW_ / slot PSLOT_).  Even slabs accumulate in acc0, odd ones in acc1.
#define SN_SLABG(S_, NG0_, NG1_, S0_, S1_, BV_, PEPI_, PW_, PSLOT_, PT_)                                                  \
  do {                                                                                                                    \
    constexpr int s_nx = ((S_) + 1 == N_USED) ? 0 : (S_) + 1;                                                             \
    constexpr int g0 = (int)(slab_byte_offset(S_) / 1024), tot = (int)(slab_byte_offset(N_USED) / 1024);                  \
    if (((S_) & 1) == 0)                                                                                                  \
      slab_f32g<NG0_, NG1_, S0_, S1_, g0, tot, NSTEPS, FD>(acc0, acc1, fr, rs, voff, BV_,                                             \
          lds_bias, s_nx, h, [&](int st) __attribute__((always_inline)) { PEPI_(SN_C(PW_), SN_C(PSLOT_), PT_, st, acc1); }); \
    else                                                                                                                  \
      slab_f32g<NG0_, NG1_, S0_, S1_, g0, tot, NSTEPS, FD>(acc1, acc0, fr, rs, voff, BV_,                                             \
          lds_bias, s_nx, h, [&](int st) __attribute__((always_inline)) { PEPI_(SN_C(PW_), SN_C(PSLOT_), PT_, st, acc0); }); \
  } while (0)
  // the 8 output tiles of layer slot L_ (stream slabs 8 L_ ..): tile 0 finishes the PREVIOUS layer's tile 7 (PEPI_ / PW_ / PSLOT_)
#define SN_LAYERG(L_, NG0_, NG1_, S0_, S1_, EPI_, W_, PEPI_, PW_, PSLOT_)                 \
  do {                                                                                    \
    SN_SLABG(8 * (L_) + 0, NG0_, NG1_, S0_, S1_, xe, PEPI_, PW_, PSLOT_, 7);              \
    SN_SLABG(8 * (L_) + 1, NG0_, NG1_, S0_, S1_, xe, EPI_, W_, L_, 0);                    \
    SN_SLABG(8 * (L_) + 2, NG0_, NG1_, S0_, S1_, xe, EPI_, W_, L_, 1);                    \
    SN_SLABG(8 * (L_) + 3, NG0_, NG1_, S0_, S1_, xe, EPI_, W_, L_, 2);                    \
    SN_SLABG(8 * (L_) + 4, NG0_, NG1_, S0_, S1_, xe, EPI_, W_, L_, 3);                    \
    SN_SLABG(8 * (L_) + 5, NG0_, NG1_, S0_, S1_, xe, EPI_, W_, L_, 4);                    \
    SN_SLABG(8 * (L_) + 6, NG0_, NG1_, S0_, S1_, xe, EPI_, W_, L_, 5);                    \
    SN_SLABG(8 * (L_) + 7, NG0_, NG1_, S0_, S1_, xe, EPI_, W_, L_, 6);                    \
  } while (0)

  // layer 0 (xyz_encoding_1, nerf.py:68) reads the xyz embedding (VGPRs), writes set 0; odd layers read set 0 and write set 1, even
  // layers the reverse; the skip layer (nerf.py:133) reads the embedding first, then set 1
  SN_LAYERG(0, 8, 0, -1, -1, relu_prog, 0, no_prog, 0, 0);
  SN_LAYERG(1, 32, 0, 0, 0, relu_prog, 1, relu_prog, 0, 0);
  SN_LAYERG(2, 32, 0, 1, 1, relu_prog, 0, relu_prog, 1, 1);
  SN_LAYERG(3, 32, 0, 0, 0, relu_prog, 1, relu_prog, 0, 2);
  SN_LAYERG(4, 8, 32, -1, 1, relu_prog, 0, relu_prog, 1, 3);
  SN_LAYERG(5, 32, 0, 0, 0, relu_prog, 1, relu_prog, 0, 4);
  SN_LAYERG(6, 32, 0, 1, 1, relu_prog, 0, relu_prog, 1, 5);
  SN_LAYERG(7, 32, 0, 0, 0, relu_prog, 1, relu_prog, 0, 6);

  if (SIGMA_ONLY) {
    mfma32_result_fence(acc1);                   // slab 63's result: the one epilogue of the tile that is not deferred
#pragma unroll
    for (int st = 0; st < EPI_STEPS; ++st) relu_prog(SN_C(1), SN_C(7), 7, st, acc1);
    const float sigma = sg + __shfl_xor(sg, 32, 64) + lds_aux[snl::AUX_HEADB];
    if (valid && h == 0) out[p_raw] = sigma;
    continue;                                    // (acc0 already holds slab 0's bias: slab 63 requested it as its s_next)
  }

  // xyz_encoding_final (nerf.py:140), no activation: reads set 1, writes set 0; its first slab finishes layer 8 (and the sigma head)
  SN_LAYERG(8, 32, 0, 1, 1, copy_prog, 0, relu_prog, 1, 7);
  const float sigma = sg + __shfl_xor(sg, 32, 64) + lds_aux[snl::AUX_HEADB];     // nerf.py:136

  // dir_encoding + ShiftedSoftplus (nerf.py:142-143): reads set 0 and the dir embedding (VGPRs)
  float de[16];
  if (INPUT_MODE == 0) {
    const float* rp = in0 + (p / S) * 8;
#ifdef SN_F32G_NO_VALU_BLOCKS
#pragma unroll
    for (int e = 0; e < 16; ++e) de[e] = rp[3] + (float)e;
#else
    embed_dir(rp[3], rp[4], rp[5], h, de);
#endif
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
  auto ssp_slice = [&](auto, auto, int t, int st, const f32x16& r) __attribute__((always_inline)) {
    if (st >= 24) { row_step(9, t, st - 24); return; }
    if (st % 6 != 0) return;                     // VALU work (ShiftedSoftplus + the rgb head): slice q in ONE gap, six MFMAs apart
    const int q = st / 6;
    float v[4];
#pragma unroll
#ifdef SN_F32G_NO_VALU_BLOCKS
    for (int i = 0; i < 4; ++i) v[i] = r[4 * q + i];
#else
    for (int i = 0; i < 4; ++i) v[i] = SN_NEWACT ? shifted_softplus_fast(r[4 * q + i]) : relu1(r[4 * q + i]);   // nerf.py:84 / :94
#endif
#pragma unroll
#ifdef SN_F32G_NO_VALU_BLOCKS
    for (int c = 0; c < 1; ++c) c3[0] += v[0];
    if (false)
#endif
    for (int c = 0; c < 3; ++c) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(lds_aux + snl::AUX_RGBW + c * 128 + h * 64 + 16 * t + 4 * q);
      c3[c] = __builtin_fmaf(w[0], v[0], c3[c]);
      c3[c] = __builtin_fmaf(w[1], v[1], c3[c]);
      c3[c] = __builtin_fmaf(w[2], v[2], c3[c]);
      c3[c] = __builtin_fmaf(w[3], v[3], c3[c]);
    }
    asm volatile("" : "+v"(c3[0]), "+v"(c3[1]), "+v"(c3[2]));
    if (STORE) {                                 // slot 9 of the training state: the softplus outputs
      f32x4 o;
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
      *reinterpret_cast<f32x4*>(xp + xp_w + 32 * q) = o;
    }
  };
  SN_SLABG(72, 32, 4, 0, -1, de, copy_prog, 0, 8, 7);
  SN_SLABG(73, 32, 4, 0, -1, de, ssp_slice, 0, 9, 0);
  SN_SLABG(74, 32, 4, 0, -1, de, ssp_slice, 0, 9, 1);
  SN_SLABG(75, 32, 4, 0, -1, de, ssp_slice, 0, 9, 2);
  mfma32_result_fence(acc1);
#pragma unroll
  for (int q = 0; q < 4; ++q) ssp_slice(SN_C(0), SN_C(9), 3, 6 * q, acc1);
  if (STORE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) row_store(9, 3, i, row_read(i));
  }

  // WidenedSigmoid (resp. Sigmoid; nerf.py:144) of the three cross-half sums
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
#undef SN_C
#undef SN_SLABG
#undef SN_LAYERG
}

}  // namespace snk

// ---------------------------------------------------------------------------------------------------
// The file is compiled as TWO translation units (csrc/Makefile; compile time: six instantiations of a fully unrolled point tile): the
// inference kernels (sn_mlp_forward_f32g[_classic]_launch) and, with -DSN_F32G_TU_STORE, the training forward
// (sn_mlp_forward_f32g_store[_classic]_launch).
#ifdef SN_F32G_TU_STORE
#define SN_F32G_ENTRY sn_mlp_forward_f32g_store
#else
#define SN_F32G_ENTRY sn_mlp_forward_f32g
#endif
#define SN_F32G_LAUNCH_NAME(base) SN_LAUNCH_NAME(base)        // (one more level: the argument is expanded before ## pastes it)
extern "C" int SN_F32G_LAUNCH_NAME(SN_F32G_ENTRY)(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                                             int sigma_only, int input_mode, float* out, float* acts, float* emb, long slot_rows,
                                             hipStream_t stream) {
  using namespace snk;
  if (n_points <= 0) return 0;
  const long tiles = (n_points + 127) / 128;
  const bool store = acts != nullptr;
  if (store && (sigma_only || emb == nullptr || slot_rows < tiles * 128)) return -1;
  const int n_cu = snh::cu_count();              // persistent: one workgroup per CU (~400 registers per lane: one wave per SIMD)
  dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu)), block(256);
  const size_t lds = store ? F32G_LDS_BYTES_STORE : F32G_LDS_BYTES;
  const char* b = reinterpret_cast<const char*>(blob);
#define SN_LAUNCH(SO, IM, ST)                                                                                       \
  do {                                                                                                              \
    auto kfn = mlp_fwd_f32g_kernel<SO, IM, ST>;                                                                     \
    SN_ENSURE_DYN_LDS(kfn, lds);                                                                                    \
    hipLaunchKernelGGL(kfn, grid, block, lds, stream, b, in0, in1, n_points, s_or_ld, out, acts, emb, slot_rows);   \
  } while (0)
#if defined(SN_F32G_AB)                         // timing builds (tools/build_variant_f32g.sh): ONE instantiation: the frame render's, or
#ifdef SN_F32G_AB_STORE                         // the training forward's with -DSN_F32G_AB_STORE (which replaces BOTH objects)
  if (!store || input_mode != 0) return -4;
  SN_LAUNCH(false, 0, true);
#else
  if (store || sigma_only || input_mode != 0) return -4;
  SN_LAUNCH(false, 0, false);
#endif
#elif defined(SN_F32G_TU_STORE)
  if (!store) return -4;
  if (input_mode == 0) SN_LAUNCH(false, 0, true); else SN_LAUNCH(false, 1, true);
#elif defined(SN_CLASSIC_HEADS)                 // the sigma-only kernels never reach the heads: sn_api.hip routes them to the main pass
  if (store || sigma_only) return -4;
  if (input_mode == 0) SN_LAUNCH(false, 0, false); else SN_LAUNCH(false, 1, false);
#else
  if (store) return -4;
  if (input_mode == 0) { if (sigma_only) SN_LAUNCH(true, 0, false); else SN_LAUNCH(false, 0, false); }
  else { if (sigma_only) SN_LAUNCH(true, 1, false); else SN_LAUNCH(false, 1, false); }
#endif
#undef SN_LAUNCH
  return (int)hipGetLastError();
}
