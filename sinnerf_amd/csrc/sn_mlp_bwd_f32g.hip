// sn_mlp_bwd_f32g.hip -- backward "chain" of the fused NeRF MLP for gfx950 (fp32), second generation (round 6).
//
// Computes exactly what sn_mlp_bwd.hip computes -- g_x = W^T g_y, g_y = g_h (.) act'(.) for every layer, what torch autograd derives from
// models/nerf.py:122-148 (+ models/activations.py); writes the per-layer pre-activation gradients G[10][slot_rows][256] and g_out -- with
// the same MFMA order and the same VALU arithmetic (bit-identical G / g_out: tests/test_f32_kernels_gpu.py), on the data flow of
// sn_mlp_fwd_f32g.hip (its header says why: next to the f32-input MFMA only VALU instructions cost matrix time):
//   * transposed-weight A fragments straight from L2 into a register ring of 16 groups (the activation-tile loads below come from HBM and
//     share the vector-memory counter with the fragment loads: the ring has to outlast them); no LDS ring, no LDS-DMA, no barrier;
//   * the derivative mask of an output tile -- v = (h > 0) ? x : 0, 16 compare + select pairs, plus the sigma head's 16 FMAs in
//     xyz_encoding_final^T -- is ONE VALU gap per slab (9.6 + 4 n cycles) instead of four; everything else of the epilogue is single
//     LDS / memory instructions dealt one per MFMA gap: the masked values go through the wave's [point][feature] staging tile into the
//     AGPRs of the next transposed layer (ds_read_b128 with an AGPR destination) and out to G as whole 128-byte rows;
//   * the forward activation tile a mask needs is requested a whole slab (8 192 cycles) ahead of the gap that uses it (four loads in the
//     accumulator layout, each lane its own 16 bytes, as in the round-2 kernel);
//   * epilogues are deferred across layer boundaries (the next transposed layer reads a tile's K-slots only from group 4 t on).
// The rgb.0^T / heads prologue of a point tile is the round-2 code (VALU work by nature).
#include "sn_mlp_f32g.h"

namespace snk {

constexpr int BWDG_TAIL_BYTES = snl::B_TAIL_FLOATS * 4;                     // 2816: zero "bias" slot + aux table
constexpr int BWDG_LDS_BYTES = BWDG_TAIL_BYTES + XPOSE_LDS_BYTES;           // + the staging tiles of the row stores
constexpr int FD_CHAIN = 16;
constexpr int CHAIN_STEPS = 24;                                             // 0: mask VALU; 1..4 quads; 5..8 AGPR loads; 9..14 rows; 16..19 / 20: own loads

// stream slab s (0..71) of a point tile: 8 x dir_encoding^T (K = 128), 8 x xyz_encoding_final^T, then xyz_encoding_{li+1}^T for li = 7..1
constexpr int chain_ng(int s) { return s < 8 ? 16 : 32; }
constexpr int chain_li(int s) { return 7 - (s - 16) / 8; }                                  // s >= 16
constexpr int chain_set(int s) { return s < 8 ? 0 : s < 16 ? 1 : ((chain_li(s) & 1) ? 0 : 1); }      // B operands: AGPR set read
constexpr int chain_w(int s) { return s < 8 ? 1 : s < 16 ? 0 : ((chain_li(s) & 1) ? 1 : 0); }        // ... set written by its epilogue
constexpr int chain_slot(int s) { return s < 8 ? 8 : s < 16 ? 7 : chain_li(s) - 1; }                 // G slot written (= acts slot of the mask)
constexpr int chain_kind(int s) { return s < 0 ? -1 : s < 8 ? 0 : s < 16 ? 2 : 1; }                  // 0 copy, 1 mask, 2 mask + sigma term
constexpr unsigned chain_byte_offset(int s) { return (unsigned)(snl::bslab_elem_offset(s) * 4); }
constexpr int chain_g0(int s) { return (int)(chain_byte_offset(s) / 1024); }
constexpr int chain_prev(int s) { return s < 1 ? 0 : s - 1; }                                        // descriptor index of the previous slab (slab 0: none, kind -1)

// acts / G slots: 0..7 = h1..h8 (resp. g_y of xyz_encoding_1..8), 8 = final, 9 = h2 / g_y2 (128 wide, ld 256)
__global__ void __launch_bounds__(256)
mlp_bwd_chain_f32g_kernel(const char* __restrict__ bblob, const float* __restrict__ acts, const float* __restrict__ out_raw,
                          const float* __restrict__ g_raw, long P, long slot_rows, float* __restrict__ G,
                          float* __restrict__ g_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_zero = reinterpret_cast<float*>(smem);                       // the slab loop's "bias" slot: all zero
  const float* lds_aux = lds_zero + snl::B_ZERO_FLOATS;
  asm volatile("" ::: "a0", "a255");             // size the kernel for the whole hand-managed AGPR file (sn_mlp_pipe.h)
  constexpr int FD = FD_CHAIN;
  constexpr int TOT = chain_g0(snl::NB_SLABS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  const long n_tiles = (P + 127) / 128;
  static_assert(TOT % FD == 0, "ring index = stream index mod FD");

  {
    const float4* gb = reinterpret_cast<const float4*>(bblob + snl::b_tail_byte_offset());
    float4* lb = reinterpret_cast<float4*>(lds_zero);
    for (int i = tid; i < snl::B_TAIL_FLOATS / 4; i += 256) lb[i] = gb[i];
  }
  __syncthreads();                               // zero / aux table visible; the ONLY barrier of the kernel

  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(bblob), 0, (int)snl::bblob_bytes(), 0x00020000);
  unsigned voff = lane * 16;
  asm volatile("" : "+v"(voff));
  f32x4 fr[FD];
#pragma unroll
  for (int g = 0; g < FD - 1; ++g) fr[g] = load_frag(rs, voff, g * 1024);      // slab 0 .., groups 0 .. FD-2 of the stream
  f32x16 acc0 = load_bias(lds_zero, 0, h), acc1;
  // per-wave [point][feature] staging tile of the g_y row stores (sn_mlp_pipe.h XPOSE_*)
  char* const xp = smem + BWDG_TAIL_BYTES + wave * XPOSE_WAVE_BYTES;
  const unsigned xp_w = (unsigned)(j * XPOSE_PITCH + 4 * h) * 4u;                       // this lane's register quads
  const unsigned xp_r = (unsigned)((lane >> 3) * XPOSE_PITCH + 4 * (lane & 7)) * 4u;    // row lane>>3, 16-byte chunk lane&7
  unsigned g_off = (unsigned)((lane >> 3) * 256 + 4 * (lane & 7)) * 4u;                // row stores: row lane>>3, 16-byte chunk lane&7
  unsigned a_off = (unsigned)(j * 256 + 4 * h) * 4u;                                   // activation tile, accumulator layout
  unsigned epi_a = (unsigned)(size_t)xp + xp_w;
  asm volatile("" : "+v"(g_off), "+v"(epi_a), "+v"(a_off));                           // opaque ONCE: three live registers, no per-use copies

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

    // row group i (8 points x 128 B) of the staged 32-point x 32-feature tile -> G[slot][point][32t .. 32t+31], non-temporal (5 GB of
    // write-once data must not evict the L2-resident weights).  Rows are allocated for whole 128-point tiles: no predicate.
    auto row_read = [&](int i) __attribute__((always_inline)) {
      return *reinterpret_cast<const f32x4*>(xp + xp_r + 8 * i * XPOSE_PITCH * 4);
    };
    auto row_store = [&](int slot, int t, int i, const f32x4& o) __attribute__((always_inline)) {
      const char* base = reinterpret_cast<const char*>(G) + (((long)slot * slot_rows + p_wave + 8 * i) * 256 + 32 * t) * 4;
      // (wave-uniform 64-bit base + a zero-extended 32-bit VGPR offset: hipcc selects the saddr form, no VALU.  The round-2 kernel made the
      // offset opaque per store -- a v_mov each, i.e. a VALU gap of 13.6 cycles per store next to the f32-input MFMA)
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(const_cast<char*>(base) + g_off));
    };
    f32x4 rowbuf[2];
    auto row_step = [&](int slot, int t, int k) __attribute__((always_inline)) {      // two reads ahead of four (store, read) steps
      if (k == 0) rowbuf[0] = row_read(0);
      else if (k == 1) rowbuf[1] = row_read(1);
      else if (k == 2) { row_store(slot, t, 0, rowbuf[0]); rowbuf[0] = row_read(2); }
      else if (k == 3) { row_store(slot, t, 1, rowbuf[1]); rowbuf[1] = row_read(3); }
      else if (k == 4) row_store(slot, t, 2, rowbuf[0]);
      else if (k == 5) row_store(slot, t, 3, rowbuf[1]);
    };
    // forward activation tile for the derivative mask of output tile t, accumulator layout (quad i of 4)
    f32x4 av[4] = {};
    auto load_act = [&](int slot, int t, int i) __attribute__((always_inline)) {
      const char* base = reinterpret_cast<const char*>(acts) + (((long)slot * slot_rows + p_wave) * 256 + 32 * t + 8 * i) * 4;
      av[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + a_off));
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
        f32x4 o;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
        *reinterpret_cast<f32x4*>(xp + xp_w + 32 * q) = o;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) row_store(9, t, i, row_read(i));
    }

    auto quad = [](const f32x16& r, int q) __attribute__((always_inline)) {
      f32x4 x;
      x[0] = r[4 * q]; x[1] = r[4 * q + 1]; x[2] = r[4 * q + 2]; x[3] = r[4 * q + 3];
      return x;
    };
    float sv[16];                                // the masked values of a tile between the VALU step and their LDS writes
    f32x4 sw[4];                                 // xyz_encoding_final^T: the sigma head's weights of a tile (nerf.py:136)
    // epilogue PROGRAM of the slab that just finished (kind 0: g_final, no activation; 1: g_y = g_h [h > 0]; 2: ... with the sigma head's
    // term w_sigma[f] g_sigma added first), result r -> K-slots 16t.. of set W, rows of G[slot]; one instruction per step
    auto prev_prog = [&](auto kind_c, auto w_c, auto slot_c, auto t_c, int st, const f32x16& r) __attribute__((always_inline)) {
      constexpr int KIND = decltype(kind_c)::value, W = decltype(w_c)::value, SLOT = decltype(slot_c)::value, T = decltype(t_c)::value;
      if constexpr (KIND == 0) {
        if (st < 4) lds_put_quad(epi_a, 32 * st, quad(r, st));
        else if (st < 8) lds_get_quad_agpr(W * 128 + 16 * T + 4 * (st - 4), epi_a, 32 * (st - 4));
        else if (st < 14) row_step(SLOT, T, st - 8);
      } else if constexpr (KIND > 0) {
        if (st == 0) {                           // the ONE VALU gap of the slab
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float x = r[i];
            if (KIND == 2) x = __builtin_fmaf(sw[i / 4][i % 4], gsig, x);
            sv[i] = av[i / 4][i % 4] > 0.0f ? x : 0.0f;
          }
        } else if (st < 5) {
          f32x4 x;
          x[0] = sv[4 * (st - 1)]; x[1] = sv[4 * (st - 1) + 1]; x[2] = sv[4 * (st - 1) + 2]; x[3] = sv[4 * (st - 1) + 3];
          lds_put_quad(epi_a, 32 * (st - 1), x);
        } else if (st < 9) {
          lds_get_quad_agpr(W * 128 + 16 * T + 4 * (st - 5), epi_a, 32 * (st - 5));
        } else if (st < 15) {
          row_step(SLOT, T, st - 9);
        }
      }
    };
    // ... and what the RUNNING slab requests for its own epilogue (which runs one slab later): the activation tile of its mask, steps 16..19
    // (behind the step that consumed the previous tile's), and the sigma weights of its tile, step 20
    auto own_prog = [&](auto kind_c, auto slot_c, auto t_c, int st) __attribute__((always_inline)) {
      constexpr int KIND = decltype(kind_c)::value, SLOT = decltype(slot_c)::value, T = decltype(t_c)::value;
      if constexpr (KIND >= 1) {
        if (st >= 16 && st < 20) load_act(SLOT, T, st - 16);
        if (KIND == 2 && st == 20) {
#pragma unroll
          for (int i = 0; i < 4; ++i) sw[i] = *reinterpret_cast<const f32x4*>(lds_aux + snl::B_AUX_SIGT + h * 128 + 16 * T + 4 * i);
        }
      }
    };

#define SNC(V_) std::integral_constant<int, V_>{}
    // stream slab S_ (literal).  Even slabs accumulate in acc0, odd ones in acc1; the previous slab's result sits in the other set.
#define SNC_SLAB(S_)                                                                                                         \
  do {                                                                                                                       \
    constexpr int ps = (S_) - 1;                                                                                             \
    if (((S_) & 1) == 0)                                                                                                     \
      slab_f32g<chain_ng(S_), 0, chain_set(S_), chain_set(S_), chain_g0(S_), TOT, CHAIN_STEPS, FD>(                          \
          acc0, acc1, fr, rs, voff, static_cast<const float*>(nullptr), lds_zero, 0, h, [&](int st) __attribute__((always_inline)) {  \
            prev_prog(SNC(chain_kind(ps)), SNC(chain_w(chain_prev(S_))), SNC(chain_slot(chain_prev(S_))), SNC(chain_prev(S_) % 8), st, acc1);  \
            own_prog(SNC(chain_kind(S_)), SNC(chain_slot(S_)), SNC((S_) % 8), st); });                                       \
    else                                                                                                                     \
      slab_f32g<chain_ng(S_), 0, chain_set(S_), chain_set(S_), chain_g0(S_), TOT, CHAIN_STEPS, FD>(                          \
          acc1, acc0, fr, rs, voff, static_cast<const float*>(nullptr), lds_zero, 0, h, [&](int st) __attribute__((always_inline)) {  \
            prev_prog(SNC(chain_kind(ps)), SNC(chain_w(chain_prev(S_))), SNC(chain_slot(chain_prev(S_))), SNC(chain_prev(S_) % 8), st, acc0);  \
            own_prog(SNC(chain_kind(S_)), SNC(chain_slot(S_)), SNC((S_) % 8), st); });                                       \
  } while (0)
#define SNC_SLAB8(B_) SNC_SLAB((B_) + 0); SNC_SLAB((B_) + 1); SNC_SLAB((B_) + 2); SNC_SLAB((B_) + 3); \
                      SNC_SLAB((B_) + 4); SNC_SLAB((B_) + 5); SNC_SLAB((B_) + 6); SNC_SLAB((B_) + 7)
    SNC_SLAB8(0);                                // dir_encoding.0^T (first 256 inputs): g_final = W_d[:, :256]^T g_y2; set 0 -> set 1
    SNC_SLAB8(8);                                // xyz_encoding_final^T (+ sigma^T on the VALU): g_y8; set 1 -> set 0
    SNC_SLAB8(16);                               // xyz_encoding_8^T .. xyz_encoding_2^T: g_y7 .. g_y1
    SNC_SLAB8(24);
    SNC_SLAB8(32);
    SNC_SLAB8(40);
    SNC_SLAB8(48);
    SNC_SLAB8(56);
    SNC_SLAB8(64);
    mfma32_result_fence(acc1);                   // slab 71's result (odd: acc1): the one epilogue of the tile that is not deferred
#pragma unroll
    for (int st = 0; st < 15; ++st) prev_prog(SNC(chain_kind(71)), SNC(chain_w(71)), SNC(chain_slot(71)), SNC(7), st, acc1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef SNC
#undef SNC_SLAB
#undef SNC_SLAB8
  }  // persistent tile loop
}

}  // namespace snk

extern "C" int SN_LAUNCH_NAME(sn_mlp_backward_chain_f32g)(const void* bblob, const float* acts, const float* out_raw,
                                                          const float* g_raw, long n_points, long slot_rows, float* G,
                                                          float* g_out, hipStream_t stream) {
  using namespace snk;
  if (n_points <= 0) return 0;
  if (slot_rows < (n_points + 127) / 128 * 128) return -1;      // whole 128-point tiles of G are written
  const long tiles = (n_points + 127) / 128;
  if (tiles > 0x7fffffffL) return -2;
  const int n_cu = snh::cu_count();              // persistent: one workgroup per CU (~430 registers per lane: one wave per SIMD)
  auto kfn = mlp_bwd_chain_f32g_kernel;
  SN_ENSURE_DYN_LDS(kfn, BWDG_LDS_BYTES);
  hipLaunchKernelGGL(kfn, dim3((unsigned)(tiles < n_cu ? tiles : n_cu)), dim3(256), BWDG_LDS_BYTES, stream,
                     reinterpret_cast<const char*>(bblob), acts, out_raw, g_raw, n_points, slot_rows, G, g_out);
  return (int)hipGetLastError();
}
