// sn_mlp_fwd_bf16.hip -- fused NeRF MLP forward for gfx950, bf16-operand / fp32-accumulate path
// (v_mfma_f32_32x32x16_bf16: 16x the fp32 MFMA rate; BASELINE configs 3 and 5, north_star "MFMA bf16 for the dense
// W.x contractions").  Same algorithm and reference lines as sn_mlp_fwd.hip; what changes with the data type:
//
// * a wave owns TWO 32-point column tiles (PT = 2): every A fragment read from LDS (one ds_read_b128 = 8 bf16 of 32
//   weight rows) feeds two MFMAs on two independent accumulators;
// * activations live in registers as packed bf16x8 B operands: the fp32 accumulators of an output tile are biased,
//   activated and converted (v_cvt_pk_bf16_f32, RNE) straight into the two k-steps they form for the next layer
//   (K-slot order of sn_layout.h: k-step 2t holds accumulator registers 0..7 of tile t, k-step 2t+1 registers 8..15);
// * the two narrow heads never touch bf16: sigma (nerf.py:136) is accumulated on the VALU in fp32 from the ReLU'd fp32
//   accumulators of layer 8 while they are being packed, rgb (nerf.py:144) likewise from the fp32 ShiftedSoftplus
//   outputs of dir_encoding -- so the 128-wide dir activation is never materialised at all;
// * weights stream as bf16 slabs (K*64 bytes, a quarter of the fp32 LDS traffic per point), biases stay fp32.
//
// Persistent workgroups, 3-slot weight ring with a mid-slab barrier: sn_mlp_pipe.h.
#include "sn_mlp_pipe.h"

namespace snk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int PT = 2;                                       // point tiles per wave
constexpr int RING_SLOT_BYTES_BF16 = snl::MAX_SLAB_K * 64;  // 20480
constexpr int MLP_BF16_LDS_BYTES = TAIL_LDS_BYTES + 3 * RING_SLOT_BYTES_BF16;   // 73984
typedef RingT<64, RING_SLOT_BYTES_BF16> RingB;

// The layer being WRITTEN lives in the accumulator half of the register file (explicit v_accvgpr_write, "a" constraint)
// so that the layer being READ -- the MFMA B operands -- can stay in architectural VGPRs: hipcc otherwise parks half of
// the 256 activation registers in AGPRs as spill slots and pays 4 v_accvgpr_read per MFMA (measured: VALU-issue-bound).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
SN_DEV uint32_t pack2(float a, float b) {
  bf16x2 v;
  v[0] = (__bf16)a; v[1] = (__bf16)b;            // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(uint32_t, v);
}
SN_DEV uint32_t to_agpr(uint32_t x) {
  uint32_t a;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(x));
  return a;
}
SN_DEV uint32_t from_agpr(uint32_t a) {
  uint32_t x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a));
  return x;
}

SN_DEV bf16x8 pack8(const float* v) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
  asm volatile("" : "+v"(o));                    // pin: convert here, not lazily at the consumer (register pressure)
  return o;
}

// One slab: NK0 + NK1 k-steps (two K segments, B operands b0 / b1 laid out [k-step][PT]), barrier after k-step GB.
//   af       4-entry ring of A fragments, prefetch distance 3 k-steps (a bf16 k-step is only 2 x 32 MFMA cycles, one step
//            of lookahead does not cover the LDS latency).  Invariant at entry: fragments of k-steps 0,1,2 of this slab
//            sit in af[(PHASE+0..2) & 3]; at exit the same holds for the next slab with PHASE' = (PHASE + NK) & 3
//            (NK % 4 == 0 everywhere except the four dir_encoding slabs, whose phases 0,2,0,2 are still static).
template <int NK0, int NK1, int GB, int PHASE, class Pending>
SN_DEV void slab_bf16(f32x16 (&acc)[PT], bf16x8 (&af)[4], f32x16& acc_pre, const char* lw, const bf16x8* b0,
                      const bf16x8* b1, const char* lw_next, const float* lds_bias, int s_next, int h, RingB& ring,
                      Pending&& pending) {
  constexpr int NK = NK0 + NK1;
  constexpr int PPG = (5 + (NK - GB) - 1) / (NK - GB);      // <= 5 pieces of 4 KB per slab (K = 320)
  static_assert(GB >= 1 && GB < NK && NK >= 4, "sync point inside the slab");
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    {
      const int kn = ks + 3;
      af[(PHASE + kn) & 3] = (kn < NK) ? *reinterpret_cast<const bf16x8*>(lw + kn * 1024)
                                       : *reinterpret_cast<const bf16x8*>(lw_next + (kn - NK) * 1024);
    }
    if (ks == GB) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      ring.begin_stage();
      acc_pre = load_bias(lds_bias, s_next, h);
    }
    if (ks >= GB && ks < GB + (5 + PPG - 1) / PPG) {          // <= 5 pieces per slab: no dead issue sites after them
#pragma unroll
      for (int j = 0; j < PPG; ++j) ring.issue_piece();
    }
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8* b = (ks < NK0) ? (b0 + ks * PT) : (b1 + (ks - NK0) * PT);
    const bf16x8 a_cur = af[(PHASE + ks) & 3];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur, b[pt], acc[pt], 0, 0, 0);
    if (ks == 0) pending();
  }
  ring.end_stage();
}

template <bool SIGMA_ONLY, int INPUT_MODE>
__global__ void __launch_bounds__(256)
mlp_fwd_bf16_kernel(const char* __restrict__ blob, const float* __restrict__ in0, const float* __restrict__ in1,
                    long P, int S, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_bias = reinterpret_cast<float*>(smem);
  const float* lds_aux = lds_bias + snl::BIAS_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  constexpr int TILE_PTS = 4 * PT * 32;                      // 256 points per workgroup pass
  const long n_tiles = (P + TILE_PTS - 1) / TILE_PTS;
  const long my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;

  RingB ring;
  ring.blob = blob;
  ring.gnext = blob;
  ring.base = smem + TAIL_LDS_BYTES;
  ring.n_used = SIGMA_ONLY ? snl::SLAB_FIN : snl::N_SLABS;
  ring.stage_id = 0;
  ring.stage_slot = 0;
  ring.remaining = my_tiles * ring.n_used;
  ring.tid = tid;
  ring.wbase = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
  ring.pieces = 0; ring.piece = 0; ring.slab_bytes = 0;
  ring.stage_whole();
  ring.stage_whole();
  {
    const float4* gb = reinterpret_cast<const float4*>(blob + snl::bias_byte_offset(snl::DT_BF16));
    float4* lb = reinterpret_cast<float4*>(lds_bias);
    for (int i = tid; i < snl::TAIL_FLOATS / 4; i += 256) lb[i] = gb[i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cslot = 0;
  bf16x8 af[4];
#pragma unroll
  for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ring.slot(0) + lane * 16 + i * 1024);
  f32x16 acc_pre = load_bias(lds_bias, 0, h);
  const int n_used = ring.n_used;

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    long p_raw[PT], p[PT];
    bool valid[PT];
    bf16x8 xe[4 * PT];                                       // [k-step][PT]
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      p_raw[pt] = ((tile * 4 + wave) * PT + pt) * 32 + j;
      valid[pt] = p_raw[pt] < P;
      p[pt] = valid[pt] ? p_raw[pt] : P - 1;
      float f[32];
      if (INPUT_MODE == 0) {
        const float* rp = in0 + (p[pt] / S) * 8;
        const float zz = in1[p[pt]];
        const float x = __fadd_rn(rp[0], __fmul_rn(rp[3], zz));
        const float y = __fadd_rn(rp[1], __fmul_rn(rp[4], zz));
        const float z = __fadd_rn(rp[2], __fmul_rn(rp[5], zz));
        embed_xyz(x, y, z, h, f);
      } else {
        const float* row = in0 + p[pt] * (long)S;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
          const int c0 = snl::xyz_slot_col(0, e), c1 = snl::xyz_slot_col(1, e);
          const int c = h ? c1 : c0;
          f[e] = (c >= 0) ? row[c < 0 ? 0 : c] : 0.0f;
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xe[ks * PT + pt] = pack8(f + 8 * ks);
    }

    int s = 0;
    bf16x8 hid[16 * PT];                                     // [k-step][PT]: layer being read (VGPR tuples)
    uint32_t nxt[16 * PT * 4];                               // layer being written: dwords (2 bf16), AGPR-resident
    f32x16 acc[PT], pacc[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[pt] = acc_pre;
    float sg[PT];                                            // sigma head partial (fp32, this lane half)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) sg[pt] = 0.0f;

    // epilogue of output tile t of a ReLU layer: relu, (optionally) sigma partial, pack to the two k-steps 2t, 2t+1
    auto relu_tile = [&](int t, bool with_sigma) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = relu1(pacc[pt][r]);
        if (with_sigma) {
          const f32x4* ws = reinterpret_cast<const f32x4*>(lds_aux + snl::AUX_SIGW + h * 128 + 16 * t);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w = ws[q];
            sg[pt] = __builtin_fmaf(w[0], v[4 * q + 0], sg[pt]);
            sg[pt] = __builtin_fmaf(w[1], v[4 * q + 1], sg[pt]);
            sg[pt] = __builtin_fmaf(w[2], v[4 * q + 2], sg[pt]);
            sg[pt] = __builtin_fmaf(w[3], v[4 * q + 3], sg[pt]);
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)                          // dword q of the tile = accumulator registers 2q, 2q+1
          nxt[((2 * t + (q >> 2)) * PT + pt) * 4 + (q & 3)] = to_agpr(pack2(v[2 * q], v[2 * q + 1]));
      }
    };
    auto promote = [&]() {                                   // layer boundary: written layer -> read layer (AGPR -> VGPR)
#pragma unroll
      for (int i = 0; i < 16 * PT; ++i) {
        u32x4_t q;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) q[jj] = from_agpr(nxt[i * 4 + jj]);
        hid[i] = __builtin_bit_cast(bf16x8, q);
      }
    };
#define SNB_LW_CUR (ring.slot(cslot) + lane * 16)
#define SNB_LW_NEXT (ring.slot(cslot == 2 ? 0 : cslot + 1) + lane * 16)
#define SNB_SNEXT (s + 1 == n_used ? 0 : s + 1)
#define SNB_ADVANCE()                                                    \
  do {                                                                   \
    _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) { pacc[pt] = acc[pt]; acc[pt] = acc_pre; } \
    ++s; cslot = (cslot == 2) ? 0 : cslot + 1;                           \
  } while (0)

    // ---- layer 0
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      slab_bf16<4, 0, 1, 0>(acc, af, acc_pre, SNB_LW_CUR, xe, xe, SNB_LW_NEXT, lds_bias, SNB_SNEXT, h, ring,
                         [&] { if (t > 0) relu_tile(t - 1, false); });
      SNB_ADVANCE();
    }
    relu_tile(7, false);
    promote();

    // ---- layers 1..7 (skip concat at layer 4); layer 7's epilogues also feed the sigma head
#pragma unroll 1
    for (int l = 1; l < 8; ++l) {
      const bool ws = (l == 7);
      if (l == 4) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          slab_bf16<4, 16, 2, 0>(acc, af, acc_pre, SNB_LW_CUR, xe, hid, SNB_LW_NEXT, lds_bias, SNB_SNEXT, h, ring,
                              [&] { if (t > 0) relu_tile(t - 1, false); });
          SNB_ADVANCE();
        }
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          slab_bf16<16, 0, 2, 0>(acc, af, acc_pre, SNB_LW_CUR, hid, hid, SNB_LW_NEXT, lds_bias, SNB_SNEXT, h, ring,
                              [&] { if (t > 0) relu_tile(t - 1, ws); });
          SNB_ADVANCE();
        }
      }
      relu_tile(7, ws);
      promote();
    }

    float sigma[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) sigma[pt] = sg[pt] + __shfl_xor(sg[pt], 32, 64) + lds_aux[snl::AUX_HEADB];
    if (SIGMA_ONLY) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        if (valid[pt] && h == 0) out[p_raw[pt]] = sigma[pt];
      continue;
    }

    // ---- xyz_encoding_final (no activation)
    auto copy_tile = [&](int t) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = pacc[pt][r];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          nxt[((2 * t + (q >> 2)) * PT + pt) * 4 + (q & 3)] = to_agpr(pack2(v[2 * q], v[2 * q + 1]));
      }
    };
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      slab_bf16<16, 0, 2, 0>(acc, af, acc_pre, SNB_LW_CUR, hid, hid, SNB_LW_NEXT, lds_bias, SNB_SNEXT, h, ring,
                          [&] { if (t > 0) copy_tile(t - 1); });
      SNB_ADVANCE();
    }
    copy_tile(7);
    promote();

    // ---- dir_encoding + ShiftedSoftplus; the rgb head is accumulated from the fp32 softplus outputs
    bf16x8 de[2 * PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      float f[16];
      if (INPUT_MODE == 0) {
        const float* rp = in0 + (p[pt] / S) * 8;
        embed_dir(rp[3], rp[4], rp[5], h, f);
      } else {
        const float* row = in0 + p[pt] * (long)S;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int c0 = snl::dir_slot_col(0, e), c1 = snl::dir_slot_col(1, e);
          const int c = h ? c1 : c0;
          f[e] = (c >= 0) ? row[63 + (c < 0 ? 0 : c)] : 0.0f;
        }
      }
      de[0 * PT + pt] = pack8(f);
      de[1 * PT + pt] = pack8(f + 8);
    }
    float c3[PT][3];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) c3[pt][0] = c3[pt][1] = c3[pt][2] = 0.0f;
    auto ssp_tile = [&](int t) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = shifted_softplus_fast(pacc[pt][r]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x4* wr = reinterpret_cast<const f32x4*>(lds_aux + snl::AUX_RGBW + c * 128 + h * 64 + 16 * t);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w = wr[q];
            c3[pt][c] = __builtin_fmaf(w[0], v[4 * q + 0], c3[pt][c]);
            c3[pt][c] = __builtin_fmaf(w[1], v[4 * q + 1], c3[pt][c]);
            c3[pt][c] = __builtin_fmaf(w[2], v[4 * q + 2], c3[pt][c]);
            c3[pt][c] = __builtin_fmaf(w[3], v[4 * q + 3], c3[pt][c]);
          }
        }
      }
    };
    // 18 k-steps per slab: the fragment-ring phase alternates 0,2,0,2 (static)
    slab_bf16<16, 2, 2, 0>(acc, af, acc_pre, SNB_LW_CUR, hid, de, SNB_LW_NEXT, lds_bias, SNB_SNEXT, h, ring, [&] {});
    SNB_ADVANCE();
    slab_bf16<16, 2, 2, 2>(acc, af, acc_pre, SNB_LW_CUR, hid, de, SNB_LW_NEXT, lds_bias, SNB_SNEXT, h, ring, [&] { ssp_tile(0); });
    SNB_ADVANCE();
    slab_bf16<16, 2, 2, 0>(acc, af, acc_pre, SNB_LW_CUR, hid, de, SNB_LW_NEXT, lds_bias, SNB_SNEXT, h, ring, [&] { ssp_tile(1); });
    SNB_ADVANCE();
    slab_bf16<16, 2, 2, 2>(acc, af, acc_pre, SNB_LW_CUR, hid, de, SNB_LW_NEXT, lds_bias, SNB_SNEXT, h, ring, [&] { ssp_tile(2); });
    SNB_ADVANCE();
    ssp_tile(3);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      float o3[3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        o3[c] = widened_sigmoid(c3[pt][c] + __shfl_xor(c3[pt][c], 32, 64) + lds_aux[snl::AUX_HEADB + 1 + c]);
      if (valid[pt] && h == 0) {
        float4 o;
        o.x = o3[0]; o.y = o3[1]; o.z = o3[2]; o.w = sigma[pt];
        reinterpret_cast<float4*>(out)[p_raw[pt]] = o;
      }
    }
#undef SNB_LW_CUR
#undef SNB_LW_NEXT
#undef SNB_SNEXT
#undef SNB_ADVANCE
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace snk

extern "C" int sn_mlp_forward_bf16_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                                          int sigma_only, int input_mode, float* out, hipStream_t stream) {
  using namespace snk;
  if (n_points <= 0) return 0;
  const long tiles = (n_points + 255) / 256;
  int dev = 0, n_cu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
  dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu)), block(256);
  const size_t lds = MLP_BF16_LDS_BYTES;
  const char* b = reinterpret_cast<const char*>(blob);
#define SN_LAUNCH(SO, IM)                                                                                        \
  do {                                                                                                           \
    auto kfn = mlp_fwd_bf16_kernel<SO, IM>;                                                                      \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return (int)e;                                                                          \
    hipLaunchKernelGGL(kfn, grid, block, lds, stream, b, in0, in1, n_points, s_or_ld, out);                      \
  } while (0)
  if (input_mode == 0) { if (sigma_only) SN_LAUNCH(true, 0); else SN_LAUNCH(false, 0); }
  else { if (sigma_only) SN_LAUNCH(true, 1); else SN_LAUNCH(false, 1); }
#undef SN_LAUNCH
  return (int)hipGetLastError();
}
