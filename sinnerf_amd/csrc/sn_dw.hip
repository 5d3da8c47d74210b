// sn_dw.hip -- weight-gradient contractions of the NeRF MLP backward for gfx950 (fp32):
//     dW[m, n] = sum_p  G[p, m] * X[p, n]          (+ optionally  db[m] = sum_p G[p, m])
// i.e. the "gW = g_y^T x" terms torch autograd accumulates for every nn.Linear of models/nerf.py:66-103, with the
// contraction running over ALL sample points p (K = 0.25..1 M) and a small M x N <= 256 x 256 result.
//
// G (pre-activation gradients, written by sn_mlp_bwd.hip) and X (activations / embedded inputs, written by the
// training forward) are row-major [P][ld]: a row = one point.  With v_mfma_f32_32x32x2_f32 the A operand is
// A[i][k] = G[p=k][m0+i] and the B operand B[k][j] = X[p=k][n0+j]: both read 32 CONSECUTIVE floats of a row per lane
// half, so the tiles are staged row-major (global_load_lds DMA, lane-linear) and fragment reads are conflict-free
// ds_read_b32 -- no transposes anywhere.
//
// One workgroup = one task = (problem, K-range): 2x2 waves, each wave owns an (MT*32) x (NT*32) block of accumulators
// (MT=NT=4: 256 accumulator registers) and walks its K-range in chunks of 16 points.  G and X stream from HBM exactly
// once per problem, so unlike the L2-resident weight slabs of the MLP kernels the DMA needs depth: a 4-deep LDS ring,
// three chunks in flight, COUNTED s_waitcnt vmcnt(N) + raw s_barrier (a __syncthreads() would drain the queue).
// Partial results go to a per-task slab; the K-split partials are summed afterwards (deterministic, no atomics).
#include "sn_device.h"
#include "sn_launch.h"

namespace snd {

#ifndef SN_DW_CPOL
#define SN_DW_CPOL 2      // nt: G and X are streamed once (measured -3 % in the bandwidth-bound bf16 mode, neutral in fp32)
#endif
constexpr int KB = 16;                          // points per staged chunk
constexpr int NBUF = 4;                         // LDS ring depth (NBUF-1 chunks in flight)
constexpr int DW_LDS_BYTES = NBUF * KB * (256 + 256) * 4;   // 131072

struct Task {                                   // 64 bytes, built on the host (sinnerf_amd/autograd.py)
  const void* a;                                // G  + column offset (fp32, or bf16 with 0x200)
  const void* b;                                // X  + column offset (fp32; bf16 with 0x200 except in variants 1 / 3)
  float* c;                                     // partial dW  [M_wg][ldc]
  float* bias;                                  // partial db  [M_wg] or nullptr
  long k0, k1;                                  // point range: (k1-k0) % 16 == 0, rows [k0,k1) readable (callers zero-pad G)
  int lda, ldb;
  int ldc, variant;                             // M x N: 0 = 256x256, 1 = 256x64, 2 = 128x256, 3 = 128x64, 4 = 32x256, 5 = 32x128;
                                                // | 0x100: bf16 operands (mixed-precision training), fp32 accumulate;
                                                // | 0x200: G and the 256-wide activations are stored as bf16 (the embedded
                                                //   inputs of variants 1 / 3 stay fp32); lda / ldb stay in ELEMENTS
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

// copy KB x W floats (row-major, W*4 bytes per row) global -> LDS.  A chunk past k_end (the ring's prefetch overrun) is
// replaced by the last real chunk of the task ((k1-k0) % KB == 0) -- a wave-uniform select on the chunk base, so the
// per-thread part of the address is a 32-bit byte offset computed once per task (off[it]) and a DMA instruction costs one
// address add instead of a 64-bit multiply + per-row clamp.
template <int W, int ES>                        // ES = element size in bytes (4: fp32 tile, 2: bf16 tile)
struct RowStager {
  static constexpr int CHUNKS = KB * W * ES / 16;   // 16-byte pieces per chunk
  static constexpr int PER_ROW = W * ES / 16;
  static constexpr int IT = (CHUNKS + 255) / 256;
  unsigned off[IT];
  SN_DEV void init(int ld, int tid) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int c = it * 256 + tid;
      off[it] = (unsigned)((c / PER_ROW) * ld * ES + (c % PER_ROW) * 16);
    }
  }
  SN_DEV void stage(const void* __restrict__ g, int ld, long k, long k_end, char* lds, int tid) const {
    const long kc = k < k_end ? k : k_end - KB;
    const char* base = reinterpret_cast<const char*>(g) + kc * ld * ES;        // wave-uniform
    const int wbase = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      if (CHUNKS % 256 == 0 || it * 256 + tid < CHUNKS)
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(base + off[it]), (lds_void*)(lds + it * 4096 + wbase), 16, 0, SN_DW_CPOL);
    }
  }
};

template <int N>
SN_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// WM x WN waves, each wave an (MT*32) x (NT*32) accumulator block.
typedef __bf16 dw_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned dw_u32x4 __attribute__((ext_vector_type(4)));
typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 dw_bf16x2 __attribute__((ext_vector_type(2)));
// {bf16(a), bf16(b)}, RNE: one v_cvt_pk_bf16_f32.  A builtin, NOT inline asm: the MFMAs of this kernel are builtins too and
// the compiler must see the VALU write -> MFMA read dependence to pad it (an asm conversion right in front of the MFMA
// that consumes it returned garbage in the narrow variants).
SN_DEV unsigned dw_pack2(float a, float b) {
  const dw_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dw_bf16x2));
}

// BF16: the same fp32 row-major tiles are staged (nothing changes on the memory side), but a chunk of 16 points is ONE
// k-step of v_mfma_f32_32x32x16_bf16: lane (i, h) gathers its 8 points of feature i from the LDS tile (stride = row pitch,
// conflict-free across lanes), converts them to a bf16x8 fragment (RNE) and keeps the fp32 column sums for the bias
// gradient.  16 MFMAs of 32 cycles per chunk instead of 128 of 64: the kernel becomes HBM-bound.
// MODE 0: fp32 MFMAs.  1: bf16 MFMAs, fp32 tiles.  2: bf16 MFMAs, bf16 A and B tiles.  3: bf16 MFMAs, bf16 A tile, fp32 B tile.
template <int MT, int NT, int WM, int WN, int MODE>
SN_DEV void run_task(const Task& t, char* smem, int tid) {
  constexpr bool BF16 = MODE != 0;
  constexpr int EA = (MODE >= 2) ? 2 : 4, EB = (MODE == 2) ? 2 : 4;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int WA = WM * MT * 32, WB = WN * NT * 32;
  constexpr int A_BYTES = KB * WA * EA, B_BYTES = KB * WB * EB, BUF = A_BYTES + B_BYTES;
  // DMA instructions per thread per chunk.  A 32-wide A tile (variants 4/5) is only 128 16-byte pieces: waves 2,3 issue
  // none of it, so their vmcnt budget is one instruction per chunk smaller (the wait must be exact per wave).
  constexpr int CH_A = KB * WA * EA / 16, CH_B = KB * WB * EB / 16;
  static_assert(CH_B % 256 == 0 && CH_A % 64 == 0, "staging predicates must be wave-uniform");
  constexpr int IT_A = (CH_A + 255) / 256, IT_B = CH_B / 256, PART_A = (CH_A % 256) / 64;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int wr = wave / WN, wc = wave % WN;
  const int m0 = wr * MT * 32, n0 = wc * NT * 32;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  float bsum[MT];
#pragma unroll
  for (int a = 0; a < MT; ++a) bsum[a] = 0.0f;

  const long k0 = t.k0, k1 = t.k1;
  if (k0 >= k1) return;
  RowStager<WA, EA> sa;
  RowStager<WB, EB> sb;
  sa.init(t.lda, tid);
  sb.init(t.ldb, tid);
  const int n_chunks = (int)((k1 - k0 + KB - 1) / KB);
  // prologue: NBUF-1 chunks in flight (chunks past the end are staged as clamped copies and never consumed)
#pragma unroll
  for (int c = 0; c < NBUF - 1; ++c) {
    sa.stage(t.a, t.lda, k0 + (long)c * KB, k1, smem + c * BUF, tid);
    sb.stage(t.b, t.ldb, k0 + (long)c * KB, k1, smem + c * BUF + A_BYTES, tid);
  }
  if (BF16) {
    // One 32x32x16 k-step per 16-point chunk.  The fragment gathers of chunk c are ISSUED in iteration c and consumed
    // (packed, summed, multiplied) in iteration c+1: their LDS latency hides behind the 16 MFMAs of chunk c-1 -- consumed in
    // place hipcc waits on them 19 times per chunk (measured: 2.5 k cycles per chunk against 512 of MFMA work).
    static_assert(KB == 16, "one 32x32x16 k-step per chunk");
    unsigned ra[MT][8], rb[NT][8];                 // raw gathered values: bf16 bits (zero-extended) or fp32 bits
    for (int c = 0; c <= n_chunks; ++c) {
      char* bc = smem + (c % NBUF) * BUF;
      if (c < n_chunks) {
        const long k = k0 + (long)c * KB;
        if (PART_A != 0 && wave >= PART_A) wait_vmcnt<(NBUF - 2) * (IT_A - 1 + IT_B)>();
        else wait_vmcnt<(NBUF - 2) * (IT_A + IT_B)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own gathers of chunk c-1 done before its slot is restaged
        __builtin_amdgcn_s_barrier();             // all waves' pieces of chunk c landed; chunk c-1 fully gathered
        const int slot = (c + NBUF - 1) % NBUF;   // = slot of chunk c-1
        char* bn = smem + slot * BUF;
        sa.stage(t.a, t.lda, k + (long)(NBUF - 1) * KB, k1, bn, tid);
        sb.stage(t.b, t.ldb, k + (long)(NBUF - 1) * KB, k1, bn + A_BYTES, tid);
      }
      if (c > 0) {                                 // chunk c-1: pack, column sums, 16 MFMAs
        dw_bf16x8 af[MT], bf[NT];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          dw_u32x4 q;
          float sum = 0.0f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            if (EA == 2) {
              q[w] = ra[a][2 * w] | (ra[a][2 * w + 1] << 16);
              sum += __builtin_bit_cast(float, ra[a][2 * w] << 16) + __builtin_bit_cast(float, ra[a][2 * w + 1] << 16);
            } else {
              const float v0 = __builtin_bit_cast(float, ra[a][2 * w]), v1 = __builtin_bit_cast(float, ra[a][2 * w + 1]);
              q[w] = dw_pack2(v0, v1);
              sum += v0 + v1;
            }
          }
          bsum[a] += sum;
          af[a] = __builtin_bit_cast(dw_bf16x8, q);
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          dw_u32x4 q;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            if (EB == 2) q[w] = rb[b][2 * w] | (rb[b][2 * w + 1] << 16);
            else q[w] = dw_pack2(__builtin_bit_cast(float, rb[b][2 * w]), __builtin_bit_cast(float, rb[b][2 * w + 1]));
          }
          bf[b] = __builtin_bit_cast(dw_bf16x8, q);
        }
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
      }
      if (c < n_chunks) {                          // gathers of chunk c: lane (i, h) takes rows 8h .. 8h+7 of its feature
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            if (EA == 2) ra[a][jj] = reinterpret_cast<const unsigned short*>(bc)[(8 * h + jj) * WA + m0 + i + 32 * a];
            else ra[a][jj] = reinterpret_cast<const unsigned*>(bc)[(8 * h + jj) * WA + m0 + i + 32 * a];
          }
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            if (EB == 2) rb[b][jj] = reinterpret_cast<const unsigned short*>(bc + A_BYTES)[(8 * h + jj) * WB + n0 + i + 32 * b];
            else rb[b][jj] = reinterpret_cast<const unsigned*>(bc + A_BYTES)[(8 * h + jj) * WB + n0 + i + 32 * b];
          }
      }
    }
  } else {
    for (int c = 0; c < n_chunks; ++c) {
      const long k = k0 + (long)c * KB;
      // chunk c was issued NBUF-1 chunks ago: everything but the (NBUF-2) younger chunks must have landed
      if (PART_A != 0 && wave >= PART_A) wait_vmcnt<(NBUF - 2) * (IT_A - 1 + IT_B)>();
      else wait_vmcnt<(NBUF - 2) * (IT_A + IT_B)>();
      __builtin_amdgcn_s_barrier();               // all waves' pieces of chunk c landed; chunk c-1 fully consumed
      {
        const int slot = (c + NBUF - 1) % NBUF;   // = slot of chunk c-1
        char* bn = smem + slot * BUF;
        sa.stage(t.a, t.lda, k + (long)(NBUF - 1) * KB, k1, bn, tid);
        sb.stage(t.b, t.ldb, k + (long)(NBUF - 1) * KB, k1, bn + A_BYTES, tid);
      }
      char* bc = smem + (c % NBUF) * BUF;
      const float* la = reinterpret_cast<const float*>(bc) + h * WA + m0 + i;
      const float* lb = reinterpret_cast<const float*>(bc + A_BYTES) + h * WB + n0 + i;
#pragma unroll
      for (int s = 0; s < KB / 2; ++s) {
        float av[MT], bv[NT];
#pragma unroll
        for (int a = 0; a < MT; ++a) av[a] = la[(2 * s) * WA + 32 * a];
#pragma unroll
        for (int b = 0; b < NT; ++b) bv[b] = lb[(2 * s) * WB + 32 * b];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          bsum[a] += av[a];
#pragma unroll
          for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
      }
    }
  }
  wait_vmcnt<0>();                               // drain the over-issued tail chunks before the LDS is released
  // epilogue: accumulator (row = (r&3)+8(r>>2)+4h, col = i) -> c[m][n], 128 B per lane-half per store
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        t.c[(long)m * t.ldc + n0 + 32 * b + i] = acc[a][b][r];
      }
  if (t.bias != nullptr && wc == 0) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const float v = bsum[a] + __shfl_xor(bsum[a], 32, 64);
      if (h == 0) t.bias[m0 + 32 * a + i] = v;
    }
  }
}

__global__ void __launch_bounds__(256) dw_kernel(const Task* __restrict__ tasks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Task t = tasks[blockIdx.x];
  const int tid = threadIdx.x;
  if (t.variant & 0x100) {
    const int mode = (t.variant & 0x200) ? 2 : 1;     // 0x200: G and the activations are stored as bf16
#define SN_DW_CASES(MODE_, MODE_EMB_)                                   \
    switch (t.variant & 0xff) {                                         \
      case 0: run_task<4, 4, 2, 2, MODE_>(t, smem, tid); break;         \
      case 1: run_task<4, 1, 2, 2, MODE_EMB_>(t, smem, tid); break;     \
      case 2: run_task<2, 4, 2, 2, MODE_>(t, smem, tid); break;         \
      case 3: run_task<2, 1, 2, 2, MODE_EMB_>(t, smem, tid); break;     \
      case 4: run_task<1, 2, 1, 4, MODE_>(t, smem, tid); break;         \
      default: run_task<1, 1, 1, 4, MODE_>(t, smem, tid); break;        \
    }
    // variants 1 / 3 contract with the embedded inputs, which stay fp32 in every mode
    if (mode == 1) { SN_DW_CASES(1, 1) } else { SN_DW_CASES(2, 3) }
#undef SN_DW_CASES
    return;
  }
  switch (t.variant) {
    case 0: run_task<4, 4, 2, 2, 0>(t, smem, tid); break;
    case 1: run_task<4, 1, 2, 2, 0>(t, smem, tid); break;
    case 2: run_task<2, 4, 2, 2, 0>(t, smem, tid); break;
    case 3: run_task<2, 1, 2, 2, 0>(t, smem, tid); break;
    case 4: run_task<1, 2, 1, 4, 0>(t, smem, tid); break;
    default: run_task<1, 1, 1, 4, 0>(t, smem, tid); break;
  }
}

}  // namespace snd

extern "C" int sn_dw_launch(const void* tasks, int n_tasks, hipStream_t stream) {
  using namespace snd;
  if (n_tasks <= 0) return 0;
  static_assert(sizeof(Task) == 64, "Task must be 64 bytes (host packs it as 8 x int64)");
  SN_ENSURE_DYN_LDS(dw_kernel, DW_LDS_BYTES);
  hipLaunchKernelGGL(dw_kernel, dim3((unsigned)n_tasks), dim3(256), DW_LDS_BYTES, stream,
                     reinterpret_cast<const Task*>(tasks));
  return (int)hipGetLastError();
}
