// sn_dw_f32.hip -- the fp32 256x256 weight-gradient contractions  dW[m, n] = sum_p G[p, m] X[p, n]  (+ db[m] = sum_p G[p, m])
// with a HAND-SCHEDULED inner loop: the eight 256-wide nn.Linear layers of a NeRF (models/nerf.py:66-76), 85 % of the
// weight-gradient work of an fp32 training step.  Same tasks, tiles, ring and results as variant 0 of sn_dw.hip (one
// workgroup = one K-range of one problem; 2x2 waves x 128x128 accumulator blocks; row-major 16-point chunks staged by LDS-DMA
// through a 4-deep ring; partials summed afterwards by dw_finish_kernel) -- what changes is who lays out the instruction
// stream of a chunk: tools/gen_dw_f32.py (one asm statement per chunk: MFMAs back to back, the next pair's fragment reads,
// the DMA pieces and the bias sums dealt into their shadow).  The compiler-scheduled loop sits out an LDS round trip per
// 16 MFMAs (117 of 157 TF); see the generator for the plan.
//
// The 256 accumulators of a wave ARE the AGPR file for the whole task, across the per-chunk statements and the C++ glue
// between them: tools/check_agpr.py verifies on the generated code that the compiler allocated no AGPR and spilled nothing.
#include "sn_dw_common.h"
#include "sn_dw_f32_chunk.inc"

namespace snd {

constexpr int F32_A_BYTES = KB * 256 * 4;        // 16384: A tile of a chunk; the B tile follows
constexpr int F32_BUF = 2 * F32_A_BYTES;         // 32768 per ring slot
constexpr int F32_NBUF = 4;
static_assert(F32_NBUF * F32_BUF == DW_LDS_BYTES, "ring = the LDS allocation of the weight-gradient kernels");

template <int R>
SN_DEV float acc_read() {                        // accumulator register R of this lane (a[R])
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R));
  return x;
}
template <int A, int B, int R>
SN_DEV void store_block(const Task& t, int m0, int n0, int i, int h) {
  if constexpr (R < 16) {
    const int m = m0 + 32 * A + (R & 3) + 8 * (R >> 2) + 4 * h;
    t.c[(long)m * t.ldc + n0 + 32 * B + i] = acc_read<16 * (4 * A + B) + R>();
    store_block<A, B, R + 1>(t, m0, n0, i, h);
  }
}
template <int AB>
SN_DEV void store_all(const Task& t, int m0, int n0, int i, int h) {
  if constexpr (AB < 16) {
    store_block<AB / 4, AB % 4, 0>(t, m0, n0, i, h);
    store_all<AB + 1>(t, m0, n0, i, h);
  }
}

__global__ void __launch_bounds__(256) dw_f32_asm_kernel(const Plan plan) {
  asm volatile("" ::: "a0", "a255");             // size the kernel for the whole hand-managed AGPR file
  const Task t = task_of(plan, (int)blockIdx.x);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = wr * 128, n0 = wc * 128;
  const long k0 = t.k0, k1 = t.k1;
  if (k0 >= k1) return;
  const int n_chunks = (int)((k1 - k0 + KB - 1) / KB);
  // per-thread global byte offsets of the 4 + 4 DMA pieces of a chunk (16 rows x 1 KB per tile; piece = 256 threads x 16 B)
  unsigned oa[4], ob[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = it * 256 + tid;
    oa[it] = (unsigned)((c >> 6) * t.lda * 4 + (c & 63) * 16);
    ob[it] = (unsigned)((c >> 6) * t.ldb * 4 + (c & 63) * 16);
  }
  const char* ga0 = reinterpret_cast<const char*>(t.a);
  const char* gb0 = reinterpret_cast<const char*>(t.b);
  auto chunk_base = [&](const char* g, int ld, long k) __attribute__((always_inline)) {      // wave-uniform; a chunk past the
    const long kc = k < k1 ? k : k1 - KB;                                                    // end re-reads the last one
    return g + kc * ld * 4;
  };
  // prologue: three chunks in flight (dynamic LDS starts at address 0: the kernel has no static __shared__)
#pragma unroll
  for (int c = 0; c < F32_NBUF - 1; ++c) {
    const char* ba = chunk_base(ga0, t.lda, k0 + (long)c * KB);
    const char* bb = chunk_base(gb0, t.ldb, k0 + (long)c * KB);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)(ba + oa[it]), (lds_void*)(size_t)(c * F32_BUF + it * 4096 + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)(bb + ob[it]), (lds_void*)(size_t)(c * F32_BUF + F32_A_BYTES + it * 4096 + wave * 1024), 16, 0, 0);
    }
  }
  asm volatile(SN_DWF32_ZERO_ASM ::: SN_DWF32_AGPR_CLOBBERS);
  float bs0 = 0.0f, bs1 = 0.0f, bs2 = 0.0f, bs3 = 0.0f;
  // LDS read addresses of this lane inside a slot: A tile row = point (1 KB), this lane: point parity h, feature m0 + i
  const unsigned la = (unsigned)(h * 1024 + (m0 + i) * 4);
  const unsigned lb = (unsigned)(F32_A_BYTES + h * 1024 + (n0 + i) * 4);
#pragma unroll 1
  for (int c = 0; c < n_chunks; ++c) {
    const unsigned slot = (unsigned)(c % F32_NBUF) * F32_BUF;
    const unsigned la0 = la + slot, la1 = la + slot + 128, lb0 = lb + slot, lb1 = lb + slot + 128;
    const long kn = k0 + (long)(c + F32_NBUF - 1) * KB;                     // the chunk staged while this one is consumed
    const char* ga = chunk_base(ga0, t.lda, kn);
    const char* gb = chunk_base(gb0, t.ldb, kn);
    const unsigned md = (unsigned)((c + F32_NBUF - 1) % F32_NBUF) * F32_BUF + (unsigned)wave * 1024u;     // = slot of chunk c-1
    asm volatile(SN_DWF32_CHUNK_ASM
                 : [bs0] "+v"(bs0), [bs1] "+v"(bs1), [bs2] "+v"(bs2), [bs3] "+v"(bs3)
                 : [la0] "v"(la0), [la1] "v"(la1), [lb0] "v"(lb0), [lb1] "v"(lb1),
                   [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [oa2] "v"(oa[2]), [oa3] "v"(oa[3]),
                   [ob0] "v"(ob[0]), [ob1] "v"(ob[1]), [ob2] "v"(ob[2]), [ob3] "v"(ob[3]),
                   [ga] "s"(ga), [gb] "s"(gb), [md] "s"(md)
                 : SN_DWF32_FRAG_CLOBBERS, SN_DWF32_AGPR_CLOBBERS, "memory", "scc");
  }
  // drain the over-issued tail chunks before the LDS is released; MFMA (16 passes) -> accumulator read: 18 wait states
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
  store_all<0>(t, m0, n0, i, h);
  if (t.bias != nullptr && wc == 0) {
    const float b[4] = {bs0, bs1, bs2, bs3};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float v = b[a] + __shfl_xor(b[a], 32, 64);
      if (h == 0) t.bias[m0 + 32 * a + i] = v;
    }
  }
}

}  // namespace snd

extern "C" int sn_dw_f32_asm_launch(const snd::Plan* plan_host, hipStream_t stream) {
  using namespace snd;
  if (plan_host->n_tasks <= 0) return 0;
  SN_ENSURE_DYN_LDS(dw_f32_asm_kernel, DW_LDS_BYTES);
  hipLaunchKernelGGL(dw_f32_asm_kernel, dim3((unsigned)plan_host->n_tasks), dim3(256), DW_LDS_BYTES, stream, *plan_host);
  return (int)hipGetLastError();
}
