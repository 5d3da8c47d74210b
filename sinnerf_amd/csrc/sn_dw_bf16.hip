// sn_dw_bf16.hip -- the 256x256 weight-gradient contractions  dW[m, n] = sum_p G[p, m] X[p, n]  (+ db[m] = sum_p G[p, m]) of the
// mixed-precision training step with bf16 training state (SN_DTYPE_BF16_STATE), with a HAND-SCHEDULED inner loop: the eight
// 256-wide nn.Linear layers of a NeRF (models/nerf.py:66-76), 70 % of the bytes the weight-gradient stage streams.  Same tasks,
// tiles, swizzled DMA image, transpose-read fragments and results as variant 0 / MODE 2 of sn_dw.hip (one workgroup = one
// K-range of one problem; 2x2 waves x 128x128 accumulator blocks; row-major 16-point chunks of bf16 G and X staged by LDS-DMA
// through an 8-deep ring; partials summed afterwards by dw_finish_kernel) -- what changes is who lays out the instruction
// stream: tools/gen_dw_bf16.py (one asm statement per PAIR of chunks).  The compiler-scheduled loop is latency-bound at
// ~2300 cycles per chunk (512 of them MFMAs): 4.4 TB/s on a box that copies at 5.2-6.2 TB/s; see the generator for the plan.
//
// The 256 accumulators of a wave ARE the AGPR file for the whole task, across the per-pair statements and the C++ glue
// between them: tools/check_agpr.py verifies on the generated code that the compiler allocated no AGPR and spilled nothing.
#include "sn_dw_common.h"
#include "sn_dw_bf16_chunk.inc"

namespace snd {

constexpr int B16_A_BYTES = KB * 256 * 2;        // 8192: A tile of a chunk (bf16); the B tile follows
constexpr int B16_BUF = 2 * B16_A_BYTES;         // 16384 per ring slot
constexpr int B16_NBUF = 8;
static_assert(B16_NBUF * B16_BUF == DW_LDS_BYTES, "ring = the LDS allocation of the weight-gradient kernels");

template <int R>
SN_DEV float acc_read16() {                      // accumulator register R of this lane (a[R])
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R));
  return x;
}
template <int A, int B, int R>
SN_DEV void store_block16(const Task& t, int m0, int n0, int i, int h) {
  if constexpr (R < 16) {
    const int m = m0 + 32 * A + (R & 3) + 8 * (R >> 2) + 4 * h;
    t.c[(long)m * t.ldc + n0 + 32 * B + i] = acc_read16<16 * (4 * A + B) + R>();
    store_block16<A, B, R + 1>(t, m0, n0, i, h);
  }
}
template <int AB>
SN_DEV void store_all16(const Task& t, int m0, int n0, int i, int h) {
  if constexpr (AB < 16) {
    store_block16<AB / 4, AB % 4, 0>(t, m0, n0, i, h);
    store_all16<AB + 1>(t, m0, n0, i, h);
  }
}

// byte offset inside a staged 256-wide bf16 tile of the 8-byte group (row, columns col .. col+3), col % 4 == 0: the DMA image
// is swizzled at 16-byte granularity (sn_dw.hip RowStager: LDS piece (row, p) holds global piece (row, p ^ 4 (row & 3)))
SN_DEV unsigned tr_offset16(int row, int col) {
  const int cp = col / 8;
  const int lp = cp ^ (4 * (row & 3));
  return (unsigned)(row * 512 + lp * 16 + (col * 2) % 16);
}

__global__ void __launch_bounds__(256) dw_bf16_asm_kernel(const Plan plan) {
  asm volatile("" ::: "a0", "a255");             // size the kernel for the whole hand-managed AGPR file
  const Task t = task_of(plan, (int)blockIdx.x);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = wr * 128, n0 = wc * 128;
  const long k0 = t.k0, k1 = t.k1;
  if (k0 >= k1) return;
  const int n_chunks = (int)((k1 - k0) / KB);
  const int n_pairs = n_chunks >> 1;
  // per-thread global byte offsets of the 2 + 2 DMA pieces of a chunk (16 rows x 512 B per tile; piece = 256 threads x 16 B),
  // swizzled: LDS piece (row, lp) receives the global piece (row, lp ^ 4 (row & 3))
  unsigned oa[2], ob[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = it * 256 + tid;
    const int row = c >> 5, lp = c & 31;
    const int gp = lp ^ (4 * (row & 3));
    oa[it] = (unsigned)(row * t.lda * 2 + gp * 16);
    ob[it] = (unsigned)(row * t.ldb * 2 + gp * 16);
  }
  // transpose-read offsets of this lane inside a slot: lane (q, G): feature block G & 1, point rows 8 (G >> 1) + (q >> 2)
  unsigned ta[4], tb[4];
  {
    const int q = lane & 15, G = lane >> 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      ta[a] = tr_offset16(8 * (G >> 1) + (q >> 2), m0 + 32 * a + 16 * (G & 1) + 4 * (q & 3));
      tb[a] = tr_offset16(8 * (G >> 1) + (q >> 2), n0 + 32 * a + 16 * (G & 1) + 4 * (q & 3)) + B16_A_BYTES;
    }
  }
  const char* ga_base = reinterpret_cast<const char*>(t.a);
  const char* gb_base = reinterpret_cast<const char*>(t.b);
  auto chunk_base = [&](const char* g, int ld, long k) __attribute__((always_inline)) {      // wave-uniform; a chunk past the
    const long kc = k < k1 ? k : k1 - KB;                                                    // end re-reads the last one
    return g + kc * ld * 2;
  };
  // prologue: six chunks in flight (dynamic LDS starts at address 0: the kernel has no static __shared__)
#pragma unroll
  for (int c = 0; c < B16_NBUF - 2; ++c) {
    const char* ba = chunk_base(ga_base, t.lda, k0 + (long)c * KB);
    const char* bb = chunk_base(gb_base, t.ldb, k0 + (long)c * KB);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)(ba + oa[it]), (lds_void*)(size_t)(c * B16_BUF + it * 4096 + wave * 1024), 16, 0, 2);
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)(bb + ob[it]), (lds_void*)(size_t)(c * B16_BUF + B16_A_BYTES + it * 4096 + wave * 1024), 16, 0, 2);
    }
  }
  asm volatile(SN_DWBF16_ZERO_ASM ::: SN_DWBF16_AGPR_CLOBBERS);
  float bs0 = 0.0f, bs1 = 0.0f, bs2 = 0.0f, bs3 = 0.0f;
  unsigned one = 0x3f803f80u;                      // bf16 (1, 1): v_dot2c_f32_bf16 with it = fp32 sum of a packed pair
  asm volatile("" : "+v"(one));
#pragma unroll 1
  for (int p = 0; p < n_pairs; ++p) {
    const int c = 2 * p;
    const unsigned sl0 = (unsigned)(c % B16_NBUF) * B16_BUF, sl1 = (unsigned)((c + 1) % B16_NBUF) * B16_BUF;
    // staged while this pair is consumed: chunks c+6, c+7 into the slots of chunks c-2, c-1
    const long kn0 = k0 + (long)(c + B16_NBUF - 2) * KB, kn1 = kn0 + KB;
    const char* ga0 = chunk_base(ga_base, t.lda, kn0);
    const char* gb0 = chunk_base(gb_base, t.ldb, kn0);
    const char* ga1 = chunk_base(ga_base, t.lda, kn1);
    const char* gb1 = chunk_base(gb_base, t.ldb, kn1);
    const unsigned md0 = (unsigned)((c + B16_NBUF - 2) % B16_NBUF) * B16_BUF + (unsigned)wave * 1024u;
    const unsigned md1 = (unsigned)((c + B16_NBUF - 1) % B16_NBUF) * B16_BUF + (unsigned)wave * 1024u;
    asm volatile(SN_DWBF16_PAIR_ASM
                 : [bs0] "+v"(bs0), [bs1] "+v"(bs1), [bs2] "+v"(bs2), [bs3] "+v"(bs3)
                 : [ta0] "v"(ta[0]), [ta1] "v"(ta[1]), [ta2] "v"(ta[2]), [ta3] "v"(ta[3]),
                   [tb0] "v"(tb[0]), [tb1] "v"(tb[1]), [tb2] "v"(tb[2]), [tb3] "v"(tb[3]),
                   [one] "v"(one), [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [ob0] "v"(ob[0]), [ob1] "v"(ob[1]),
                   [sl0] "s"(sl0), [sl1] "s"(sl1), [ga0] "s"(ga0), [gb0] "s"(gb0), [ga1] "s"(ga1), [gb1] "s"(gb1),
                   [md0] "s"(md0), [md1] "s"(md1)
                 : SN_DWBF16_VGPR_CLOBBERS, SN_DWBF16_AGPR_CLOBBERS, "memory", "scc");
  }
  if (n_chunks & 1) {                              // odd last chunk (it landed long ago: the ring runs six chunks ahead)
    const unsigned sl0 = (unsigned)((n_chunks - 1) % B16_NBUF) * B16_BUF;
    asm volatile(SN_DWBF16_TAIL_ASM
                 : [bs0] "+v"(bs0), [bs1] "+v"(bs1), [bs2] "+v"(bs2), [bs3] "+v"(bs3)
                 : [ta0] "v"(ta[0]), [ta1] "v"(ta[1]), [ta2] "v"(ta[2]), [ta3] "v"(ta[3]),
                   [tb0] "v"(tb[0]), [tb1] "v"(tb[1]), [tb2] "v"(tb[2]), [tb3] "v"(tb[3]), [one] "v"(one), [sl0] "s"(sl0)
                 : SN_DWBF16_VGPR_CLOBBERS, SN_DWBF16_AGPR_CLOBBERS, "memory", "scc");
  }
  // drain the over-issued tail chunks before the LDS is released; MFMA (8 passes) -> accumulator read: 11 wait states
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  store_all16<0>(t, m0, n0, i, h);
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

extern "C" int sn_dw_bf16_asm_launch(const snd::Plan* plan_host, hipStream_t stream) {
  using namespace snd;
  if (plan_host->n_tasks <= 0) return 0;
  SN_ENSURE_DYN_LDS(dw_bf16_asm_kernel, DW_LDS_BYTES);
  hipLaunchKernelGGL(dw_bf16_asm_kernel, dim3((unsigned)plan_host->n_tasks), dim3(256), DW_LDS_BYTES, stream, *plan_host);
  return (int)hipGetLastError();
}
