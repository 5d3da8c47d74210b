// micro-benchmark: what does ONE instruction of each class cost when it sits between two v_mfma_f32_32x32x2_f32 of a wave that has its SIMD
// to itself (the fused fp32 MLP kernel's situation: 512 registers per lane, one wave per SIMD)?  The round-1 law "64.2 N_mfma + 4.6 N_other" lumps
// every class together; a generated fp32 trunk needs to know WHICH of them are worth removing (VALU / v_accvgpr_write / SALU / s_waitcnt /
// ds_read_b128 / global_load / LDS-DMA) and whether the f32-input MFMA (which runs at the f32 VECTOR rate) shares its pipe with the VALU.
// Every loop body is ONE asm statement (no compiler scheduling): 32 MFMAs on two accumulator chains in alternation (or one chain), filler F
// behind every MFMA, filler Gk behind every fourth.  Cycles per MFMA from s_memtime (shader clock), wall time from HIP events.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/f32_gap_cost tools/ubench/f32_gap_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// operands: %0 c0  %1 c1  %2 a  %3 b  %4..%7 t0..t3  %8 s0  %9 lds byte address  %10..%13 q0..q3  %14 global lane offset  %15 global base (sgpr pair)
//           %16 m0 value (this wave's 4 KB LDS window)
#define MA "v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\t"
#define MB "v_mfma_f32_32x32x2_f32 %1, %2, %3, %1\n\t"
#define MA_AG "v_mfma_f32_32x32x2_f32 %0, %2, a8, %0\n\t"
#define MB_AG "v_mfma_f32_32x32x2_f32 %1, %2, a9, %1\n\t"
#define GROUP2(F, G) MA F MB F MA F MB F G
#define GROUP1(F, G) MA F MA F MA F MA F G
#define GROUPAG(F, G) MA_AG F MB_AG F MA_AG F MB_AG F G
#define BODY2(F, G0, G1, G2, G3) GROUP2(F, G0) GROUP2(F, G1) GROUP2(F, G2) GROUP2(F, G3) GROUP2(F, G0) GROUP2(F, G1) GROUP2(F, G2) GROUP2(F, G3)
#define BODY1(F, G0, G1, G2, G3) GROUP1(F, G0) GROUP1(F, G1) GROUP1(F, G2) GROUP1(F, G3) GROUP1(F, G0) GROUP1(F, G1) GROUP1(F, G2) GROUP1(F, G3)
#define BODYAG(F, G0, G1, G2, G3) GROUPAG(F, G0) GROUPAG(F, G1) GROUPAG(F, G2) GROUPAG(F, G3) GROUPAG(F, G0) GROUPAG(F, G1) GROUPAG(F, G2) GROUPAG(F, G3)

#define F_NONE ""
#define F_NOP "s_nop 0\n\t"
#define F_VMOV1 "v_mov_b32 %4, %2\n\t"
#define F_VMOV2 "v_mov_b32 %4, %2\n\tv_mov_b32 %5, %3\n\t"
#define F_VMOV4 "v_mov_b32 %4, %2\n\tv_mov_b32 %5, %3\n\tv_mov_b32 %6, %2\n\tv_mov_b32 %7, %3\n\t"
#define F_VMAX1 "v_max_f32 %4, 0, %2\n\t"
#define F_VMAX2 "v_max_f32 %4, 0, %2\n\tv_max_f32 %5, 0, %3\n\t"
#define F_VFMA1 "v_fma_f32 %4, %2, %3, %4\n\t"
#define F_ACCW1 "v_accvgpr_write_b32 a0, %2\n\t"
#define F_ACCW2 "v_accvgpr_write_b32 a0, %2\n\tv_accvgpr_write_b32 a1, %3\n\t"
#define F_RELU2 "v_max_f32 %4, 0, %2\n\tv_accvgpr_write_b32 a0, %4\n\t"
#define F_SALU1 "s_add_u32 %8, %8, 1\n\t"
#define F_SALU2 "s_add_u32 %8, %8, 1\n\ts_add_u32 %8, %8, 3\n\t"
#define F_SALU4 "s_add_u32 %8, %8, 1\n\ts_add_u32 %8, %8, 3\n\ts_add_u32 %8, %8, 5\n\ts_add_u32 %8, %8, 7\n\t"
#define F_WAIT1 "s_waitcnt lgkmcnt(15)\n\t"
#define F_WAITV "s_waitcnt vmcnt(63)\n\t"
// LDS atomics (the ReLU-by-integer-max of the first round-6 epilogue) and LDS stores
#define F_DSMAX16 "ds_max_i32 %9, %4\n\t"                 /* lane stride 16 B: 4-way bank conflict */
#define F_DSMAX4 "ds_max_i32 %17, %4\n\t"                 /* lane stride 4 B: conflict-free */
#define F_DSW128 "ds_write_b128 %9, %10\n\t"
#define G_VMAX16 "v_max_f32 %4, 0, %2\n\tv_max_f32 %5, 0, %3\n\tv_max_f32 %6, 0, %2\n\tv_max_f32 %7, 0, %3\n\t" \
                 "v_max_f32 %4, 0, %2\n\tv_max_f32 %5, 0, %3\n\tv_max_f32 %6, 0, %2\n\tv_max_f32 %7, 0, %3\n\t" \
                 "v_max_f32 %4, 0, %2\n\tv_max_f32 %5, 0, %3\n\tv_max_f32 %6, 0, %2\n\tv_max_f32 %7, 0, %3\n\t" \
                 "v_max_f32 %4, 0, %2\n\tv_max_f32 %5, 0, %3\n\tv_max_f32 %6, 0, %2\n\tv_max_f32 %7, 0, %3\n\t"
// per-group fillers (k = group 0..3)
#define G_NONE ""
#define G_DS0 "ds_read_b128 %10, %9\n\t"
#define G_DS1 "ds_read_b128 %11, %9 offset:1024\n\t"
#define G_DS2 "ds_read_b128 %12, %9 offset:2048\n\t"
#define G_DS3 "ds_read_b128 %13, %9 offset:3072\n\t"
#define G_DSW0 "s_waitcnt lgkmcnt(2)\n\tds_read_b128 %10, %9\n\t"
#define G_DSW1 "s_waitcnt lgkmcnt(2)\n\tds_read_b128 %11, %9 offset:1024\n\t"
#define G_DSW2 "s_waitcnt lgkmcnt(2)\n\tds_read_b128 %12, %9 offset:2048\n\t"
#define G_DSW3 "s_waitcnt lgkmcnt(2)\n\tds_read_b128 %13, %9 offset:3072\n\t"
#define G_GL0 "s_waitcnt vmcnt(3)\n\tglobal_load_dwordx4 %10, %14, %15\n\t"
#define G_GL1 "s_waitcnt vmcnt(3)\n\tglobal_load_dwordx4 %11, %14, %15 offset:1024\n\t"
#define G_GL2 "s_waitcnt vmcnt(3)\n\tglobal_load_dwordx4 %12, %14, %15 offset:2048\n\t"
#define G_GL3 "s_waitcnt vmcnt(3)\n\tglobal_load_dwordx4 %13, %14, %15 offset:3072\n\t"
#define G_GLN0 "global_load_dwordx4 %10, %14, %15\n\t"
#define G_GLN1 "global_load_dwordx4 %11, %14, %15 offset:1024\n\t"
#define G_GLN2 "global_load_dwordx4 %12, %14, %15 offset:2048\n\t"
#define G_GLN3 "global_load_dwordx4 %13, %14, %15 offset:3072\n\t"
#define G_DMA0 "s_mov_b32 m0, %16\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %14, %15\n\t"
#define G_DMA1 "s_mov_b32 m0, %16\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %14, %15 offset:1024\n\t"
#define G_DMA2 "s_mov_b32 m0, %16\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %14, %15 offset:2048\n\t"
#define G_DMA3 "s_mov_b32 m0, %16\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %14, %15 offset:3072\n\t"
// the shipped kernel's mix per group of four MFMAs: one ds_read_b128 (+ its counted wait every other group), ~0.7 DMA pieces (m0 + load),
// ~0.8 SALU of ring arithmetic, ~0.9 epilogue VALU
#define G_MIX0 "s_waitcnt lgkmcnt(1)\n\tds_read_b128 %10, %9\n\ts_mov_b32 m0, %16\n\ts_add_u32 %8, %8, 1\n\tglobal_load_lds_dwordx4 %14, %15\n\tv_max_f32 %4, 0, %2\n\t"
#define G_MIX1 "ds_read_b128 %11, %9 offset:1024\n\ts_mov_b32 m0, %16\n\ts_add_u32 %8, %8, 1\n\tglobal_load_lds_dwordx4 %14, %15 offset:1024\n\tv_accvgpr_write_b32 a0, %4\n\t"
#define G_MIX2 "s_waitcnt lgkmcnt(1)\n\tds_read_b128 %12, %9 offset:2048\n\ts_add_u32 %8, %8, 1\n\tv_max_f32 %5, 0, %3\n\t"
#define G_MIX3 "ds_read_b128 %13, %9 offset:3072\n\ts_mov_b32 m0, %16\n\ts_add_u32 %8, %8, 1\n\tglobal_load_lds_dwordx4 %14, %15 offset:3072\n\tv_accvgpr_write_b32 a1, %5\n\t"
// a generated trunk's mix: A fragments by global_load (no LDS, no DMA, no barrier), epilogue VALU only
#define G_GEN0 "s_waitcnt vmcnt(3)\n\tglobal_load_dwordx4 %10, %14, %15\n\tv_max_f32 %4, 0, %2\n\t"
#define G_GEN1 "s_waitcnt vmcnt(3)\n\tglobal_load_dwordx4 %11, %14, %15 offset:1024\n\tv_accvgpr_write_b32 a0, %4\n\t"
#define G_GEN2 "s_waitcnt vmcnt(3)\n\tglobal_load_dwordx4 %12, %14, %15 offset:2048\n\tv_max_f32 %5, 0, %3\n\t"
#define G_GEN3 "s_waitcnt vmcnt(3)\n\tglobal_load_dwordx4 %13, %14, %15 offset:3072\n\tv_accvgpr_write_b32 a1, %5\n\t"
// ... with the fragments still through LDS (ds_read + counted wait per pair), DMA with immediate offsets (no SALU)
#define G_GENL0 "s_waitcnt lgkmcnt(1)\n\tds_read_b128 %10, %9\n\tglobal_load_lds_dwordx4 %14, %15\n\tv_max_f32 %4, 0, %2\n\t"
#define G_GENL1 "ds_read_b128 %11, %9 offset:1024\n\tv_accvgpr_write_b32 a0, %4\n\t"
#define G_GENL2 "s_waitcnt lgkmcnt(1)\n\tds_read_b128 %12, %9 offset:2048\n\tglobal_load_lds_dwordx4 %14, %15 offset:2048\n\tv_max_f32 %5, 0, %3\n\t"
#define G_GENL3 "ds_read_b128 %13, %9 offset:3072\n\tv_accvgpr_write_b32 a1, %5\n\t"

#define KERNEL(NAME, BODYSTR)                                                                                                       \
  __global__ void __launch_bounds__(256) NAME(const float* __restrict__ src, const char* __restrict__ wts, float* out, long long* clk, \
                                              int iters) {                                                                          \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                                                     \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                                                      \
    float a = src[threadIdx.x & 1023], b = src[(threadIdx.x * 3 + 7) & 1023];                                                        \
    f32x16 c0, c1;                                                                                                                   \
    for (int r = 0; r < 16; ++r) { c0[r] = 0.0f; c1[r] = 0.0f; }                                                                     \
    float t0 = a, t1 = b, t2 = a, t3 = b;                                                                                            \
    int s0 = 0;                                                                                                                      \
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(smem)[i] = src[i & 1023];                                \
    __syncthreads();                                                                                                                 \
    unsigned la = (unsigned)(size_t)smem + wave * 4096 + lane * 16;                                                            \
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + 32768 + wave * 4096);                               \
    f32x4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;                                                                              \
    const unsigned goff = lane * 16;                                                                                                 \
    const unsigned la4 = (unsigned)(size_t)smem + 16384 + wave * 4096 + lane * 4;                                                    \
    asm volatile("v_accvgpr_write_b32 a8, %0\n\tv_accvgpr_write_b32 a9, %1" ::"v"(b), "v"(a) : "a0", "a1", "a8", "a9");             \
    long long t_begin = 0, t_end = 0, r_begin = 0, r_end = 0;                                                                        \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin), "=s"(r_begin));                         \
    for (int it = 0; it < iters; ++it) {                                                                                             \
      const char* gb = wts + ((it & 255) << 12);                /* walks a 1 MB window: L2-resident, the same for every wave */     \
      asm volatile(BODYSTR                                                                                                           \
                   : "+v"(c0), "+v"(c1), "+v"(a), "+v"(b), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+s"(s0), "+v"(la), "+v"(q0),     \
                     "+v"(q1), "+v"(q2), "+v"(q3)                                                                                    \
                   : "v"(goff), "s"(gb), "s"(m0v), "v"(la4)                                                                          \
                   : "memory", "a0", "a1");                                                                                          \
    }                                                                                                                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end), "=s"(r_end)); \
    float s = t0 + t1 + t2 + t3 + (float)s0 + q0[0] + q1[1] + q2[2] + q3[3];                                                         \
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];                                                                                 \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                                                         \
    if (lane == 0) { clk[(blockIdx.x * 4 + wave) * 2] = t_end - t_begin; clk[(blockIdx.x * 4 + wave) * 2 + 1] = r_end - r_begin; }   \
  }

KERNEL(k_bare2, BODY2(F_NONE, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_bare1, BODY1(F_NONE, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_bare_agprB, BODYAG(F_NONE, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_nop1, BODY2(F_NOP, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_nop1_1chain, BODY1(F_NOP, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_vmov1, BODY2(F_VMOV1, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_vmov1_1chain, BODY1(F_VMOV1, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_vmov2, BODY2(F_VMOV2, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_vmov4, BODY2(F_VMOV4, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_vmax1, BODY2(F_VMAX1, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_vmax2, BODY2(F_VMAX2, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_vfma1, BODY2(F_VFMA1, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_accw1, BODY2(F_ACCW1, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_accw2, BODY2(F_ACCW2, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_relu2, BODY2(F_RELU2, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_salu1, BODY2(F_SALU1, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_salu2, BODY2(F_SALU2, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_salu4, BODY2(F_SALU4, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_wait1, BODY2(F_WAIT1, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_waitv1, BODY2(F_WAITV, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_ds_per4, BODY2(F_NONE, G_DS0, G_DS1, G_DS2, G_DS3))
KERNEL(k_ds_per4_1chain, BODY1(F_NONE, G_DS0, G_DS1, G_DS2, G_DS3))
KERNEL(k_dsw_per4, BODY2(F_NONE, G_DSW0, G_DSW1, G_DSW2, G_DSW3))
KERNEL(k_ds_per1, BODY2(G_DS0, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_gl_per4, BODY2(F_NONE, G_GL0, G_GL1, G_GL2, G_GL3))
KERNEL(k_gl_nowait_per4, BODY2(F_NONE, G_GLN0, G_GLN1, G_GLN2, G_GLN3))
KERNEL(k_dma_per4, BODY2(F_NONE, G_DMA0, G_DMA1, G_DMA2, G_DMA3))
KERNEL(k_mix_shipped, BODY1(F_NONE, G_MIX0, G_MIX1, G_MIX2, G_MIX3))
KERNEL(k_mix_shipped_2chains, BODY2(F_NONE, G_MIX0, G_MIX1, G_MIX2, G_MIX3))
KERNEL(k_mix_gen_global, BODY1(F_NONE, G_GEN0, G_GEN1, G_GEN2, G_GEN3))
KERNEL(k_mix_gen_global_2chains, BODY2(F_NONE, G_GEN0, G_GEN1, G_GEN2, G_GEN3))
KERNEL(k_mix_gen_lds, BODY1(F_NONE, G_GENL0, G_GENL1, G_GENL2, G_GENL3))
KERNEL(k_mix_gen_lds_2chains, BODY2(F_NONE, G_GENL0, G_GENL1, G_GENL2, G_GENL3))

// ---- instruction FETCH: the same bodies, but 64 / 256 / 512 copies of them in straight-line code (22 / 90 / 180 KB: the fused MLP kernel
// is ~195 KB of straight-line code per point tile; the instruction cache holds 64 KB per pair of CUs)
#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))
#define REP64(X) REP4(REP16(X))
#define REP256(X) REP4(REP64(X))
#define REP512(X) REP256(X) REP256(X)
KERNEL(k_gl_per4_code22k, REP64(BODY1(F_NONE, G_GL0, G_GL1, G_GL2, G_GL3)))
KERNEL(k_gl_per4_code90k, REP256(BODY1(F_NONE, G_GL0, G_GL1, G_GL2, G_GL3)))
KERNEL(k_gl_per4_code180k, REP512(BODY1(F_NONE, G_GL0, G_GL1, G_GL2, G_GL3)))
KERNEL(k_bare_code130k, REP512(BODY1(F_NONE, G_NONE, G_NONE, G_NONE, G_NONE)))
KERNEL(k_nop_gl_code250k, REP512(BODY1(F_NOP, G_GL0, G_GL1, G_GL2, G_GL3)))

KERNEL(k_dsmax_stride16_per1, BODY1(F_DSMAX16, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_dsmax_stride4_per1, BODY1(F_DSMAX4, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_dswrite128_per1, BODY1(F_DSW128, G_NONE, G_NONE, G_NONE, G_NONE))
KERNEL(k_vmax16_one_gap_per16, BODY1(F_NONE, G_VMAX16, G_NONE, G_NONE, G_NONE))

typedef void (*kern_t)(const float*, const char*, float*, long long*, int);

static void run(const char* name, kern_t kern, double n_other, const float* src, const char* wts, float* d, long long* clk, int n_cu, int rep = 1) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  int iters = 2000 / rep + 1;
  float ms = 0;
  for (int pass = 0; pass < 3; ++pass) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(256), 65536, 0, src, wts, d, clk, iters);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("%-28s FAILED: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
    hipEventElapsedTime(&ms, e0, e1);
    if (pass == 0) iters = (int)(iters * 12.0 / ms) + 1;
  }
  std::vector<long long> h(n_cu * 8);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < n_cu * 4; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  cyc /= n_cu * 4; rt /= n_cu * 4;
  const double n = (double)iters * 32 * rep;
  const double tf = (double)n_cu * 4 * n * 4096 / ms / 1e9;
  printf("%-28s other/MFMA %.2f  %7.3f ms  %6.1f TF  frac %.3f  cycles/MFMA %6.2f  (+%5.2f per other)  clock %.2f GHz\n", name, n_other, ms, tf,
         tf / 157.3, cyc / n, n_other > 0 ? (cyc / n - 64.0) / n_other : 0.0, cyc / (rt * 10.0) );
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  float h[1024];
  srand(3);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *src, *d;
  char* wts;
  long long* clk;
  hipMalloc(&src, sizeof(h)); hipMalloc(&d, n_cu * 256 * 4); hipMalloc(&wts, (1 << 20) + 8192); hipMalloc(&clk, n_cu * 8 * 8);
  hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(wts, 0x3c, (1 << 20) + 8192);
  printf("%s: %d CUs, one workgroup of 4 waves per CU (one wave per SIMD), ~12 ms launches; cycles from s_memtime, clock = cycles / s_memrealtime (100 MHz)\n",
         p.name, n_cu);
#define RUN(K, N) run(#K, K, N, src, wts, d, clk, n_cu)
  for (int rep = 0; rep < 2; ++rep) {
    RUN(k_bare2, 0); RUN(k_bare1, 0); RUN(k_bare_agprB, 0);
    RUN(k_nop1, 1); RUN(k_nop1_1chain, 1);
    RUN(k_vmov1, 1); RUN(k_vmov1_1chain, 1); RUN(k_vmov2, 2); RUN(k_vmov4, 4);
    RUN(k_vmax1, 1); RUN(k_vmax2, 2); RUN(k_vfma1, 1);
    RUN(k_accw1, 1); RUN(k_accw2, 2); RUN(k_relu2, 2);
    RUN(k_salu1, 1); RUN(k_salu2, 2); RUN(k_salu4, 4);
    RUN(k_wait1, 1); RUN(k_waitv1, 1);
    RUN(k_ds_per4, 0.25); RUN(k_ds_per4_1chain, 0.25); RUN(k_dsw_per4, 0.5); RUN(k_ds_per1, 1);
    RUN(k_gl_per4, 0.5); RUN(k_gl_nowait_per4, 0.25); RUN(k_dma_per4, 0.75);
    RUN(k_mix_shipped, 1.06); RUN(k_mix_shipped_2chains, 1.06);
    RUN(k_mix_gen_global, 0.75); RUN(k_mix_gen_global_2chains, 0.75);
    RUN(k_mix_gen_lds, 0.75); RUN(k_mix_gen_lds_2chains, 0.75);
    RUN(k_dsmax_stride16_per1, 1); RUN(k_dsmax_stride4_per1, 1); RUN(k_dswrite128_per1, 1); RUN(k_vmax16_one_gap_per16, 1);
    run("k_gl_per4_code22k", k_gl_per4_code22k, 0.5, src, wts, d, clk, n_cu, 64);
    run("k_gl_per4_code90k", k_gl_per4_code90k, 0.5, src, wts, d, clk, n_cu, 256);
    run("k_gl_per4_code180k", k_gl_per4_code180k, 0.5, src, wts, d, clk, n_cu, 512);
    run("k_bare_code130k", k_bare_code130k, 0, src, wts, d, clk, n_cu, 512);
    run("k_nop_gl_code250k", k_nop_gl_code250k, 1.5, src, wts, d, clk, n_cu, 512);
  }
  return 0;
}
