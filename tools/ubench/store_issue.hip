// micro-benchmark: what does a row store cost a wave that also feeds the MFMA pipe, and is the cost the four waves of a CU queueing at the
// CU's one texture-addresser?  One workgroup per CU (139 KB of LDS), four waves, an iteration = 32 back-to-back v_mfma_f32_32x32x16_bf16
// (1024 cycles: one weight slab of the training kernels) with 1 KB global_store_dwordx4 nt instructions dealt into the gaps.
//   hipcc --offload-arch=gfx950 -O3 store_issue.hip -o store_issue   (store_issue_bodies.inc: gen_store_issue.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "store_issue_bodies.inc"

#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v24","v25","v26","v27", \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
  "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63", "memory"

// MODE: 0 no stores | 1 8 stores, barrier per iteration (the kernels' situation) | 2 8 stores, no barrier | 3 4 stores + barrier | 4 16 stores + barrier
//       5 a store slot in every gap, live for wave (gap % 4) only (same 8 per wave, never two waves at once) + barrier
//       6 8 stores from wave 0 only + barrier | 7 8 loads (dwordx4) + barrier | 8 mode 1 with rows 512 B apart (128-byte pieces)
//       9 mode 1 + a wave-dependent delay after every barrier (64 w cycles)
template <int MODE>
__global__ void __launch_bounds__(256) k(char* buf, int iters) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = buf + ((size_t)(blockIdx.x * 4 + wave) << 21);             // 2 MB window per wave
  unsigned off = MODE == 8 ? (lane >> 3) * 512 + (lane & 7) * 16 : lane * 16;
  const unsigned wrap = 0x1fffffu;
  asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\t"
               "v_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\tv_mov_b32 v16, 1.0\n\tv_mov_b32 v17, 1.0\n\tv_mov_b32 v18, 1.0\n\tv_mov_b32 v19, 1.0" ::"v"(off) : CLOB);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (MODE != 2) __builtin_amdgcn_s_barrier();
    if (MODE == 9) for (int w = 0; w < wave; ++w) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
#define RUN(B) asm volatile(B ::[base] "s"(base), [wrap] "s"(wrap), [wave] "s"(wave), [inc] "s"(MODE == 8 ? 4096 : 1024) : CLOB, "s40", "s41", "s42", "s43", "scc")
    if (MODE == 0) RUN(BODY_NONE);
    if (MODE == 1 || MODE == 2 || MODE == 8 || MODE == 9) RUN(BODY_ST8);
    if (MODE == 3) RUN(BODY_ST4);
    if (MODE == 4) RUN(BODY_ST16);
    if (MODE == 5) RUN(BODY_ROT);
    if (MODE == 6) RUN(BODY_SOLO);
    if (MODE == 7) RUN(BODY_LD8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 77 && iters < 0) buf[0] = 1;
}

// plain streaming-write kernels: the write ceiling of the chip without any compute
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void __launch_bounds__(256) wr(vf4* p, size_t n) {
  const vf4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(v, &p[i]); else p[i] = v;
  }
}
template <int NT>
void run_wr(const char* name, char* d, int blocks) {
  const size_t bytes = (size_t)1024 << 21, n = bytes / 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(wr<NT>, dim3(blocks), dim3(256), 0, 0, (vf4*)d, n);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(wr<NT>, dim3(blocks), dim3(256), 0, 0, (vf4*)d, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-58s %8.3f ms for 2 GiB = %.2f TB/s\n", name, best, bytes / best / 1e9);
}

template <int MODE>
void run(const char* name, char* d, double t0) {
  const int iters = 4000, blocks = 256;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 139264);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 139264, 0, d, 200);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 139264, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-58s %8.1f ns / iteration   (x %.3f of MFMA only)\n", name, best * 1e6 / iters, t0 > 0 ? best * 1e6 / iters / t0 : 1.0);
  fflush(stdout);
}
template <int MODE> double base_ns(char* d) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 139264);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 139264, 0, d, 200);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 139264, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters;
}
int main() {
  char* d; if (hipMalloc(&d, ((size_t)1024 << 21) + (4 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0, (size_t)1024 << 21);
  const double t0 = base_ns<0>(d);
  run<0>("0 MFMA only (32 per iteration, barrier)", d, t0);
  run<1>("1 + 8 row stores (1 KB each), barrier per iteration", d, t0);
  run<2>("2 + 8 row stores, no barrier (waves drift)", d, t0);
  run<3>("3 + 4 row stores, barrier", d, t0);
  run<4>("4 + 16 row stores, barrier", d, t0);
  run<5>("5 + 8 row stores, waves take turns (exec-masked slots)", d, t0);
  run<6>("6 + 8 row stores from wave 0 only", d, t0);
  run<7>("7 + 8 row loads (dwordx4), barrier", d, t0);
  run<8>("8 + 8 row stores as 8 x 128 B pieces 512 B apart", d, t0);
  run<9>("9 + 8 row stores, 64 w cycles of delay after the barrier", d, t0);
  run_wr<1>("streaming write, nt, 2048 x 256 threads x 16 B", d, 2048);
  run_wr<0>("streaming write, plain, 2048 x 256", d, 2048);
  run_wr<1>("streaming write, nt, 8192 x 256", d, 8192);
  run_wr<0>("streaming write, plain, 8192 x 256", d, 8192);
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemsetAsync(d, 0, (size_t)1024 << 21, 0);
    hipEventRecord(e0); hipMemsetAsync(d, 1, (size_t)1024 << 21, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.3f ms for 2 GiB = %.2f TB/s\n", "hipMemsetAsync", ms, ((size_t)1024 << 21) / ms / 1e9);
  }
  return 0;
}
