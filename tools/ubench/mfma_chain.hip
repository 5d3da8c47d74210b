// micro-benchmark: what interrupts a v_mfma_f32_32x32x2_f32 stream?  (hipcc --offload-arch=gfx950 -O3)
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3f * i;
  __syncthreads();
  f32x16 a0, a1;
  for (int r = 0; r < 16; ++r) { a0[r] = 0; a1[r] = 0; }
  float b[8];
  for (int i = 0; i < 8; ++i) b[i] = 1.0f + 0.001f * (lane + i);
  f32x4 a = *reinterpret_cast<const f32x4*>(&lds[lane * 4]);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      f32x4 an = a;
      if (MODE >= 1) an = *reinterpret_cast<const f32x4*>(&lds[((it * 8 + g + 1) & 31) * 256 + lane * 4]);
      if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
      if (MODE == 2 || MODE == 4) {     // two chains
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], a0, 0, 0, 0);
      } else {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], a0, 0, 0, 0);
      }
      if (MODE >= 3 && g == 3) __syncthreads();
      a = an;
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* d, int blocks) {
  const int iters = blocks <= 256 ? 40000 : 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mf = (double)blocks * 4 * iters * 32;   // MFMAs
  printf("%-28s blocks=%d  %.3f ms  %.1f TFLOP/s  %.1f cycles/MFMA@2.4GHz\n", name, blocks, ms, mf * 4096 / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (iters * 32.0) / ((blocks + 255) / 256));
}
int main() {
  float* d; hipMalloc(&d, 1 << 22);
  for (int blocks : {256, 512, 2560}) {
    run<0>("pure dependent chain", d, blocks);
    run<1>("chain + ds_read/4", d, blocks);
    run<2>("2 chains + ds_read/4", d, blocks);
    run<3>("chain + ds_read/4 + barrier/32", d, blocks);
    run<4>("2 chains + ds_read + barrier", d, blocks);
  }
  return 0;
}
