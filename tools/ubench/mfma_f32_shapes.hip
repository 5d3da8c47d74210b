// micro-benchmark: what do the fp32-input MFMA shapes of gfx950 SUSTAIN in a bare loop?  The fused fp32 MLP kernel (csrc/sn_mlp_fwd.hip) runs its
// pipe 91 % busy at the full 2.38 GHz with v_mfma_f32_32x32x2_f32 and two accumulator chains did not change that (profiles/r05_f32_two_chains_ab.txt):
// is 0.91 the instruction's ceiling, and does the other shape (16x16x4: half the issue time, same FLOP rate) sit closer to the 157.3 TF peak?
// One wave per SIMD, 256 workgroups, N independent accumulators in rotation, register operands, ~20 ms launches.
//   32x32x2 : 64 cycles per instruction per SIMD, 4096 FLOP        16x16x4 : 32 cycles, 2048 FLOP        (both 64 FLOP / cycle / SIMD)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_f32_shapes tools/ubench/mfma_f32_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int FILL>
__global__ void __launch_bounds__(256) k32(const float* __restrict__ src, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + 64 * i) & 1023]; b[i] = src[(threadIdx.x * 3 + 64 * i + 7) & 1023]; }
  f32x16 c[CHAINS];
  for (int j = 0; j < CHAINS; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.0f;
  float junk = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c[g % CHAINS]) : "v"(a[g]), "v"(b[g]));
      if (FILL) asm volatile("v_add_f32 %0, %0, %1\n\tv_max_f32 %0, 0, %0" : "+v"(junk) : "v"(a[g]));
    }
    if ((it & 255) == 255) for (int j = 0; j < CHAINS; ++j) for (int r = 0; r < 16; ++r) c[j][r] *= 1e-6f;
  }
  float s = junk;
  for (int j = 0; j < CHAINS; ++j) for (int r = 0; r < 16; ++r) s += c[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS, int FILL>
__global__ void __launch_bounds__(256) k16(const float* __restrict__ src, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + 64 * i) & 1023]; b[i] = src[(threadIdx.x * 3 + 64 * i + 7) & 1023]; }
  f32x4 c[CHAINS];
  for (int j = 0; j < CHAINS; ++j) for (int r = 0; r < 4; ++r) c[j][r] = 0.0f;
  float junk = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[g % CHAINS]) : "v"(a[g]), "v"(b[g]));
      if (FILL && (g & 1)) asm volatile("v_add_f32 %0, %0, %1\n\tv_max_f32 %0, 0, %0" : "+v"(junk) : "v"(a[g]));
    }
    if ((it & 255) == 255) for (int j = 0; j < CHAINS; ++j) for (int r = 0; r < 4; ++r) c[j][r] *= 1e-6f;
  }
  float s = junk;
  for (int j = 0; j < CHAINS; ++j) for (int r = 0; r < 4; ++r) s += c[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K>
void run(const char* name, K kern, double flop_per_mfma, double cyc_per_mfma, const float* src, float* d, int n_cu) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 4000;
  float ms = 0;
  for (int pass = 0; pass < 3; ++pass) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(256), 0, 0, src, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (pass == 0) iters = (int)(iters * 20.0 / ms);
  }
  const double n = (double)iters * 8;
  const double tf = (double)n_cu * 4 * n * flop_per_mfma / ms / 1e9;
  printf("%-58s %7.3f ms  %6.1f TFLOP/s  frac of 157.3 TF %.3f  cycles per MFMA at 2.4 GHz %.1f (issue time %g)\n", name, ms, tf, tf / 157.3,
         ms * 1e-3 * 2.4e9 / n, cyc_per_mfma);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  float h[1024];
  srand(3);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *src, *d;
  hipMalloc(&src, sizeof(h)); hipMalloc(&d, n_cu * 256 * 4);
  hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  printf("%s: %d CUs, one wave per SIMD, ~20 ms launches, random fp32 operands in registers\n", p.name, n_cu);
  for (int rep = 0; rep < 2; ++rep) {
    run("32x32x2, 1 chain", k32<1, 0>, 4096, 64, src, d, n_cu);
    run("32x32x2, 2 chains", k32<2, 0>, 4096, 64, src, d, n_cu);
    run("32x32x2, 4 chains", k32<4, 0>, 4096, 64, src, d, n_cu);
    run("32x32x2, 1 chain + 2 VALU per MFMA", k32<1, 1>, 4096, 64, src, d, n_cu);
    run("32x32x2, 2 chains + 2 VALU per MFMA", k32<2, 1>, 4096, 64, src, d, n_cu);
    run("16x16x4, 1 chain", k16<1, 0>, 2048, 32, src, d, n_cu);
    run("16x16x4, 2 chains", k16<2, 0>, 2048, 32, src, d, n_cu);
    run("16x16x4, 4 chains", k16<4, 0>, 2048, 32, src, d, n_cu);
    run("16x16x4, 8 chains", k16<8, 0>, 2048, 32, src, d, n_cu);
    run("16x16x4, 4 chains + 2 VALU per MFMA pair", k16<4, 1>, 2048, 32, src, d, n_cu);
  }
  return 0;
}
