// micro-benchmark: what bf16 MFMA rate does an MI355X SUSTAIN (power-limited clock, MI355X_MICROARCH.md "DVFS give-back")?
// One wave per SIMD (4 per CU, 256 workgroups = one per CU, all co-resident), two independent accumulator chains of
// v_mfma_f32_32x32x16_bf16 issued back to back -- the backbone of csrc/sn_mlp_fwd_bf16_v3.hip without anything else -- for
// ~25 ms per launch (the length of the bench's fine-pass launch), with
//   mode 0: all-zero operands            (no toggling in the multiplier array: the chip's best case)
//   mode 1: random bf16 operands         (8 A and 8 B fragments in rotation)
//   mode 2: mode 1 + one ds_read_b128 of a random LDS tile per MFMA pair (the A-fragment stream of the fused MLP kernel)
//   mode 3: mode 2 + 3 VALU instructions (v_cvt_pk_bf16_f32 / v_pk_max / v_accvgpr_write-like moves) per MFMA
//   mode 4: mode 1 with HALF of the B operands zero (a ReLU layer's activations: the fused MLP's actual B operand statistics)
//   mode 5: mode 4 + the ds_read_b128 per MFMA pair of mode 2
//   mode 6: mode 4 + one ds_read_b128 per FOUR MFMAs -- the A-fragment stream of a wave that owned FOUR point tiles (half
//           the fragment reads per MFMA; the AGPR file holds the activations of two tiles, so this flow does not exist as a kernel)
//   mode 7: mode 6 with the B operands of every second MFMA read from LDS as well (one more ds_read_b128 per two MFMAs): the same four-tile
//           flow with the activations of two of the tiles kept in LDS instead of AGPRs -- the only way it would fit the register file
// prints TFLOP/s (dense, 32*32*16*2 FLOP per MFMA) and the clock an MFMA-bound stream implies (32 cycles per MFMA per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_bf16_peak tools/ubench/mfma_bf16_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE_>
__global__ void __launch_bounds__(256) k(const u32x4* __restrict__ src, float* out, int iters) {
  constexpr int MODE = MODE_ == 4 ? 1 : MODE_ == 5 ? 2 : MODE_ >= 6 ? 1 : MODE_;      // 4, 5: the loops of 1, 2 over ReLU-like B operands

  __shared__ __attribute__((aligned(16))) u32x4 lds[2048];           // 32 KB of operand tiles
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = (MODE == 0) ? u32x4{0, 0, 0, 0} : src[i];
  __syncthreads();
  u32x4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = lds[i * 64 + lane];
    b[i] = lds[(8 + i) * 64 + lane];
    if (MODE_ >= 4) {                                                 // zero half of the bf16 values (pseudo-random pattern per lane / slot)
      unsigned m = (unsigned)(lane * 2654435761u + i * 40503u);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (m & (1u << (2 * w))) b[i][w] &= 0xffff0000u;
        if (m & (1u << (2 * w + 1))) b[i][w] &= 0x0000ffffu;
      }
    }
  }
  f32x16 c0, c1;
  for (int r = 0; r < 16; ++r) { c0[r] = 0.0f; c1[r] = 0.0f; }
  unsigned junk = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      u32x4 an = a[g];
      if (MODE >= 2) an = lds[((it * 8 + g) & 31) * 64 + lane];
      if (MODE_ >= 6 && (g & 1) == 0) an = lds[((it * 8 + g) & 31) * 64 + lane];          // one A fragment per FOUR MFMAs
      u32x4 bn = b[g];
      if (MODE_ == 7) bn = lds[((it * 8 + g + 16) & 31) * 64 + lane];                     // ... and the B operand of every second MFMA from LDS
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a[g]), "v"(b[g]));
      if (MODE >= 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0\n\tv_mov_b32 %0, %0" : "+v"(junk) : "v"(c0[0]), "v"(c0[1]));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a[g]), "v"(b[(g + 1) & 7]));
      if (MODE >= 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0\n\tv_mov_b32 %0, %0" : "+v"(junk) : "v"(c1[0]), "v"(c1[1]));
      if (MODE >= 2 || (MODE_ >= 6 && (g & 1) == 0)) a[g] = an;
      if (MODE_ == 7) b[g] = bn;
    }
    if ((it & 63) == 63) {                                            // keep the accumulators finite: scale back now and then
      for (int r = 0; r < 16; ++r) { c0[r] *= 1e-6f; c1[r] *= 1e-6f; }
    }
  }
  float s = (float)junk;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, const u32x4* src, float* d, int n_cu, double target_ms) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 2000;
  float ms = 0;
  for (int pass = 0; pass < 3; ++pass) {                             // calibrate, then two measured launches (the second is reported)
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(n_cu), dim3(256), 0, 0, src, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (pass == 0) iters = (int)(iters * target_ms / ms);
  }
  const double mfma_per_simd = (double)iters * 16;
  const double flops = (double)n_cu * 4 * mfma_per_simd * 32 * 32 * 16 * 2;
  printf("%-44s %8.3f ms  %8.1f TFLOP/s  implied clock (32 cyc/MFMA) %.3f GHz  frac of 2.5 PF %.3f\n", name, ms, flops / ms / 1e9,
         mfma_per_simd * 32 / (ms * 1e-3) / 1e9, flops / ms / 1e9 / 2500.0);
}

int main(int argc, char** argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 25.0;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  std::vector<unsigned> h(2048 * 4);
  srand(7);
  for (auto& w : h) {                                                 // two random bf16 in [-1, 1): sign, exponent 118..126, 7 mantissa bits
    unsigned lo = ((rand() & 1) << 15) | ((118 + rand() % 9) << 7) | (rand() & 127);
    unsigned hi = ((rand() & 1) << 15) | ((118 + rand() % 9) << 7) | (rand() & 127);
    w = lo | (hi << 16);
  }
  u32x4* src; float* d;
  hipMalloc(&src, h.size() * 4); hipMalloc(&d, n_cu * 256 * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("%s: %d CUs, launches of ~%.0f ms, one wave per SIMD, two MFMA chains per wave\n", p.name, n_cu, target_ms);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("mode 0: zero operands", src, d, n_cu, target_ms);
    run<1>("mode 1: random bf16 operands", src, d, n_cu, target_ms);
    run<2>("mode 2: random + ds_read_b128 / 2 MFMA", src, d, n_cu, target_ms);
    run<3>("mode 3: mode 2 + 3 VALU / MFMA", src, d, n_cu, target_ms);
    run<4>("mode 4: random A, half-zero B (ReLU-like)", src, d, n_cu, target_ms);
    run<5>("mode 5: mode 4 + ds_read_b128 / 2 MFMA", src, d, n_cu, target_ms);
    run<6>("mode 6: mode 4 + ds_read_b128 / 4 MFMA", src, d, n_cu, target_ms);
    run<7>("mode 7: mode 6 + B of every 2nd MFMA from LDS", src, d, n_cu, target_ms);
  }
  return 0;
}
