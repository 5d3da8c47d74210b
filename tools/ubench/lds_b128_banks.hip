// micro-benchmark: which lanes of a ds_write_b128 / ds_read_b128 / ds_write_b64 does the LDS of gfx950 serve together, i.e. when do two
// lanes of one instruction conflict?  Every pattern below is a PERMUTATION of the 64 lanes over one contiguous 1 KB (b128) or 512 B
// (b64) region, so a wave-wide view sees no conflict in any of them; what differs is which lanes share a 256-byte bank window offset.
//   identity      lane L -> 16 L
//   stride128_g8  lanes 8g .. 8g+7 all 128 B apart  (conflicts iff the hardware serves >= 2 of them in one pass)
//   half_swap     the staging planes of sn_mlp_fwd_bf16_t.hip: lane (j, h) -> 32 j + 16 (h ^ ((j >> 3) & 1))
//   j16_alias     lane (j, h) -> 32 j + 16 h          (rows j, j + 8 share banks within 16 lanes)
//   pitch144      lane (j, h) -> 144 j + 16 h         (the fp32 / bf16x3 staging tile, XPOSE_PITCH 36)
//   pitch136      lane (j, h) -> 136 j + 16 h
// one PMC pass per build over this program gives SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel (pattern index in the kernel name).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_b128_banks tools/ubench/lds_b128_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PAT, int OP>      // OP 0: ds_write_b128, 1: ds_read_b128, 2: ds_write_b64
__global__ void __launch_bounds__(64) k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32 * 144 + 64];
  const int L = threadIdx.x, j = L & 31, h = L >> 5;
  const int W = OP == 2 ? 8 : 16;
  int off;
  if (PAT == 0) off = W * L;
  else if (PAT == 1) off = W * ((L & 7) * 8 + (L >> 3));
  else if (PAT == 2) off = 2 * W * j + W * (h ^ ((j >> 3) & 1));
  else if (PAT == 3) off = 2 * W * j + W * h;
  else if (PAT == 4) off = 144 * j + W * h;
  else off = 136 * j + W * h;
  const unsigned a = (unsigned)(size_t)lds + off;
  f32x4 v = {1.0f * L, 2.0f, 3.0f, 4.0f};
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v) : "memory");
    else if (OP == 1) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    else { f32x2 w = {v[0], v[1]}; asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(w) : "memory"); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (v[0] == -1.0f) out[L] = v[1];
}

template <int PAT, int OP>
static void run(float* d, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL((k<PAT, OP>), dim3(1024), dim3(64), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<PAT, OP>), dim3(1024), dim3(64), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("op %d pattern %d %-14s %8.3f ms\n", OP, PAT, name, ms);
}
int main() {
  float* d; hipMalloc(&d, 4096);
  const char* n[6] = {"identity", "stride128_g8", "half_swap", "j16_alias", "pitch144", "pitch136"};
#define ROW(OP) run<0, OP>(d, n[0]); run<1, OP>(d, n[1]); run<2, OP>(d, n[2]); run<3, OP>(d, n[3]); run<4, OP>(d, n[4]); run<5, OP>(d, n[5]);
  ROW(0) ROW(1) ROW(2)
  return 0;
}
