// micro-benchmark: HBM write bandwidth of MI355X as a function of the address pattern (no compute).  1024 waves (256 workgroups x 4, one
// workgroup per CU unless noted), every wave issues global_store_dwordx4 (1 KB per instruction); the patterns differ in where instruction t
// of wave w lands.  Total 2 GiB per launch.       hipcc --offload-arch=gfx950 -O3 write_patterns.hip -o write_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vf4 __attribute__((ext_vector_type(4)));

// PAT 0: one sequential stream per wave (2 MB windows)            f = w * 2 MB + t KB
//     1: chip-wide sweep                                         f = (t * NW + w) KB
//     2: 32 KB bursts per wave, bursts swept                     f = ((t / 32) * NW + w) * 32 KB + (t % 32) KB
//     3: as 2 over 10 arrays in turn (the activation slots)      burst b -> array b % 10
//     4: 8 pieces of 128 B at 512 B pitch, 4 instructions fill 4 KB, 32 KB bursts per wave (the training kernels' row stores)
//     5: as 4 over 10 arrays in turn
template <int PAT, int NT>
__global__ void __launch_bounds__(256) wr(char* buf, int n_t, int nw) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const vf4 v = {1.f, 2.f, 3.f, 4.f};
  const size_t KB = 1024;
#pragma unroll 4
  for (int t = 0; t < n_t; ++t) {
    size_t f;
    if (PAT == 0) f = (size_t)w * n_t * KB + t * KB + lane * 16;
    else if (PAT == 1) f = ((size_t)t * nw + w) * KB + lane * 16;
    else if (PAT == 2) f = ((size_t)(t >> 5) * nw + w) * 32 * KB + (t & 31) * KB + lane * 16;
    else if (PAT == 3) {
      const int b = t >> 5, arr = b % 10, bb = b / 10;        // n_t / 32 bursts = 10 arrays x (n_t / 320)
      f = (size_t)arr * ((size_t)n_t / 10 * nw * KB) + ((size_t)bb * nw + w) * 32 * KB + (t & 31) * KB + lane * 16;
    } else {
      const int b = t >> 5, u = t & 31;                         // u: 8 groups of 4 KB x 4 column pieces
      const size_t in_burst = (size_t)(u >> 2) * 4096 + (lane >> 3) * 512 + (u & 3) * 128 + (lane & 7) * 16;
      if (PAT == 4) f = ((size_t)b * nw + w) * 32 * KB + in_burst;
      else {
        const int arr = b % 10, bb = b / 10;
        f = (size_t)arr * ((size_t)n_t / 10 * nw * KB) + ((size_t)bb * nw + w) * 32 * KB + in_burst;
      }
    }
    vf4* p = reinterpret_cast<vf4*>(buf + f);
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
  }
}
template <int PAT, int NT>
void run(const char* name, char* d, int blocks, int threads) {
  const int nw = blocks * threads / 64;
  const int n_t = (int)(((size_t)1 << 31) / 1024 / nw / 320 * 320);
  const size_t bytes = (size_t)n_t * nw * 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((wr<PAT, NT>), dim3(blocks), dim3(threads), 0, 0, d, n_t, nw);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((wr<PAT, NT>), dim3(blocks), dim3(threads), 0, 0, d, n_t, nw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-64s %s %4d x %3d  %7.3f ms  %.2f TB/s\n", name, NT ? "nt   " : "plain", blocks, threads, best, bytes / best / 1e9);
  fflush(stdout);
}
int main() {
  char* d; if (hipMalloc(&d, ((size_t)1 << 31) + (64 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0, (size_t)1 << 31);
#define ALL(PAT, NAME) run<PAT, 1>(NAME, d, 256, 256); run<PAT, 0>(NAME, d, 256, 256); run<PAT, 1>(NAME, d, 512, 256); run<PAT, 1>(NAME, d, 1024, 256);
  ALL(0, "0 one sequential stream per wave")
  ALL(1, "1 chip-wide sweep, 1 KB per wave")
  ALL(2, "2 32 KB bursts per wave, swept")
  ALL(3, "3 32 KB bursts, 10 arrays in turn")
  ALL(4, "4 row pieces (8 x 128 B at 512 B pitch), 32 KB bursts")
  ALL(5, "5 row pieces, 10 arrays in turn")
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemsetAsync(d, 0, (size_t)1 << 31, 0);
    hipEventRecord(e0); hipMemsetAsync(d, 1, (size_t)1 << 31, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.3f ms  %.2f TB/s\n", "hipMemsetAsync (2 GiB)", ms, ((size_t)1 << 31) / ms / 1e9);
  }
  return 0;
}
