// micro-benchmark: HBM read bandwidth of MI355X when only PART of every 512-byte row is read (the narrow weight-gradient problems read
// 64 .. 256 B sub-ranges of the 512-byte rows of acts / G): useful bytes per second, plain global_load_dwordx4 streams.
//   hipcc --offload-arch=gfx950 -O3 read_patterns.hip -o read_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vf4 __attribute__((ext_vector_type(4)));

// every thread reads 16-byte pieces; piece q of the launch = (row, k): row = q / PER_ROW, byte offset OFF + 16 (q % PER_ROW) inside a row of PITCH bytes
template <int PITCH, int OFF, int BYTES>
__global__ void __launch_bounds__(256) rd(const char* __restrict__ buf, size_t n_pieces, float* out) {
  constexpr int PER_ROW = BYTES / 16;
  vf4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; q + 7 * stride < n_pieces; q += 8 * stride) {
    vf4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t p = q + u * stride;
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(buf + (p / PER_ROW) * PITCH + OFF + 16 * (p % PER_ROW)));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
template <int PITCH, int OFF, int BYTES>
void run(const char* name, const char* d, size_t rows, float* out) {
  const size_t n_pieces = rows * (BYTES / 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((rd<PITCH, OFF, BYTES>), dim3(4096), dim3(256), 0, 0, d, n_pieces, out);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((rd<PITCH, OFF, BYTES>), dim3(4096), dim3(256), 0, 0, d, n_pieces, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-58s %7.3f ms  %6.2f TB/s useful (%zu MB)\n", name, best, n_pieces * 16.0 / best / 1e9, n_pieces * 16 >> 20);
  fflush(stdout);
}
int main() {
  const size_t rows = (size_t)10 * 524288;          // acts[10][524288][256] bf16 = 2.68 GB
  char* d; float* out;
  if (hipMalloc(&d, rows * 512) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 64);
  hipMemset(d, 0, rows * 512);
  run<512, 0, 512>("whole 512-byte rows", d, rows, out);
  run<512, 0, 256>("first 256 B of every 512-byte row", d, rows, out);
  run<512, 256, 256>("second 256 B of every 512-byte row", d, rows, out);
  run<512, 256, 64>("64 B at offset 256 of every 512-byte row", d, rows, out);
  run<512, 0, 128>("first 128 B of every 512-byte row", d, rows, out);
  run<256, 0, 128>("first 128 B of every 256-byte row", d, rows * 2, out);
  run<256, 0, 256>("whole 256-byte rows", d, rows * 2, out);
  run<64, 0, 64>("whole 64-byte rows (contiguous)", d, rows, out);
  return 0;
}
