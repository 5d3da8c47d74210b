// Probe of ds_read_b64_tr_b16 (gfx950): checks the lane/element mapping the bf16 weight-gradient kernel relies on
// (csrc/sn_dw.hip).  Model: within a 16-lane group, lane p supplies the address of 4 contiguous bf16 = row (p >> 2),
// columns 4 (p & 3) .. +3 of a 4 x 16 matrix; lane q receives column q, rows 0..3:
//     out[16 G + q][e] = LDS[ addr(lane 16 G + 4 e + (q >> 2)) + 2 (q & 3) ]
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
__global__ void probe(const int* offs, short* out) {
  __shared__ __attribute__((aligned(16))) short buf[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) buf[i] = (short)i;
  __syncthreads();
  const v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)((char*)buf + offs[threadIdx.x]));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = r[e];
}
int main() {
  int fails = 0;
  for (int pattern = 0; pattern < 3; ++pattern) {
    std::vector<int> offs(64);
    for (int l = 0; l < 64; ++l) {
      const int G = l >> 4, p = l & 15;
      if (pattern == 0) offs[l] = l * 8;                                              // lane-linear
      else if (pattern == 1) offs[l] = (4 * G + (p >> 2)) * 512 + (p & 3) * 8 + 64;   // row pitch 512 B
      else offs[l] = ((l * 37) % 61) * 8 * 16 + 8 * (l & 1);                          // arbitrary 8-byte aligned addresses
    }
    int* d_offs; short* d_out;
    hipMalloc(&d_offs, 256); hipMalloc(&d_out, 512);
    hipMemcpy(d_offs, offs.data(), 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_offs, d_out);
    std::vector<short> out(256);
    hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 4; ++e) {
        const int G = l >> 4, q = l & 15;
        const int src = 16 * G + 4 * e + (q >> 2);
        const int want = (offs[src] + 2 * (q & 3)) / 2;
        if (out[l * 4 + e] != (short)want) {
          if (bad < 6) printf("pattern %d lane %d elem %d: got %d want %d\n", pattern, l, e, out[l * 4 + e], want);
          ++bad;
        }
      }
    printf("pattern %d: %s (%d mismatches)\n", pattern, bad ? "MODEL WRONG" : "model holds", bad);
    fails += bad;
    hipFree(d_offs); hipFree(d_out);
  }
  return fails ? 1 : 0;
}
