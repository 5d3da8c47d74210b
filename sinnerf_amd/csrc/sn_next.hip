// sn_next.hip -- the steps immediately before / after the hot path (SURVEY.md §8f "next" rows), gfx950.
//
//   generate_rays_kernel   datasets/ray_utils.py:86-133 (get_ray_directions + get_rays) and the [o, d, near, far] packing
//                          of the datasets (blender_ray_patch_1image_rot3d.py:201-211, strided patches :487-498):
//                          rays are produced ON the GPU from (c2w, focal) -- no (H*W, 8) host array, no H2D copy.
//   adam_kernel            utils/__init__.py:19-21 (torch.optim.Adam, eps=1e-8) over ONE flat parameter/gradient buffer
//                          (the buffer the single RCCL all-reduce runs on): one elementwise launch per step.
// Both are HBM-bound elementwise kernels: coalesced, 16-byte accesses where the layout allows.
#include "sn_device.h"

namespace snx {

// pixel (x = x0 + ix*sx, y = y0 + iy*sy), ix < pw, iy < ph, row-major over (iy, ix).
__global__ void __launch_bounds__(256)
generate_rays_kernel(const float* __restrict__ c2w, int H, int W, float focal, float near, float far, int x0, int y0,
                     int sx, int sy, int pw, int ph, float* __restrict__ rays) {
  const long total = (long)pw * ph;
  const float r00 = c2w[0], r01 = c2w[1], r02 = c2w[2], tx = c2w[3];
  const float r10 = c2w[4], r11 = c2w[5], r12 = c2w[6], ty = c2w[7];
  const float r20 = c2w[8], r21 = c2w[9], r22 = c2w[10], tz = c2w[11];
  const float hw = (float)W / 2.0f, hh = (float)H / 2.0f;      // Python: W/2, H/2 (true division)
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int iy = (int)(idx / pw), ix = (int)(idx - (long)iy * pw);
    const float i = (float)(x0 + ix * sx), j = (float)(y0 + iy * sy);
    // directions = [(i - W/2)/focal, -(j - H/2)/focal, -1]                                  ray_utils.py:89-91
    const float d0 = __fdiv_rn(__fsub_rn(i, hw), focal);
    const float d1 = -__fdiv_rn(__fsub_rn(j, hh), focal);
    const float d2 = -1.0f;
    // rays_d = directions @ c2w[:, :3].T  (left-to-right accumulation of the 3 products)   ray_utils.py:109
    float4 lo, hi;
    lo.x = tx; lo.y = ty; lo.z = tz;                                                         // rays_o = c2w[:, 3]  :112
    lo.w = __fadd_rn(__fadd_rn(__fmul_rn(d0, r00), __fmul_rn(d1, r01)), __fmul_rn(d2, r02));
    hi.x = __fadd_rn(__fadd_rn(__fmul_rn(d0, r10), __fmul_rn(d1, r11)), __fmul_rn(d2, r12));
    hi.y = __fadd_rn(__fadd_rn(__fmul_rn(d0, r20), __fmul_rn(d1, r21)), __fmul_rn(d2, r22));
    hi.z = near; hi.w = far;
    float4* out = reinterpret_cast<float4*>(rays + idx * 8);
    out[0] = lo; out[1] = hi;
  }
}

// torch.optim.Adam (amsgrad=False, maximize=False), weight_decay folded in L2 style like torch:
//   g += wd*p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
            float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.0f) gi = gi + wd * pi;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

}  // namespace snx

extern "C" int sn_generate_rays_launch(const float* c2w, int H, int W, float focal, float near, float far, int x0, int y0,
                                       int sx, int sy, int pw, int ph, float* rays, hipStream_t stream) {
  const long total = (long)pw * ph;
  if (total <= 0) return 0;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(snx::generate_rays_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, c2w, H, W, focal, near, far,
                     x0, y0, sx, sy, pw, ph, rays);
  return (int)hipGetLastError();
}

extern "C" int sn_adam_step_launch(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                                   float eps, float wd, int step, hipStream_t stream) {
  if (n <= 0) return 0;
  const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(snx::adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, n, lr, b1, b2, eps, wd,
                     (float)bc1, (float)sqrt(bc2));
  return (int)hipGetLastError();
}
