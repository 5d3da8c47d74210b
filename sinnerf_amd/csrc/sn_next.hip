// sn_next.hip -- the steps immediately before / after the hot path (SURVEY.md §8f "next" rows), gfx950.
//
//   generate_rays_kernel   datasets/ray_utils.py:86-133 (get_ray_directions + get_rays) and the [o, d, near, far] packing
//                          of the datasets (blender_ray_patch_1image_rot3d.py:201-211, strided patches :487-498):
//                          rays are produced ON the GPU from (c2w, focal) -- no (H*W, 8) host array, no H2D copy.
//   adam_kernel            utils/__init__.py:19-21 (torch.optim.Adam, eps=1e-8) over ONE flat parameter/gradient buffer
//                          (the buffer the single RCCL all-reduce runs on): one elementwise launch per step.
//   loss_partials_kernel / loss_finish_kernel
//                          the losses computed on the rendered rays right after the path: MSE coarse + fine
//                          (losses.py:12-22, nn.MSELoss mean), SmoothL1 depth coarse + fine (models/sinnerf.py:32-42 SL1Loss,
//                          call sites :310-319) and PSNR (metrics.py:5-15), together with dLoss/d{rgb,depth}_{coarse,fine}
//                          -- two small launches instead of ~20 elementwise/reduction launches and their HBM round trips.
// All are HBM-bound elementwise kernels: coalesced, 16-byte accesses where the layout allows.
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

// ---- losses on the rendered rays ---------------------------------------------------------------------------------
// Deterministic two-phase reduction: every block accumulates its grid-stride share in double and writes 5 partials
// (sum (rgb_c-gt)^2, sum (rgb_f-gt)^2, sum sl1(depth_c-gt), sum sl1(depth_f-gt), number of depth elements counted);
// the finish kernel re-reduces the <= LOSS_BLOCKS partials in index order in every block (so each block knows the
// normalisers) and writes the gradients; block 0 writes the scalars.
constexpr int LOSS_BLOCKS = 256;

SN_DEV bool depth_counted(const float* depth_gt, const unsigned char* mask, int mask_mode, long i) {
  // SL1Loss.forward: mask given -> it; mask None and useMask -> depth_gt > 0; useMask=False -> everything
  return mask_mode == 0 ? true : (mask_mode == 1 ? depth_gt[i] > 0.0f : mask[i] != 0);
}
SN_DEV double block_sum(double v, double* sh) {   // all 256 threads; result valid in every thread
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

__global__ void __launch_bounds__(256)
loss_partials_kernel(const float* __restrict__ rgb_c, const float* __restrict__ rgb_f, const float* __restrict__ depth_c,
                     const float* __restrict__ depth_f, const float* __restrict__ rgb_gt,
                     const float* __restrict__ depth_gt, const unsigned char* __restrict__ mask, int mask_mode, long n,
                     double* __restrict__ partials) {
  __shared__ double sh[4];
  double a[5] = {0, 0, 0, 0, 0};
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (rgb_gt != nullptr)
    for (long i = t0; i < 3 * n; i += stride) {
      const float g = rgb_gt[i];
      if (rgb_c != nullptr) { const float d = rgb_c[i] - g; a[0] += (double)(d * d); }
      if (rgb_f != nullptr) { const float d = rgb_f[i] - g; a[1] += (double)(d * d); }
    }
  if (depth_gt != nullptr)
    for (long i = t0; i < n; i += stride) {
      if (!depth_counted(depth_gt, mask, mask_mode, i)) continue;
      const float g = depth_gt[i];
      if (depth_c != nullptr) { const float d = fabsf(depth_c[i] - g); a[2] += (double)(d < 1.0f ? 0.5f * d * d : d - 0.5f); }
      if (depth_f != nullptr) { const float d = fabsf(depth_f[i] - g); a[3] += (double)(d < 1.0f ? 0.5f * d * d : d - 0.5f); }
      a[4] += 1.0;
    }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double v = block_sum(a[k], sh);
    if (threadIdx.x == 0) partials[blockIdx.x * 5 + k] = v;
  }
}

// out[8] = mse_coarse, mse_fine, sl1_coarse, sl1_fine, total = w_rgb (mse_c + mse_f) + w_depth (sl1_c + sl1_f),
//          psnr_coarse, psnr_fine, depth elements counted.  Gradients are those of `total`.
__global__ void __launch_bounds__(256)
loss_finish_kernel(const float* __restrict__ rgb_c, const float* __restrict__ rgb_f, const float* __restrict__ depth_c,
                   const float* __restrict__ depth_f, const float* __restrict__ rgb_gt,
                   const float* __restrict__ depth_gt, const unsigned char* __restrict__ mask, int mask_mode, long n,
                   float w_rgb, float w_depth, const double* __restrict__ partials, int n_partials,
                   float* __restrict__ g_rgb_c, float* __restrict__ g_rgb_f, float* __restrict__ g_depth_c,
                   float* __restrict__ g_depth_f, float* __restrict__ out) {
  __shared__ double sh[4];
  double tot[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double v = (int)threadIdx.x < n_partials ? partials[threadIdx.x * 5 + k] : 0.0;
    tot[k] = block_sum(v, sh);
  }
  const double n_rgb = 3.0 * (double)n, n_d = tot[4];
  const float mse_c = rgb_c && rgb_gt ? (float)(tot[0] / n_rgb) : 0.0f, mse_f = rgb_f && rgb_gt ? (float)(tot[1] / n_rgb) : 0.0f;
  const float sl_c = depth_c && depth_gt ? (float)(tot[2] / n_d) : 0.0f, sl_f = depth_f && depth_gt ? (float)(tot[3] / n_d) : 0.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = mse_c; out[1] = mse_f; out[2] = sl_c; out[3] = sl_f;
    out[4] = w_rgb * (mse_c + mse_f) + w_depth * (sl_c + sl_f);
    out[5] = -10.0f * log10f(mse_c); out[6] = -10.0f * log10f(mse_f);            // metrics.py:14-15
    out[7] = (float)n_d;
  }
  const float s_rgb = (float)((double)w_rgb * 2.0 / n_rgb), s_d = (float)((double)w_depth / n_d);
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (rgb_gt != nullptr)
    for (long i = t0; i < 3 * n; i += stride) {
      const float g = rgb_gt[i];
      if (g_rgb_c != nullptr && rgb_c != nullptr) g_rgb_c[i] = s_rgb * (rgb_c[i] - g);
      if (g_rgb_f != nullptr && rgb_f != nullptr) g_rgb_f[i] = s_rgb * (rgb_f[i] - g);
    }
  if (depth_gt != nullptr)
    for (long i = t0; i < n; i += stride) {
      const bool on = depth_counted(depth_gt, mask, mask_mode, i);
      const float g = depth_gt[i];
      if (g_depth_c != nullptr && depth_c != nullptr) {
        const float d = depth_c[i] - g;
        g_depth_c[i] = on ? s_d * (fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f)) : 0.0f;
      }
      if (g_depth_f != nullptr && depth_f != nullptr) {
        const float d = depth_f[i] - g;
        g_depth_f[i] = on ? s_d * (fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f)) : 0.0f;
      }
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

extern "C" long sn_render_loss_workspace_bytes_impl() { return (long)snx::LOSS_BLOCKS * 5 * sizeof(double); }
extern "C" int sn_render_loss_launch(const float* rgb_c, const float* rgb_f, const float* depth_c, const float* depth_f,
                                     const float* rgb_gt, const float* depth_gt, const unsigned char* mask, int mask_mode,
                                     long n, float w_rgb, float w_depth, float* g_rgb_c, float* g_rgb_f, float* g_depth_c,
                                     float* g_depth_f, void* workspace, float* out, hipStream_t stream) {
  using namespace snx;
  long blocks = (3 * n + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > LOSS_BLOCKS) blocks = LOSS_BLOCKS;
  double* partials = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(loss_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rgb_c, rgb_f, depth_c, depth_f,
                     rgb_gt, depth_gt, mask, mask_mode, n, partials);
  hipLaunchKernelGGL(loss_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rgb_c, rgb_f, depth_c, depth_f,
                     rgb_gt, depth_gt, mask, mask_mode, n, w_rgb, w_depth, partials, (int)blocks, g_rgb_c, g_rgb_f,
                     g_depth_c, g_depth_f, out);
  return (int)hipGetLastError();
}
