// sn_render.hip -- the per-ray stages of render_rays around the MLP (gfx950), one 64-lane wave per ray.
//
//   sample_coarse_kernel   models/rendering.py:264-282   stratified depths (+ perturb)
//   composite_fwd_kernel   models/rendering.py:215-246   deltas, alpha, transmittance scan, weights, rgb/depth
//   sample_pdf_kernel      models/rendering.py:15-61 + :310-315  cdf scan, per-lane binary search, inverse-CDF
//                          lerp, then the merge (torch.sort of cat([z_coarse, z_fine])) as an in-LDS rank sort
//
// These stages are HBM/latency-bound elementwise+scan work over (N_rays, S) arrays: coalesced loads, wave
// scans through DPP/shuffles, no LDS except the small per-ray tables of the sampler.
//
// Numerics follow the reference op by op (one rounding per torch op; compiled with -ffp-contract=off).
// torch's CPU cumsum/cumprod accumulate fp32 inputs in double and round per element; the scans here do the
// same (fp64 scan, one rounding), which also makes the result independent of the scan tree.
#include "sn_device.h"

namespace snr {

SN_DEV float linspace01(int i, int n) {
  // torch.linspace(0,1,n) fp32: step = 1/(n-1); lower half i*step, upper half fma(-(n-1-i), step, 1)
  if (n <= 1) return 0.0f;
  const float step = __fdiv_rn(1.0f, (float)(n - 1));
  return (i < n / 2) ? __fmul_rn((float)i, step) : __builtin_fmaf(-(float)(n - 1 - i), step, 1.0f);
}

SN_DEV float coarse_z(float near, float far, float t, int use_disp) {
  const float omt = __fsub_rn(1.0f, t);
  if (!use_disp) return __fadd_rn(__fmul_rn(near, omt), __fmul_rn(far, t));               // rendering.py:268
  const float a = __fmul_rn(__fdiv_rn(1.0f, near), omt), b = __fmul_rn(__fdiv_rn(1.0f, far), t);
  return __fdiv_rn(1.0f, __fadd_rn(a, b));                                                 // rendering.py:270
}

__global__ void __launch_bounds__(256)
sample_coarse_kernel(const float* __restrict__ rays, long n_rays, int S, int use_disp, float perturb,
                     const float* __restrict__ perturb_rand, float* __restrict__ z_out) {
  const long total = n_rays * (long)S;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long ray = idx / S;
    const int i = (int)(idx - ray * S);
    const float near = rays[ray * 8 + 6], far = rays[ray * 8 + 7];
    const float z = coarse_z(near, far, linspace01(i, S), use_disp);
    float r = z;
    if (perturb > 0.0f) {                                                                   // rendering.py:274-282
      float lower = z, upper = z;
      if (i > 0) lower = __fmul_rn(0.5f, __fadd_rn(coarse_z(near, far, linspace01(i - 1, S), use_disp), z));
      if (i < S - 1) upper = __fmul_rn(0.5f, __fadd_rn(z, coarse_z(near, far, linspace01(i + 1, S), use_disp)));
      const float pr = __fmul_rn(perturb, perturb_rand[idx]);
      r = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), pr));
    }
    z_out[idx] = r;
  }
}

// ---- wave primitives -----------------------------------------------------------------------------
SN_DEV double wave_incl_scan_mul(double v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(v, off, 64);
    if (lane >= off) v *= o;
  }
  return v;
}
SN_DEV double wave_incl_scan_add(double v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  return v;
}
SN_DEV double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

SN_DEV void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- compositor forward ----------------------------------------------------------------------------
// One wave per ray; lane l owns the C consecutive samples [l*C, (l+1)*C).  HAS_RGB: raw is (N,S,4) [rgb,sigma]
// (nerf.py:146) else (N,S) raw sigma (weights_only path, rendering.py:238-239).
template <int C, bool HAS_RGB>
__global__ void __launch_bounds__(256)
composite_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays,
                     const float* __restrict__ noise, float noise_std, long n_rays, int S, int white_back,
                     float* __restrict__ rgb_out, float* __restrict__ depth_out, float* __restrict__ w_out) {
  const int lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const float dx = rays[ray * 8 + 3], dy = rays[ray * 8 + 4], dz = rays[ray * 8 + 5];
  // torch.norm(dir, dim=-1): sqrt(sum of squares)                                          rendering.py:222
  const float dnorm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const long base = ray * (long)S;
  const int i0 = lane * C;

  float z[C + 1], sg[C], cr[C], cg[C], cb[C];
#pragma unroll
  for (int c = 0; c < C + 1; ++c) {
    const int i = i0 + c;
    z[c] = (i < S) ? z_vals[base + i] : 0.0f;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = i0 + c;
    if (i < S) {
      if (HAS_RGB) {
        const float4 v = reinterpret_cast<const float4*>(raw)[base + i];
        cr[c] = v.x; cg[c] = v.y; cb[c] = v.z; sg[c] = v.w;
      } else {
        sg[c] = raw[base + i];
        cr[c] = cg[c] = cb[c] = 0.0f;
      }
    } else {
      sg[c] = cr[c] = cg[c] = cb[c] = 0.0f;
    }
  }
  float alpha[C];
  double fprod = 1.0;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = i0 + c;
    float a = 0.0f, f = 1.0f;
    if (i < S) {
      float delta = (i < S - 1) ? __fsub_rn(z[c + 1], z[c]) : 1e10f;                        // :215-218
      delta = __fmul_rn(delta, dnorm);                                                      // :222
      float s = sg[c];
      if (noise != nullptr) s = __fadd_rn(s, __fmul_rn(noise[base + i], noise_std));        // :224
      s = fmaxf(s, 0.0f);
      a = __fsub_rn(1.0f, expf(__fmul_rn(-delta, s)));                                      // :228
      f = __fadd_rn(__fsub_rn(1.0f, a), 1e-10f);                                            // :229-231
    }
    alpha[c] = a;
    fprod *= (double)f;
  }
  // exclusive transmittance: T_i = prod_{j<i} f_j (fp64 scan, rounded per element like torch's cumprod)  :233
  const double incl = wave_incl_scan_mul(fprod, lane);
  double t = __shfl_up(incl, 1, 64);
  if (lane == 0) t = 1.0;
  double wsum = 0.0, sr = 0.0, sgc = 0.0, sb = 0.0, sd = 0.0;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = i0 + c;
    const float w = __fmul_rn(alpha[c], (float)t);                                          // :232-234
    if (i < S) {
      w_out[base + i] = w;
      wsum += (double)w;
      if (HAS_RGB) {
        sr += (double)__fmul_rn(w, cr[c]);                                                  // :242
        sgc += (double)__fmul_rn(w, cg[c]);
        sb += (double)__fmul_rn(w, cb[c]);
        sd += (double)__fmul_rn(w, z[c]);                                                   // :243
      }
      const float a = alpha[c];
      t *= (double)__fadd_rn(__fsub_rn(1.0f, a), 1e-10f);
    }
  }
  if (HAS_RGB) {
    wsum = wave_sum(wsum); sr = wave_sum(sr); sgc = wave_sum(sgc); sb = wave_sum(sb); sd = wave_sum(sd);
    if (lane == 0) {
      float r = (float)sr, g = (float)sgc, b = (float)sb;
      if (white_back) {                                                                     // :245-246
        const float ws = (float)wsum;
        r = __fsub_rn(__fadd_rn(r, 1.0f), ws);
        g = __fsub_rn(__fadd_rn(g, 1.0f), ws);
        b = __fsub_rn(__fadd_rn(b, 1.0f), ws);
      }
      rgb_out[ray * 3 + 0] = r; rgb_out[ray * 3 + 1] = g; rgb_out[ray * 3 + 2] = b;
      depth_out[ray] = (float)sd;
    }
  }
}

// ---- compositor backward ----------------------------------------------------------------------------
// Autograd of rendering.py:215-246 w.r.t. raw = [rgb, sigma] (z_vals / deltas are data: no gradient).
//   G_i      = g_rgb . c_i + g_depth * z_i + g_w_i - [white_back] * sum(g_rgb)        (dL/dw_i)
//   dL/dc_i  = w_i * g_rgb
//   dL/da_i  = G_i T_i - (sum_{k>i} G_k w_k) / f_i        (cumprod backward, f_i = 1 - a_i + 1e-10 > 0)
//   dL/ds_i  = dL/da_i * delta_i * exp(-delta_i s_i) * [sigma_i + noise_i > 0]
// One wave per ray, same sample-to-lane mapping as the forward; suffix sum as an fp64 wave scan.
template <int C>
__global__ void __launch_bounds__(256)
composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays,
                     const float* __restrict__ noise, float noise_std, long n_rays, int S, int white_back,
                     const float* __restrict__ g_rgb, const float* __restrict__ g_depth, const float* __restrict__ g_w,
                     float* __restrict__ g_raw) {
  const int lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const float dx = rays[ray * 8 + 3], dy = rays[ray * 8 + 4], dz = rays[ray * 8 + 5];
  const float dnorm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const long base = ray * (long)S;
  const int i0 = lane * C;
  const float gr = g_rgb ? g_rgb[ray * 3 + 0] : 0.0f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.0f,
              gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.0f, gd = g_depth ? g_depth[ray] : 0.0f;
  const float gwhite = white_back ? (gr + gg + gb) : 0.0f;

  float zc[C + 1];
#pragma unroll
  for (int c = 0; c < C + 1; ++c) zc[c] = (i0 + c < S) ? z_vals[base + i0 + c] : 0.0f;
  float alpha[C], ex[C], dl[C], Gi[C], fi[C];
  bool pos[C];
  double fprod = 1.0;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = i0 + c;
    float a = 0.0f, f = 1.0f, e = 1.0f, d = 0.0f, G = 0.0f;
    bool ps = false;
    if (i < S) {
      const float4 v = reinterpret_cast<const float4*>(raw)[base + i];
      d = (i < S - 1) ? __fsub_rn(zc[c + 1], zc[c]) : 1e10f;
      d = __fmul_rn(d, dnorm);
      float sp = v.w;
      if (noise != nullptr) sp = __fadd_rn(sp, __fmul_rn(noise[base + i], noise_std));
      ps = sp > 0.0f;
      e = expf(__fmul_rn(-d, fmaxf(sp, 0.0f)));
      a = __fsub_rn(1.0f, e);
      f = __fadd_rn(__fsub_rn(1.0f, a), 1e-10f);
      G = gr * v.x + gg * v.y + gb * v.z + gd * zc[c] - gwhite;
      if (g_w != nullptr) G += g_w[base + i];
    }
    alpha[c] = a; ex[c] = e; dl[c] = d; Gi[c] = G; fi[c] = f; pos[c] = ps;
    fprod *= (double)f;
  }
  const double incl = wave_incl_scan_mul(fprod, lane);
  double t = __shfl_up(incl, 1, 64);
  if (lane == 0) t = 1.0;
  float w[C], T[C];
  double local = 0.0;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    T[c] = (float)t;
    w[c] = __fmul_rn(alpha[c], T[c]);
    local += (double)Gi[c] * (double)w[c];
    t *= (double)fi[c];
  }
  const double total = wave_sum(local);
  const double prefix_incl = wave_incl_scan_add(local, lane);
  double suffix = total - prefix_incl;            // sum over samples owned by higher lanes
#pragma unroll
  for (int c = C - 1; c >= 0; --c) {
    const int i = i0 + c;
    if (i < S) {
      const float g_alpha = (float)((double)Gi[c] * (double)T[c] - suffix / (double)fi[c]);
      const float g_sigma = pos[c] ? g_alpha * dl[c] * ex[c] : 0.0f;
      float4 o;
      o.x = w[c] * gr; o.y = w[c] * gg; o.z = w[c] * gb; o.w = g_sigma;
      reinterpret_cast<float4*>(g_raw)[base + i] = o;
    }
    suffix += (double)Gi[c] * (double)w[c];
  }
}

// ---- importance sampler + merge --------------------------------------------------------------------
// One wave per ray.  LDS per wave: cdf[S-1] | bins[S-1] | keys[S+NI] | merged[S+NI] (FROM_Z only).
// FROM_Z = true : render_rays path -- bins are the mid points of z_vals (N,S), pdf weights are weights[:,1:-1],
//                 the S+NI merged depths are written sorted.
// FROM_Z = false: plain sample_pdf(bins (N,S-1), weights (N,S-2)) of rendering.py:15-61, no merge.
template <bool FROM_Z>
__global__ void __launch_bounds__(256)
sample_pdf_kernel(const float* __restrict__ z_vals, const float* __restrict__ weights, const float* __restrict__ u_in,
                  long n_rays, int S, int NI, float eps, float* __restrict__ z_fine_out, float* __restrict__ z_merged_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const long ray = (long)blockIdx.x * 4 + wv;
  const int M = S - 2;                       // pdf bins   (weights[:, 1:-1])      rendering.py:311
  const int L = S - 1;                       // cdf / bins entries
  const int n = S + NI;
  const int per_wave = 2 * L + n + (FROM_Z ? n : 0);
  float* cdf = reinterpret_cast<float*>(smem) + (long)wv * per_wave;
  float* bins = cdf + L;
  float* keys = bins + L;
  float* merged = keys + n;                  // the sorted depths are scattered here and leave as coalesced rows
  if (ray >= n_rays) return;                 // whole wave exits together (no block-level barrier below)
  const long base = FROM_Z ? ray * (long)S : ray * (long)M - 1;   // weights[base + k + 1] = pdf weight k

  // pdf -> cdf: lanes own C consecutive bins
  const int C = (M + 63) / 64;
  double local = 0.0;
  for (int c = 0; c < C; ++c) {
    const int k = lane * C + c;
    if (k < M) local += (double)__fadd_rn(weights[base + k + 1], eps);                       // :30
  }
  const float tot = (float)wave_sum(local);                                                 // :32 (sum)
  double run = 0.0;
  for (int c = 0; c < C; ++c) {
    const int k = lane * C + c;
    if (k < M) run += (double)__fdiv_rn(__fadd_rn(weights[base + k + 1], eps), tot);        // pdf  :32
  }
  const double incl = wave_incl_scan_add(run, lane);
  double acc = incl - run;                   // exclusive prefix of this lane's first bin
  for (int c = 0; c < C; ++c) {
    const int k = lane * C + c;
    if (k < M) {
      acc += (double)__fdiv_rn(__fadd_rn(weights[base + k + 1], eps), tot);
      cdf[k + 1] = (float)acc;                                                              // :34-36
    }
  }
  if (lane == 0) cdf[0] = 0.0f;
  if (FROM_Z) {
    for (int k = lane; k < L; k += 64)                                                      // z_vals_mid :310
      bins[k] = __fmul_rn(0.5f, __fadd_rn(z_vals[base + k], z_vals[base + k + 1]));
    for (int k = lane; k < S; k += 64) keys[k] = z_vals[base + k];
  } else {
    for (int k = lane; k < L; k += 64) bins[k] = z_vals[ray * (long)L + k];
  }
  // Waves of a block work on different rays and never exchange data: only this wave's own LDS writes must
  // be ordered before its (cross-lane) reads.  The LDS executes one wave's instructions in order, so a
  // wavefront-scope fence (compiler ordering + lgkmcnt) is sufficient -- no s_barrier.
  wave_lds_sync();

  for (int i = lane; i < NI; i += 64) {
    const float u = (u_in != nullptr) ? u_in[ray * (long)NI + i] : linspace01(i, NI);       // :40-43
    // searchsorted(cdf, u, right=True): number of entries <= u                             // :46
    int lo = 0, hi = L;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int inds = lo;
    const int below = max(inds - 1, 0);                                                     // :47
    const int above = min(inds, M);                                                         // :48
    const float c0 = cdf[below], c1 = cdf[above], b0 = bins[below], b1 = bins[above];
    float denom = __fsub_rn(c1, c0);                                                        // :54
    if (denom < eps) denom = 1.0f;                                                          // :56
    const float smp = __fadd_rn(b0, __fmul_rn(__fdiv_rn(__fsub_rn(u, c0), denom), __fsub_rn(b1, b0)));   // :59-60
    keys[S + i] = smp;
    if (z_fine_out != nullptr) z_fine_out[ray * (long)NI + i] = smp;
  }
  if (!FROM_Z) return;
  wave_lds_sync();
  // sort(cat([z_vals, z_samples])) of rendering.py:315 (values only matter: torch.sort(...)[0]).  The coarse depths are
  // ascending by construction (linspace + stratified perturb, :264-282); the samples are ascending whenever u is (det=True:
  // u = linspace -- every eval render) and usually not with random u.  Two sorted lists merge by rank:
  //     rank(coarse i) = i + #{samples <  z_i}       rank(sample m) = m + #{coarse <= z_m}
  // -- exactly the permutation of the stable rank sort below (ties: coarse entries come first in the cat), one binary search
  // per key instead of n compares.  A wave vote picks the path per ray.
  bool ascending = true;
  for (int i = lane; i < NI - 1; i += 64) ascending = ascending && (keys[S + i] <= keys[S + i + 1]);
  for (int i = lane; i < S - 1; i += 64) ascending = ascending && (keys[i] <= keys[i + 1]);
  if (__all(ascending)) {
    const float* kf = keys + S;
    for (int i = lane; i < S; i += 64) {
      const float k = keys[i];
      int lo = 0, hi = NI;                   // #{samples < k}
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (kf[mid] < k) lo = mid + 1; else hi = mid;
      }
      merged[i + lo] = k;
    }
    for (int m = lane; m < NI; m += 64) {
      const float k = kf[m];
      int lo = 0, hi = S;                    // #{coarse <= k}
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] <= k) lo = mid + 1; else hi = mid;
      }
      merged[m + lo] = k;
    }
  } else {
    // general case: stable rank sort of the n keys
    for (int i = lane; i < n; i += 64) {
      const float k = keys[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        const float kj = keys[j];
        rank += (kj < k || (kj == k && j < i)) ? 1 : 0;
      }
      merged[rank] = k;
    }
  }
  wave_lds_sync();
  for (int i = lane; i < n; i += 64) z_merged_out[ray * (long)n + i] = merged[i];
}

}  // namespace snr

// ---------------------------------------------------------------------------------------------------
extern "C" int sn_sample_coarse_launch(const float* rays, long n_rays, int n_samples, int use_disp, float perturb,
                                       const float* perturb_rand, float* z_out, hipStream_t stream) {
  if (n_rays <= 0) return 0;
  if (perturb > 0.0f && perturb_rand == nullptr) return -3;
  const long total = n_rays * (long)n_samples;
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(snr::sample_coarse_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rays, n_rays, n_samples,
                     use_disp, perturb, perturb_rand, z_out);
  return (int)hipGetLastError();
}

extern "C" int sn_composite_forward_launch(const float* raw, int has_rgb, const float* z_vals, const float* rays,
                                           const float* noise, float noise_std, long n_rays, int n_samples,
                                           int white_back, float* rgb, float* depth, float* weights,
                                           hipStream_t stream) {
  using namespace snr;
  if (n_rays <= 0) return 0;
  const int C = (n_samples + 63) / 64;
  if (C < 1 || C > 16) return -4;
  const long blocks = (n_rays + 3) / 4;
  if (blocks > 0x7fffffffL) return -2;
  dim3 grid((unsigned)blocks), block(256);
#define SN_CL(CC)                                                                                                \
  case CC:                                                                                                       \
    if (has_rgb) hipLaunchKernelGGL((composite_fwd_kernel<CC, true>), grid, block, 0, stream, raw, z_vals, rays, noise, \
                                    noise_std, n_rays, n_samples, white_back, rgb, depth, weights);              \
    else hipLaunchKernelGGL((composite_fwd_kernel<CC, false>), grid, block, 0, stream, raw, z_vals, rays, noise,  \
                            noise_std, n_rays, n_samples, white_back, rgb, depth, weights);                      \
    break;
  switch (C) { SN_CL(1) SN_CL(2) SN_CL(3) SN_CL(4) SN_CL(5) SN_CL(6) SN_CL(7) SN_CL(8) SN_CL(9) SN_CL(10) SN_CL(11) SN_CL(12)
               SN_CL(13) SN_CL(14) SN_CL(15) SN_CL(16) }
#undef SN_CL
  return (int)hipGetLastError();
}

extern "C" int sn_composite_backward_launch(const float* raw, const float* z_vals, const float* rays, const float* noise,
                                            float noise_std, long n_rays, int n_samples, int white_back,
                                            const float* g_rgb, const float* g_depth, const float* g_w, float* g_raw,
                                            hipStream_t stream) {
  using namespace snr;
  if (n_rays <= 0) return 0;
  const int C = (n_samples + 63) / 64;
  if (C < 1 || C > 16) return -4;
  const long blocks = (n_rays + 3) / 4;
  if (blocks > 0x7fffffffL) return -2;
  dim3 grid((unsigned)blocks), block(256);
#define SN_CB(CC)                                                                                                \
  case CC:                                                                                                       \
    hipLaunchKernelGGL((composite_bwd_kernel<CC>), grid, block, 0, stream, raw, z_vals, rays, noise, noise_std, n_rays, \
                       n_samples, white_back, g_rgb, g_depth, g_w, g_raw);                                       \
    break;
  switch (C) { SN_CB(1) SN_CB(2) SN_CB(3) SN_CB(4) SN_CB(5) SN_CB(6) SN_CB(7) SN_CB(8) SN_CB(9) SN_CB(10) SN_CB(11) SN_CB(12)
               SN_CB(13) SN_CB(14) SN_CB(15) SN_CB(16) }
#undef SN_CB
  return (int)hipGetLastError();
}

extern "C" int sn_sample_pdf_launch(const float* z_vals, const float* weights, const float* u, long n_rays,
                                    int n_samples, int n_importance, float* z_fine, float* z_merged,
                                    hipStream_t stream) {
  if (n_rays <= 0) return 0;
  if (n_samples < 3 || n_importance < 1) return -5;
  const long blocks = (n_rays + 3) / 4;
  if (blocks > 0x7fffffffL) return -2;
  const size_t lds = 4 * (size_t)(2 * (n_samples - 1) + 2 * (n_samples + n_importance)) * sizeof(float);
  if (lds > 64 * 1024) return -4;
  hipLaunchKernelGGL(snr::sample_pdf_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, stream, z_vals, weights, u,
                     n_rays, n_samples, n_importance, 1e-5f /* render_rays calls sample_pdf with its default eps */, z_fine, z_merged);
  return (int)hipGetLastError();
}

// bins (n_rays, n_bins+1), weights (n_rays, n_bins): the standalone sample_pdf of rendering.py:15-61
extern "C" int sn_sample_pdf_bins_launch(const float* bins, const float* weights, const float* u, long n_rays,
                                         int n_bins, int n_importance, float eps, float* samples, hipStream_t stream) {
  if (n_rays <= 0) return 0;
  if (n_bins < 1 || n_importance < 1) return -5;
  const int S = n_bins + 2;
  const long blocks = (n_rays + 3) / 4;
  if (blocks > 0x7fffffffL) return -2;
  const size_t lds = 4 * (size_t)(2 * (S - 1) + S + n_importance) * sizeof(float);
  if (lds > 64 * 1024) return -4;
  hipLaunchKernelGGL(snr::sample_pdf_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, stream, bins, weights, u,
                     n_rays, S, n_importance, eps, samples, (float*)nullptr);
  return (int)hipGetLastError();
}
