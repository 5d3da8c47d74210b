// sn_device.h -- device-side helpers shared by the kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define SN_DEV __device__ __forceinline__

// The MLP kernel sources are compiled twice (csrc/Makefile): as they are for NeRF(use_new_activation=True) -- ShiftedSoftplus /
// WidenedSigmoid heads, both reference call sites -- and with -DSN_CLASSIC_HEADS for the constructor's default
// (models/nerf.py:91-100: ReLU after dir_encoding, Sigmoid after rgb).  The second pass lives in its own kernel namespace and
// exports <name>_classic_launch; everything but the two head activations (and their derivatives) is the same code.
// The bf16-operand INFERENCE kernels (sn_mlp_fwd_bf16.hip, sn_mlp_fwd_bf16_v3.hip) are compiled once more with -DSN_OPERAND_F16
// (round 6): the same instruction streams with v_cvt_pk_f16_f32 / v_mfma_f32_32x32x16_f16 -- fp16 operands, 11 significand bits
// instead of 8 at the same matrix rate (SN_DTYPE_F16; its own kernel namespaces and <name>_f16[_classic]_launch entry points).
#ifdef SN_OPERAND_F16
#define SN_CVT_PK "v_cvt_pk_f16_f32"
#define SN_MFMA_16 "v_mfma_f32_32x32x16_f16"
#else
#define SN_CVT_PK "v_cvt_pk_bf16_f32"
#define SN_MFMA_16 "v_mfma_f32_32x32x16_bf16"
#endif
#ifdef SN_CLASSIC_HEADS
#ifdef SN_OPERAND_F16
#define snk snkhc
#define SN_LAUNCH_NAME(base) base##_f16_classic_launch
#else
#define snk snkc
#define SN_LAUNCH_NAME(base) base##_classic_launch
#endif
constexpr bool SN_NEWACT = false;
#else
#ifdef SN_OPERAND_F16
#define snk snkh
#define SN_LAUNCH_NAME(base) base##_f16_launch
#else
#define SN_LAUNCH_NAME(base) base##_launch
#endif
constexpr bool SN_NEWACT = true;
#endif

// ---------------------------------------------------------------------------------------------
// sin/cos of 2^b * x for power-of-two frequencies (reference: models/nerf.py:36-41, torch.sin(freq*x)).
// freq*x is exact in fp32, so the reference value is sin() of an exactly known argument up to ~2^9*|x|.
// Range reduction is done in "revolutions": p = x/(2*pi) as an unevaluated sum ph+pl (error ~2^-48 |p|),
// scaling by 2^b is exact, the integer part is removed exactly, and the remaining |t| <= 0.5 rev is
// reduced to an octant and evaluated with the classic minimax kernels (max err < 1 ulp of the result).
// ---------------------------------------------------------------------------------------------
struct Rev2 { float hi, lo; };

SN_DEV Rev2 to_revolutions(float x) {
  const float C_HI = 0.15915494f;             // fl(1/(2 pi))
  const float C_LO = 6.4206382e-09f;          // 1/(2 pi) - C_HI  (rounded)
  Rev2 r;
  r.hi = x * C_HI;
  r.lo = __builtin_fmaf(x, C_HI, -r.hi) + x * C_LO;
  return r;
}

// sin/cos(2*pi*(scale*(ph+pl))) with scale an exact power of two.
SN_DEV void sincos_rev(Rev2 p, float scale, float& s, float& c) {
  float th = p.hi * scale;                    // exact
  float tl = p.lo * scale;                    // exact
  float fr = th - __builtin_rintf(th);        // exact, in [-0.5, 0.5]
  float t = fr + tl;                          // |t| <= 0.5 (+eps)
  float k = __builtin_rintf(t * 4.0f);        // quadrant -2..2
  float r = __builtin_fmaf(k, -0.25f, t);     // exact: |r| <= 1/8 rev
  float a = r * 6.2831853071795864769f;       // radians, |a| <= pi/4
  float a2 = a * a;
  // sin kernel (Cephes sinf coefficients), cos kernel
  float sp = __builtin_fmaf(a2, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = __builtin_fmaf(sp, a2, -1.6666654611e-1f);
  float sn = __builtin_fmaf(sp * a2, a, a);
  float cp = __builtin_fmaf(a2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = __builtin_fmaf(cp, a2, 4.166664568298827e-2f);
  float cs = __builtin_fmaf(cp * a2, a2, __builtin_fmaf(a2, -0.5f, 1.0f));
  int q = (int)k & 3;                          // two's complement: -1 -> 3, -2 -> 2
  float s1 = (q & 1) ? cs : sn;
  float c1 = (q & 1) ? sn : cs;
  s = (q & 2) ? -s1 : s1;
  c = ((q + 1) & 2) ? -c1 : c1;
}

// ---------------------------------------------------------------------------------------------
// activations (reference: models/activations.py:18-20, :33-35)
// ---------------------------------------------------------------------------------------------
SN_DEV float shifted_softplus(float x) {
  float sx = x - 1.0f;
  float a = fabsf(sx);
  return log1pf(expf(-a)) + (sx >= 0.0f ? sx : 0.0f);
}
SN_DEV float widened_sigmoid(float x) {
  const float SCALE = 1.002f;                  // 1 + 2*EPS, EPS = 1e-3
  return 0.5f * (1.0f + SCALE * tanhf(0.5f * x));
}

SN_DEV float plain_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }      // nn.Sigmoid (nerf.py:100)
// rgb head activation / dir_encoding activation of this compilation pass
SN_DEV float rgb_activation(float x) { return SN_NEWACT ? widened_sigmoid(x) : plain_sigmoid(x); }

SN_DEV int lane_id() { return (int)(threadIdx.x & 63); }
