// sn_mlp_common.h -- pieces shared by the fused MLP forward and backward-chain kernels (fp32 path).
#pragma once
#include "sn_device.h"
#include "sn_layout.h"
#include "sn_launch.h"

namespace snk {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

__device__ __forceinline__ int slab_k_rt(int s) {
  return s < 8 ? snl::K_L0 : s < 32 ? snl::K_HID : s < 40 ? snl::K_SKIP : s < 72 ? snl::K_HID : snl::K_DIR;
}

SN_DEV f32x16 load_bias(const float* lds_bias, int s, int h) {
  const f32x4* p = reinterpret_cast<const f32x4*>(lds_bias + s * 32 + h * 16);
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = p[q];
    acc[4 * q + 0] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
  }
  return acc;
}

// Slots this lane half computes for the first layer / skip layer (sn_layout.h: xyz_slot_col).
SN_DEV void embed_xyz(float x, float y, float z, int h, float* xe) {
  const Rev2 px = to_revolutions(x), py = to_revolutions(y), pz = to_revolutions(z);
  const float hs = h ? 32.0f : 1.0f;            // bands 5..9 on the upper lane half
#pragma unroll
  for (int p = 0; p < 15; ++p) {
    const Rev2 pc = (p % 3 == 0) ? px : (p % 3 == 1) ? py : pz;
    const float scale = hs * (float)(1 << (p / 3));
    sincos_rev(pc, scale, xe[2 * p], xe[2 * p + 1]);
  }
  xe[30] = h ? z : x;
  xe[31] = h ? 0.0f : y;
}
SN_DEV void embed_dir(float x, float y, float z, int h, float* de) {
  const Rev2 px = to_revolutions(x), py = to_revolutions(y), pz = to_revolutions(z);
  const float hs = h ? 4.0f : 1.0f;             // bands 2,3 on the upper lane half
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const Rev2 pc = (p % 3 == 0) ? px : (p % 3 == 1) ? py : pz;
    const float scale = hs * (float)(1 << (p / 3));
    sincos_rev(pc, scale, de[2 * p], de[2 * p + 1]);
  }
  de[12] = h ? z : x;
  de[13] = h ? 0.0f : y;
  de[14] = 0.0f;
  de[15] = 0.0f;
}


// Training forward: the embedded inputs of a point go to emb[point][128] in the reference's column order (nerf.py:36-41).
// A lane half owns 30 CONSECUTIVE columns of the xyz embedding (slot e <-> column c0(e) + 30 h, c0 in [3, 33)) and 12 of the
// dir embedding (c0(e) + 12 h, c0 in [3, 15)): written in column order they leave as 16-byte stores (7 + 3 per point and
// half, plus the identity columns) instead of 48 scattered 4-byte ones -- measured 0.16 ms of a 5.0 ms fp32 fine-pass
// forward, the same absolute cost in the 1 ms bf16 one.
constexpr int xyz_col_slot(int k) { return 2 * (3 * (k / 6) + k % 3) + (k % 6) / 3; }     // k = column - 3 - 30 h  ->  slot e < 30
SN_DEV void store_emb_xyz(float* er, const float* f, int h) {
  float* base = er + 3 + 30 * h;
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f[xyz_col_slot(4 * q + i)];
    __builtin_memcpy(base + 4 * q, &v, 16);                         // 4-byte aligned 16-byte store
  }
  base[28] = f[xyz_col_slot(28)];
  base[29] = f[xyz_col_slot(29)];
  er[h ? 2 : 0] = f[30];                                           // x (h = 0) / z (h = 1)
  if (!h) er[1] = f[31];                                           // y
}
constexpr int dir_col_slot(int k) { return 2 * (3 * (k / 6) + k % 3) + (k % 6) / 3; }     // k = column - 3 - 12 h  ->  slot e < 12
SN_DEV void store_emb_dir(float* er64, const float* f, int h) {   // er64 = emb row + 64
  float* base = er64 + 3 + 12 * h;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f[dir_col_slot(4 * q + i)];
    __builtin_memcpy(base + 4 * q, &v, 16);
  }
  er64[h ? 2 : 0] = f[12];                                         // dx / dz
  if (!h) er64[1] = f[13];                                         // dy
}

// bf16-operand kernels: the embedding is rounded to bf16 (8 mantissa bits) before any use, so only the LOWEST band of each
// lane half needs the exact range reduction; the higher bands follow by angle doubling (sin 2a = 2 s c, cos 2a = 1 - 2 s^2):
// 3 instructions per band instead of ~28.  The absolute error grows 2-4x per step (the error in s^2 + c^2 = 1 is amplified
// too): 4 doublings of a ~2e-7 start stay below 1.5e-5 (fp32 emulation over 2e5 arguments), two orders of magnitude inside
// half a bf16 ulp (2e-3 relative) -- a top-band value flips its bf16 rounding less than once in a hundred, lower bands
// correspondingly less, comparable with what the fp32 accumulation order already does.  The fp32 kernels keep the exact path.
SN_DEV void sincos_double(float s, float c, float& s2, float& c2) {
  const float ts = s + s;
  s2 = ts * c;
  c2 = __builtin_fmaf(-ts, s, 1.0f);
}
SN_DEV void embed_xyz_dbl(float x, float y, float z, int h, float* xe) {
  const Rev2 px = to_revolutions(x), py = to_revolutions(y), pz = to_revolutions(z);
  const float hs = h ? 32.0f : 1.0f;            // bands 5..9 on the upper lane half
  sincos_rev(px, hs, xe[0], xe[1]);
  sincos_rev(py, hs, xe[2], xe[3]);
  sincos_rev(pz, hs, xe[4], xe[5]);
#pragma unroll
  for (int p = 3; p < 15; ++p) sincos_double(xe[2 * (p - 3)], xe[2 * (p - 3) + 1], xe[2 * p], xe[2 * p + 1]);
  xe[30] = h ? z : x;
  xe[31] = h ? 0.0f : y;
}
SN_DEV void embed_dir_dbl(float x, float y, float z, int h, float* de) {
  const Rev2 px = to_revolutions(x), py = to_revolutions(y), pz = to_revolutions(z);
  const float hs = h ? 4.0f : 1.0f;             // bands 2,3 on the upper lane half
  sincos_rev(px, hs, de[0], de[1]);
  sincos_rev(py, hs, de[2], de[3]);
  sincos_rev(pz, hs, de[4], de[5]);
#pragma unroll
  for (int p = 3; p < 6; ++p) sincos_double(de[2 * (p - 3)], de[2 * (p - 3) + 1], de[2 * p], de[2 * p + 1]);
  de[12] = h ? z : x;
  de[13] = h ? 0.0f : y;
  de[14] = 0.0f;
  de[15] = 0.0f;
}
#ifndef SN_EMBED_DBL
#define SN_EMBED_DBL 1
#endif
SN_DEV void embed_xyz_bf16(float x, float y, float z, int h, float* xe) {
  if (SN_EMBED_DBL) embed_xyz_dbl(x, y, z, h, xe); else embed_xyz(x, y, z, h, xe);
}
SN_DEV void embed_dir_bf16(float x, float y, float z, int h, float* de) {
  if (SN_EMBED_DBL) embed_dir_dbl(x, y, z, h, de); else embed_dir(x, y, z, h, de);
}

}  // namespace snk
