// sn_dw_common.h -- task / plan records shared by the weight-gradient kernels (sn_dw.hip: every mode and shape, compiler-
// scheduled; sn_dw_f32.hip: the fp32 256x256 problems, hand-scheduled inner loop).
#pragma once
#include "sn_device.h"
#include "sn_launch.h"

namespace snd {

constexpr int KB = 16;                          // points per staged chunk
constexpr int DW_LDS_BYTES = 4 * KB * (256 + 256) * 4;      // 131072: LDS ring, 4 chunks of the widest fp32 problem (depth per mode: run_task)

struct Task {                                   // 64 bytes, built on the host (sinnerf_amd/autograd.py)
  const void* a;                                // G  + column offset (fp32, or bf16 with 0x200)
  const void* b;                                // X  + column offset (fp32; bf16 with 0x200 except in variants 1 / 3)
  float* c;                                     // partial dW  [M_wg][ldc]
  float* bias;                                  // partial db  [M_wg] or nullptr
  long k0, k1;                                  // point range: (k1-k0) % 16 == 0, rows [k0,k1) readable (callers zero-pad G)
  int lda, ldb;
  int ldc, variant;                             // M x N: 0 = 256x256, 1 = 256x64, 2 = 128x256, 3 = 128x64, 4 = 32x256, 5 = 32x128;
                                                // 6 / 7 = 1 / 3 with the embedded inputs stored as bf16 (sn_dw_narrow_bf16.hip only);
                                                // | 0x100: bf16 operands (mixed-precision training), fp32 accumulate;
                                                // | 0x200: G and the 256-wide activations are stored as bf16 (the embedded
                                                //   inputs of variants 1 / 3 stay fp32); lda / ldb stay in ELEMENTS;
                                                // | 0x400 (with 0x100): bf16x3 -- 3-term (hi, lo) split, fp32-level; the 256-wide operands of
                                                //   variants 0 (a, b), 1 (a), 2 (b), 4 (b) are rows of the SPLIT state (sn_layout.h "x3 state"),
                                                //   everything else fp32 and split in registers
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

// The whole network's contractions described by value (kernel arguments: no host-built table, no H2D copy, capturable in
// a HIP graph): problem q = one dW = G^T X, split into ns K-ranges of `per` points, tasks [first, first + ns).
struct Prob {
  const void* a;
  const void* b;
  float* c;                                     // ns partial results, M*N floats apart
  float* bias;                                  // ns partial column sums, M floats apart, or nullptr
  int lda, ldb, ldc, variant;
  int ns, per, first, m;
};
constexpr int MAX_PROBS = 14;
struct Plan {
  Prob p[MAX_PROBS];
  long P;                                       // rows (points) of every operand
  int n_probs, n_tasks;
};

// task of workgroup `wg` of a by-value plan
SN_DEV Task task_of(const Plan& plan, int wg) {
  Task t;
  int q = 0;
#pragma unroll 1
  for (int i = 1; i < plan.n_probs; ++i) q = (wg >= plan.p[i].first) ? i : q;      // scalar loads of the kernarg segment
  const Prob& pr = plan.p[q];
  const int j = wg - pr.first;
  t.a = pr.a; t.b = pr.b;
  t.c = pr.c + (long)j * pr.m * pr.ldc;
  t.bias = pr.bias ? pr.bias + (long)j * pr.m : nullptr;
  t.k0 = (long)j * pr.per;
  t.k1 = t.k0 + pr.per < plan.P ? t.k0 + pr.per : plan.P;
  t.lda = pr.lda; t.ldb = pr.ldb; t.ldc = pr.ldc; t.variant = pr.variant;
  return t;
}

}  // namespace snd
