// sn_api.hip -- the extern "C" boundary declared in include/sinnerf_hip.h + the weight packer.
#include <hip/hip_fp16.h>
#include "../../include/sinnerf_hip.h"
#include "sn_device.h"
#include "sn_layout.h"

extern "C" {
int sn_mlp_forward_f32_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                              int sigma_only, int input_mode, float* out, float* acts, float* emb,
                              long slot_rows, hipStream_t stream);
int sn_mlp_forward_f32g_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld, int sigma_only,
                               int input_mode, float* out, float* acts, float* emb, long slot_rows, hipStream_t stream);
int sn_mlp_forward_f32g_classic_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld, int sigma_only,
                                       int input_mode, float* out, float* acts, float* emb, long slot_rows, hipStream_t stream);
int sn_mlp_forward_f32g_store_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld, int sigma_only,
                                     int input_mode, float* out, float* acts, float* emb, long slot_rows, hipStream_t stream);
int sn_mlp_forward_f32g_store_classic_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld, int sigma_only,
                                             int input_mode, float* out, float* acts, float* emb, long slot_rows, hipStream_t stream);
int sn_mlp_forward_f32_classic_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                              int sigma_only, int input_mode, float* out, float* acts, float* emb,
                              long slot_rows, hipStream_t stream);
int sn_mlp_backward_chain_f32_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw,
                                     long n_points, long slot_rows, float* G, float* g_out, hipStream_t stream);
int sn_mlp_backward_chain_f32_classic_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw,
                                     long n_points, long slot_rows, float* G, float* g_out, hipStream_t stream);
int sn_mlp_backward_chain_f32g_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw, long n_points,
                                      long slot_rows, float* G, float* g_out, hipStream_t stream);
int sn_mlp_backward_chain_f32g_classic_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw,
                                              long n_points, long slot_rows, float* G, float* g_out, hipStream_t stream);
int sn_mlp_backward_chain_bf16_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw,
                                      long n_points, long slot_rows, float* G, float* g_out, int state_bf16,
                                      hipStream_t stream);
int sn_mlp_backward_chain_bf16_classic_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw,
                                      long n_points, long slot_rows, float* G, float* g_out, int state_bf16,
                                      hipStream_t stream);
int sn_mlp_backward_chain_bf16x3_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw,
                                        long n_points, long slot_rows, float* G, float* g_out, hipStream_t stream);
int sn_mlp_backward_chain_bf16x3_t_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw,
                                          long n_points, long slot_rows, float* G, float* g_out, hipStream_t stream);
int sn_mlp_backward_chain_bf16x3_t_classic_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw,
                                                  long n_points, long slot_rows, float* G, float* g_out, hipStream_t stream);
int sn_mlp_backward_chain_bf16x3_classic_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw,
                                                long n_points, long slot_rows, float* G, float* g_out, hipStream_t stream);
int sn_dw_launch(const void* tasks, int n_tasks, hipStream_t stream);
long sn_weight_grads_workspace_bytes_impl(long slot_rows, int dtype, int emb16);
int sn_weight_grads_launch(const void* acts, const float* emb, const void* G, long slot_rows, int dtype, int emb16, void* workspace,
                           float* const* grads, int accumulate, hipStream_t stream);
int sn_generate_rays_launch(const float* c2w, int H, int W, float focal, float near, float far, int x0, int y0, int sx,
                            int sy, int pw, int ph, float* rays, hipStream_t stream);
int sn_adam_step_launch(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                        float wd, int step, hipStream_t stream);
long sn_render_loss_workspace_bytes_impl();
int sn_render_loss_launch(const float* rgb_c, const float* rgb_f, const float* depth_c, const float* depth_f,
                          const float* rgb_gt, const float* depth_gt, const unsigned char* mask, int mask_mode, long n,
                          float w_rgb, float w_depth, float* g_rgb_c, float* g_rgb_f, float* g_depth_c, float* g_depth_f,
                          void* workspace, float* out, hipStream_t stream);
int sn_mlp_forward_bf16_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                               int sigma_only, int input_mode, float* out, float* acts, float* emb, long slot_rows,
                               int state_bf16, hipStream_t stream);
int sn_mlp_forward_bf16_classic_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                               int sigma_only, int input_mode, float* out, float* acts, float* emb, long slot_rows,
                               int state_bf16, hipStream_t stream);
// ... the fp16-operand pass of the two bf16 inference kernels (SN_DTYPE_F16, -DSN_OPERAND_F16)
int sn_mlp_forward_bf16_f16_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                               int sigma_only, int input_mode, float* out, float* acts, float* emb, long slot_rows,
                               int state_bf16, hipStream_t stream);
int sn_mlp_forward_bf16_f16_classic_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                               int sigma_only, int input_mode, float* out, float* acts, float* emb, long slot_rows,
                               int state_bf16, hipStream_t stream);
int sn_mlp_forward_bf16_v3_f16_launch(const void* blob, const float* rays, const float* z_vals, long n_points, int n_samples,
                                  float* out, hipStream_t stream);
int sn_mlp_forward_bf16_v3_f16_classic_launch(const void* blob, const float* rays, const float* z_vals, long n_points, int n_samples,
                                  float* out, hipStream_t stream);
int sn_mlp_forward_bf16x3_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld, int sigma_only,
                                 int input_mode, float* out, float* acts, float* emb, long slot_rows, hipStream_t stream);
int sn_mlp_forward_bf16x3_classic_launch(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                                         int sigma_only, int input_mode, float* out, float* acts, float* emb, long slot_rows,
                                         hipStream_t stream);
int sn_mlp_forward_bf16x3_t_launch(const void* blob, const float* rays, const float* z_vals, long n_points, int n_samples, float* out,
                                   float* acts, float* emb, long slot_rows, hipStream_t stream);
int sn_mlp_forward_bf16x3_t_classic_launch(const void* blob, const float* rays, const float* z_vals, long n_points, int n_samples, float* out,
                                           float* acts, float* emb, long slot_rows, hipStream_t stream);
int sn_mlp_forward_bf16_v3_launch(const void* blob, const float* rays, const float* z_vals, long n_points, int n_samples,
                                  float* out, hipStream_t stream);
int sn_mlp_forward_bf16_v3_classic_launch(const void* blob, const float* rays, const float* z_vals, long n_points, int n_samples,
                                  float* out, hipStream_t stream);
int sn_mlp_forward_bf16_t_launch(const void* blob, const float* rays, const float* z_vals, long n_points, int n_samples, float* out,
                                 float* acts, float* emb, long slot_rows, int emb16, hipStream_t stream);
int sn_mlp_forward_bf16_t_classic_launch(const void* blob, const float* rays, const float* z_vals, long n_points, int n_samples,
                                         float* out, float* acts, float* emb, long slot_rows, int emb16, hipStream_t stream);
int sn_mlp_backward_chain_bf16_t_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw, long n_points,
                                        long slot_rows, float* G, float* g_out, hipStream_t stream);
int sn_mlp_backward_chain_bf16_t_classic_launch(const void* bblob, const float* acts, const float* out_raw, const float* g_raw,
                                                long n_points, long slot_rows, float* G, float* g_out, hipStream_t stream);
int sn_composite_backward_launch(const float* raw, const float* z_vals, const float* rays, const float* noise,
                                 float noise_std, long n_rays, int n_samples, int white_back, const float* g_rgb,
                                 const float* g_depth, const float* g_w, float* g_raw, hipStream_t stream);
int sn_sample_coarse_launch(const float* rays, long n_rays, int n_samples, int use_disp, float perturb,
                            const float* perturb_rand, float* z_out, hipStream_t stream);
int sn_composite_forward_launch(const float* raw, int has_rgb, const float* z_vals, const float* rays,
                                const float* noise, float noise_std, long n_rays, int n_samples, int white_back,
                                float* rgb, float* depth, float* weights, hipStream_t stream);
int sn_sample_pdf_launch(const float* z_vals, const float* weights, const float* u, long n_rays, int n_samples,
                         int n_importance, float* z_fine, float* z_merged, hipStream_t stream);
int sn_sample_pdf_bins_launch(const float* bins, const float* weights, const float* u, long n_rays, int n_bins,
                              int n_importance, float eps, float* samples, hipStream_t stream);
}

namespace {
struct RawPtrs { const float* p[snl::N_RAW]; };

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__global__ void __launch_bounds__(256)
pack_kernel(RawPtrs raw, const snl::PackEntry* __restrict__ table, long n, char* __restrict__ blob, int dtype) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const snl::PackEntry e = table[i];
    float v = 0.0f;
    const bool as_f32 = (e.src == -2) || (e.src >= 0 && (e.src & snl::SRC_F32_FLAG));
    if (e.src >= 0) {
      const int t = (e.src >> 20) & 0x1ff, off = e.src & 0xfffff;
      const float* src = raw.p[0];
#pragma unroll
      for (int k = 1; k < snl::N_RAW; ++k) src = (t == k) ? raw.p[k] : src;
      v = src[off];
    }
    if (dtype == snl::DT_F32 || as_f32) *reinterpret_cast<float*>(blob + e.dst) = v;
    else if (e.src >= 0 && (e.src & snl::SRC_LO_FLAG))             // bf16x3: the remainder of the RNE high part, itself RNE
      *reinterpret_cast<unsigned short*>(blob + e.dst) = f32_to_bf16_rne(__fsub_rn(v, __uint_as_float((unsigned)f32_to_bf16_rne(v) << 16)));
    else if (dtype == snl::DT_F16) *reinterpret_cast<__half*>(blob + e.dst) = __float2half_rn(v);    // SN_DTYPE_F16: fp16 operands (RNE)
    else *reinterpret_cast<unsigned short*>(blob + e.dst) = f32_to_bf16_rne(v);
  }
}
}  // namespace

extern "C" {

int sn_abi_version(void) { return SN_ABI_VERSION; }

// layout introspection (used by the CPU layout tests; csrc/sn_layout.h is the single source of truth)
int sn_layout_xyz_slot_col(int h, int e) { return (h < 0 || h > 1 || e < 0 || e > 31) ? -2 : snl::xyz_slot_col(h, e); }
int sn_layout_dir_slot_col(int h, int e) { return (h < 0 || h > 1 || e < 0 || e > 15) ? -2 : snl::dir_slot_col(h, e); }
int sn_layout_slab_k(int slab) { return (slab < 0 || slab >= snl::N_SLABS) ? -2 : snl::slab_k(slab); }
int sn_layout_n_slabs(void) { return snl::N_SLABS; }

const char* sn_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case SN_E_BADARG: return "bad argument";
    case SN_E_TOOLARGE: return "problem too large for one launch";
    case SN_E_MISSING_RNG: return "perturb > 0 requires the perturb_rand tensor";
    case SN_E_UNSUPPORTED: return "unsupported configuration (dtype / samples per ray)";
    case SN_E_BADSHAPE: return "bad shape";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

long sn_packed_weights_bytes(int dtype) {
  if (dtype != SN_DTYPE_F32 && dtype != SN_DTYPE_BF16 && dtype != SN_DTYPE_BF16X3 && dtype != SN_DTYPE_F16) return SN_E_UNSUPPORTED;
  return snl::blob_bytes(dtype);
}
long sn_pack_table_entries(void) { return snl::table_entries(); }
long sn_pack_table_entries_dtype(int dtype) {
  if (dtype != SN_DTYPE_F32 && dtype != SN_DTYPE_BF16 && dtype != SN_DTYPE_BF16X3 && dtype != SN_DTYPE_F16) return SN_E_UNSUPPORTED;
  return snl::table_entries_dt(dtype);
}

int sn_build_pack_table(int dtype, int32_t* table_host) {
  if (dtype != SN_DTYPE_F32 && dtype != SN_DTYPE_BF16 && dtype != SN_DTYPE_BF16X3 && dtype != SN_DTYPE_F16) return SN_E_UNSUPPORTED;
  if (!table_host) return SN_E_BADARG;
  snl::build_pack_table(dtype, reinterpret_cast<snl::PackEntry*>(table_host));
  return 0;
}

long sn_packed_weights_bytes_bwd(void) { return snl::bblob_bytes(); }
long sn_pack_table_entries_bwd(void) { return snl::b_table_entries(); }
int sn_build_pack_table_bwd(int32_t* table_host) {
  if (!table_host) return SN_E_BADARG;
  snl::build_pack_table_bwd(reinterpret_cast<snl::PackEntry*>(table_host));
  return 0;
}

long sn_packed_weights_bytes_bwd_bf16(void) { return snl::bbblob_bytes(); }
long sn_pack_table_entries_bwd_bf16(void) { return snl::bb_table_entries(); }
int sn_build_pack_table_bwd_bf16(int32_t* table_host) {
  if (!table_host) return SN_E_BADARG;
  snl::build_pack_table_bwd_bf16(reinterpret_cast<snl::PackEntry*>(table_host));
  return 0;
}

long sn_packed_weights_bytes_bwd_bf16x3(void) { return snl::bbxblob_bytes(); }
long sn_pack_table_entries_bwd_bf16x3(void) { return snl::bbx_table_entries(); }
int sn_build_pack_table_bwd_bf16x3(int32_t* table_host) {
  if (!table_host) return SN_E_BADARG;
  snl::build_pack_table_bwd_bf16x3(reinterpret_cast<snl::PackEntry*>(table_host));
  return 0;
}

int sn_pack_weights(const float* const* raw, const int32_t* table, long n_entries, void* blob, int dtype, void* stream) {
  if (dtype != SN_DTYPE_F32 && dtype != SN_DTYPE_BF16 && dtype != SN_DTYPE_BF16X3 && dtype != SN_DTYPE_F16) return SN_E_UNSUPPORTED;
  if (!raw || !table || !blob || n_entries <= 0) return SN_E_BADARG;
  RawPtrs rp;
  for (int i = 0; i < snl::N_RAW; ++i) {
    if (!raw[i]) return SN_E_BADARG;
    rp.p[i] = raw[i];
  }
  hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, rp,
                     reinterpret_cast<const snl::PackEntry*>(table), n_entries, reinterpret_cast<char*>(blob), dtype);
  return (int)hipGetLastError();
}

int sn_sample_coarse(const float* rays, long n_rays, int n_samples, int use_disp, float perturb,
                     const float* perturb_rand, float* z_vals, void* stream) {
  if (!rays || !z_vals || n_rays < 0 || n_samples < 1) return SN_E_BADARG;
  return sn_sample_coarse_launch(rays, n_rays, n_samples, use_disp, perturb, perturb_rand, z_vals, (hipStream_t)stream);
}

// the two compilation passes of the MLP kernels (sn_device.h): SN_DTYPE_CLASSIC_HEADS in `dtype` selects the ReLU / Sigmoid
// pass; a sigma-only evaluation never reaches the heads and always runs the main pass
#define SN_HEADS(classic, name) ((classic) ? name##_classic_launch : name##_launch)

int sn_mlp_forward(const void* blob, int dtype, const float* rays, const float* z_vals, long n_rays, int n_samples,
                   int sigma_only, int flags, float* out, void* stream) {
  if (!blob || !rays || !z_vals || !out || n_rays < 0 || n_samples < 1) return SN_E_BADARG;
  const bool classic = (dtype & SN_DTYPE_CLASSIC_HEADS) && !sigma_only;
  dtype &= ~SN_DTYPE_CLASSIC_HEADS;
  if (dtype == SN_DTYPE_BF16X3)                  // fp32-level accuracy on the bf16 MFMA: 3-term split (csrc/sn_mlp_fwd_bf16x3.hip)
    return SN_HEADS(classic, sn_mlp_forward_bf16x3)(blob, rays, z_vals, n_rays * (long)n_samples, n_samples, sigma_only, 0, out,
                                                    nullptr, nullptr, 0, (hipStream_t)stream);
  if (dtype == SN_DTYPE_F16) {                   // fp16 operands on the bf16 kernels' instruction streams (round 6; inference only)
    if (!sigma_only && !(flags & SN_FLAG_BF16_COMPILER_SCHEDULED))
      return (classic ? sn_mlp_forward_bf16_v3_f16_classic_launch : sn_mlp_forward_bf16_v3_f16_launch)(
          blob, rays, z_vals, n_rays * (long)n_samples, n_samples, out, (hipStream_t)stream);
    return (classic ? sn_mlp_forward_bf16_f16_classic_launch : sn_mlp_forward_bf16_f16_launch)(
        blob, rays, z_vals, n_rays * (long)n_samples, n_samples, sigma_only, 0, out, nullptr, nullptr, 0, 0, (hipStream_t)stream);
  }
  if (dtype == SN_DTYPE_BF16 && !sigma_only && !(flags & SN_FLAG_BF16_COMPILER_SCHEDULED))     // the hand-scheduled kernel
    return SN_HEADS(classic, sn_mlp_forward_bf16_v3)(blob, rays, z_vals, n_rays * (long)n_samples, n_samples, out, (hipStream_t)stream);
  if (dtype == SN_DTYPE_BF16)
    return SN_HEADS(classic, sn_mlp_forward_bf16)(blob, rays, z_vals, n_rays * (long)n_samples, n_samples, sigma_only, 0, out, nullptr,
                                                  nullptr, 0, 0, (hipStream_t)stream);
  if (dtype != SN_DTYPE_F32) return SN_E_UNSUPPORTED;
  if (!(flags & SN_FLAG_F32_LDS_RING))           // round 6: fragments straight from L2, VALU-free trunk (csrc/sn_mlp_fwd_f32g.hip)
    return SN_HEADS(classic, sn_mlp_forward_f32g)(blob, rays, z_vals, n_rays * (long)n_samples, n_samples, sigma_only, 0, out,
                                                  nullptr, nullptr, 0, (hipStream_t)stream);
  return SN_HEADS(classic, sn_mlp_forward_f32)(blob, rays, z_vals, n_rays * (long)n_samples, n_samples, sigma_only, 0,
                                               out, nullptr, nullptr, 0, (hipStream_t)stream);
}

int sn_mlp_forward_train(const void* blob, int dtype, const float* rays, const float* z_vals, long n_rays, int n_samples,
                         float* out, float* acts, float* emb, long slot_rows, void* stream) {
  if (!blob || !rays || !z_vals || !out || !acts || !emb || n_rays < 0 || n_samples < 1) return SN_E_BADARG;
  const bool classic = dtype & SN_DTYPE_CLASSIC_HEADS;
  const bool compiler_scheduled = dtype & SN_DTYPE_COMPILER_SCHEDULED;
  const int emb16 = (dtype & SN_DTYPE_EMB_BF16) ? 1 : 0;
  dtype &= ~(SN_DTYPE_CLASSIC_HEADS | SN_DTYPE_COMPILER_SCHEDULED | SN_DTYPE_EMB_BF16);
  if (dtype != SN_DTYPE_F32 && dtype != SN_DTYPE_BF16 && dtype != SN_DTYPE_BF16_STATE && dtype != SN_DTYPE_BF16X3) return SN_E_UNSUPPORTED;
  const long n_points = n_rays * (long)n_samples;
  const long tile = (dtype == SN_DTYPE_F32 || dtype == SN_DTYPE_BF16X3) ? 128 : 256;      // whole point tiles are stored
  if (slot_rows < (n_points + tile - 1) / tile * tile) return SN_E_BADSHAPE;
  const bool hand = dtype == SN_DTYPE_BF16_STATE && !compiler_scheduled && n_points < (1l << 31) - 256;   // the hand-scheduled kernel
  if (emb16 && !hand) return SN_E_UNSUPPORTED;                               // (the only one that writes the bf16 form of emb)
  // fp32-level forward on the bf16 MFMA.  Its training state is the "x3 state" of sn_layout.h -- slots 0..8 hold (hi, lo) bf16 PAIRS in
  // the bytes of an fp32 row, slot 9 fp32 values + ReLU sign words -- NOT the array SN_DTYPE_F32 writes: only the SN_DTYPE_BF16X3 forms of
  // sn_mlp_backward_chain / sn_weight_grads read it (include/sinnerf_hip.h "pairing rule")
  if (dtype == SN_DTYPE_BF16X3) {
    if (slot_rows % 128 != 0) return SN_E_BADSHAPE;           // whole 128-point tiles, as the header states
    if (!compiler_scheduled && n_points < (1l << 31) - 256)  // the generated trunk (sn_mlp_fwd_bf16x3_t.hip): the same bits
      return SN_HEADS(classic, sn_mlp_forward_bf16x3_t)(blob, rays, z_vals, n_points, n_samples, out, acts, emb, slot_rows, (hipStream_t)stream);
    return SN_HEADS(classic, sn_mlp_forward_bf16x3)(blob, rays, z_vals, n_points, n_samples, 0, 0, out, acts, emb, slot_rows, (hipStream_t)stream);
  }
  if (hand)
    return SN_HEADS(classic, sn_mlp_forward_bf16_t)(blob, rays, z_vals, n_points, n_samples, out, acts, emb, slot_rows, emb16, (hipStream_t)stream);
  if (dtype != SN_DTYPE_F32)
    return SN_HEADS(classic, sn_mlp_forward_bf16)(blob, rays, z_vals, n_points, n_samples, 0, 0, out, acts, emb, slot_rows,
                                                  dtype == SN_DTYPE_BF16_STATE, (hipStream_t)stream);
  if (!compiler_scheduled)                       // round 6: the fragments-from-L2 kernel in store mode (csrc/sn_mlp_fwd_f32g.hip): the same state
    return SN_HEADS(classic, sn_mlp_forward_f32g_store)(blob, rays, z_vals, n_points, n_samples, 0, 0, out, acts, emb, slot_rows, (hipStream_t)stream);
  return SN_HEADS(classic, sn_mlp_forward_f32)(blob, rays, z_vals, n_points, n_samples, 0, 0, out, acts, emb, slot_rows,
                                               (hipStream_t)stream);
}

int sn_mlp_forward_train_embedded(const void* blob, int dtype, const float* x, long n_rows, int ld, float* out,
                                  float* acts, long slot_rows, void* stream) {
  if (!blob || !x || !out || !acts || n_rows < 0) return SN_E_BADARG;
  float* emb = acts;                             // not written for pre-embedded rows (the kernels only need it non-null)
  if (ld < 90) return SN_E_BADSHAPE;
  const bool classic = dtype & SN_DTYPE_CLASSIC_HEADS;
  dtype &= ~SN_DTYPE_CLASSIC_HEADS;
  if (dtype != SN_DTYPE_F32 && dtype != SN_DTYPE_BF16_STATE && dtype != SN_DTYPE_BF16X3) return SN_E_UNSUPPORTED;   // mixed precision keeps bf16 state
  const long tile = (dtype == SN_DTYPE_F32 || dtype == SN_DTYPE_BF16X3) ? 128 : 256;
  if (slot_rows < (n_rows + tile - 1) / tile * tile) return SN_E_BADSHAPE;
  if (dtype == SN_DTYPE_BF16X3)
    return SN_HEADS(classic, sn_mlp_forward_bf16x3)(blob, x, nullptr, n_rows, ld, 0, 1, out, acts, emb, slot_rows, (hipStream_t)stream);
  if (dtype != SN_DTYPE_F32)
    return SN_HEADS(classic, sn_mlp_forward_bf16)(blob, x, nullptr, n_rows, ld, 0, 1, out, acts, emb, slot_rows,
                                                  dtype == SN_DTYPE_BF16_STATE, (hipStream_t)stream);
  return SN_HEADS(classic, sn_mlp_forward_f32g_store)(blob, x, nullptr, n_rows, ld, 0, 1, out, acts, emb, slot_rows, (hipStream_t)stream);
}

int sn_mlp_backward_chain(const void* blob_bwd, int dtype, const float* acts, const float* out_raw, const float* g_raw,
                          long n_points, long slot_rows, float* g_acts, float* g_out, void* stream) {
  if (!blob_bwd || !acts || !out_raw || !g_raw || !g_acts || !g_out || n_points < 0) return SN_E_BADARG;
  const bool classic = dtype & SN_DTYPE_CLASSIC_HEADS;
  const bool compiler_scheduled = dtype & SN_DTYPE_COMPILER_SCHEDULED;
  dtype &= ~(SN_DTYPE_CLASSIC_HEADS | SN_DTYPE_COMPILER_SCHEDULED);
  if (dtype != SN_DTYPE_F32 && dtype != SN_DTYPE_BF16 && dtype != SN_DTYPE_BF16_STATE && dtype != SN_DTYPE_BF16X3) return SN_E_UNSUPPORTED;
  const long tile = (dtype == SN_DTYPE_F32 || dtype == SN_DTYPE_BF16X3) ? 128 : 256;      // whole point tiles are written
  if (slot_rows < (n_points + tile - 1) / tile * tile) return SN_E_BADSHAPE;
  // fp32-level accuracy on the bf16 MFMA (blob: *_bwd_bf16x3 table).  acts MUST be the x3 state sn_mlp_forward_train(SN_DTYPE_BF16X3)
  // wrote (masks from its sign words); g_acts leaves in the same layout ((hi, lo) pairs in slots 0..8) for sn_weight_grads(SN_DTYPE_BF16X3)
  if (dtype == SN_DTYPE_BF16X3) {
    if (slot_rows % 128 != 0) return SN_E_BADSHAPE;
    if (!compiler_scheduled && n_points < (1l << 31) - 256)  // the generated slab loop (sn_mlp_bwd_bf16x3_t.hip): the same bits
      return SN_HEADS(classic, sn_mlp_backward_chain_bf16x3_t)(blob_bwd, acts, out_raw, g_raw, n_points, slot_rows, g_acts, g_out, (hipStream_t)stream);
    return SN_HEADS(classic, sn_mlp_backward_chain_bf16x3)(blob_bwd, acts, out_raw, g_raw, n_points, slot_rows, g_acts, g_out, (hipStream_t)stream);
  }
  if (dtype == SN_DTYPE_BF16_STATE && !compiler_scheduled && n_points < (1l << 31) - 256)         // the hand-scheduled kernel
    return SN_HEADS(classic, sn_mlp_backward_chain_bf16_t)(blob_bwd, acts, out_raw, g_raw, n_points, slot_rows, g_acts, g_out, (hipStream_t)stream);
  if (dtype != SN_DTYPE_F32)
    return SN_HEADS(classic, sn_mlp_backward_chain_bf16)(blob_bwd, acts, out_raw, g_raw, n_points, slot_rows, g_acts, g_out,
                                                         dtype == SN_DTYPE_BF16_STATE, (hipStream_t)stream);
  if (!compiler_scheduled)                       // round 6: the fragments-from-L2 chain (csrc/sn_mlp_bwd_f32g.hip): the same bits
    return SN_HEADS(classic, sn_mlp_backward_chain_f32g)(blob_bwd, acts, out_raw, g_raw, n_points, slot_rows, g_acts, g_out, (hipStream_t)stream);
  return SN_HEADS(classic, sn_mlp_backward_chain_f32)(blob_bwd, acts, out_raw, g_raw, n_points, slot_rows, g_acts, g_out,
                                                      (hipStream_t)stream);
}

int sn_generate_rays(const float* c2w, int H, int W, float focal, float near, float far, int x0, int y0, int stride_x,
                     int stride_y, int patch_w, int patch_h, float* rays, void* stream) {
  if (!c2w || !rays || H < 1 || W < 1 || stride_x < 1 || stride_y < 1 || patch_w < 0 || patch_h < 0) return SN_E_BADARG;
  if (x0 < 0 || y0 < 0 || (patch_w > 0 && x0 + (patch_w - 1) * stride_x >= W) || (patch_h > 0 && y0 + (patch_h - 1) * stride_y >= H))
    return SN_E_BADSHAPE;
  return sn_generate_rays_launch(c2w, H, W, focal, near, far, x0, y0, stride_x, stride_y, patch_w, patch_h, rays,
                                 (hipStream_t)stream);
}

int sn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return SN_E_BADARG;
  return sn_adam_step_launch(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step,
                             (hipStream_t)stream);
}

long sn_render_loss_workspace_bytes(void) { return sn_render_loss_workspace_bytes_impl(); }

int sn_render_loss(const float* rgb_coarse, const float* rgb_fine, const float* depth_coarse, const float* depth_fine,
                   const float* rgb_gt, const float* depth_gt, const unsigned char* mask, int mask_mode, long n,
                   float w_rgb, float w_depth, float* g_rgb_coarse, float* g_rgb_fine, float* g_depth_coarse,
                   float* g_depth_fine, void* workspace, float* out, void* stream) {
  if (!workspace || !out || n < 1 || mask_mode < 0 || mask_mode > 2) return SN_E_BADARG;
  if ((mask_mode == 2 && !mask) || (mask_mode != 0 && !depth_gt)) return SN_E_BADARG;
  if ((rgb_coarse || rgb_fine) && !rgb_gt) return SN_E_BADARG;
  if ((depth_coarse || depth_fine) && !depth_gt) return SN_E_BADARG;
  return sn_render_loss_launch(rgb_coarse, rgb_fine, depth_coarse, depth_fine, rgb_gt, depth_gt, mask, mask_mode, n,
                               w_rgb, w_depth, g_rgb_coarse, g_rgb_fine, g_depth_coarse, g_depth_fine, workspace, out,
                               (hipStream_t)stream);
}

int sn_dw_gemm(const void* tasks, int n_tasks, void* stream) {
  if (!tasks || n_tasks < 0) return SN_E_BADARG;
  return sn_dw_launch(tasks, n_tasks, (hipStream_t)stream);
}

long sn_weight_grads_workspace_bytes(long slot_rows, int dtype) {
  if (slot_rows < 16 || slot_rows % 16 != 0) return SN_E_BADSHAPE;
  const int emb16 = (dtype & SN_DTYPE_EMB_BF16) ? 1 : 0;
  dtype &= ~SN_DTYPE_EMB_BF16;
  if (dtype != SN_DTYPE_F32 && dtype != SN_DTYPE_BF16 && dtype != SN_DTYPE_BF16_STATE && dtype != SN_DTYPE_BF16X3) return SN_E_UNSUPPORTED;
  if (emb16 && dtype != SN_DTYPE_BF16_STATE) return SN_E_UNSUPPORTED;
  return sn_weight_grads_workspace_bytes_impl(slot_rows, dtype, emb16);
}

int sn_weight_grads(const void* acts, const float* emb, const void* g_acts, long slot_rows, int dtype, void* workspace,
                    float* const* grads, int accumulate, void* stream) {
  if (!acts || !emb || !g_acts || !workspace || !grads) return SN_E_BADARG;
  if (slot_rows < 16 || slot_rows % 16 != 0) return SN_E_BADSHAPE;
  const int emb16 = (dtype & SN_DTYPE_EMB_BF16) ? 1 : 0;
  dtype &= ~SN_DTYPE_EMB_BF16;
  if (dtype != SN_DTYPE_F32 && dtype != SN_DTYPE_BF16 && dtype != SN_DTYPE_BF16_STATE && dtype != SN_DTYPE_BF16X3) return SN_E_UNSUPPORTED;
  if (emb16 && dtype != SN_DTYPE_BF16_STATE) return SN_E_UNSUPPORTED;
  return sn_weight_grads_launch(acts, emb, g_acts, slot_rows, dtype, emb16, workspace, grads, accumulate ? 1 : 0, (hipStream_t)stream);
}

int sn_composite_backward(const float* raw, const float* z_vals, const float* rays, const float* noise, float noise_std,
                          long n_rays, int n_samples, int white_back, const float* g_rgb, const float* g_depth,
                          const float* g_weights, float* g_raw, void* stream) {
  if (!raw || !z_vals || !rays || !g_raw || n_rays < 0 || n_samples < 1) return SN_E_BADARG;
  return sn_composite_backward_launch(raw, z_vals, rays, noise, noise_std, n_rays, n_samples, white_back, g_rgb, g_depth,
                                      g_weights, g_raw, (hipStream_t)stream);
}

int sn_mlp_forward_embedded(const void* blob, int dtype, const float* x, long n_rows, int ld, int sigma_only,
                            int flags, float* out, void* stream) {
  if (!blob || !x || !out || n_rows < 0) return SN_E_BADARG;
  if (ld < (sigma_only ? 63 : 90)) return SN_E_BADSHAPE;
  const bool classic = (dtype & SN_DTYPE_CLASSIC_HEADS) && !sigma_only;
  dtype &= ~SN_DTYPE_CLASSIC_HEADS;
  if (dtype == SN_DTYPE_BF16X3)
    return SN_HEADS(classic, sn_mlp_forward_bf16x3)(blob, x, nullptr, n_rows, ld, sigma_only, 1, out, nullptr, nullptr, 0, (hipStream_t)stream);
  if (dtype == SN_DTYPE_F16)
    return (classic ? sn_mlp_forward_bf16_f16_classic_launch : sn_mlp_forward_bf16_f16_launch)(
        blob, x, nullptr, n_rows, ld, sigma_only, 1, out, nullptr, nullptr, 0, 0, (hipStream_t)stream);
  if (dtype == SN_DTYPE_BF16)
    return SN_HEADS(classic, sn_mlp_forward_bf16)(blob, x, nullptr, n_rows, ld, sigma_only, 1, out, nullptr, nullptr, 0, 0, (hipStream_t)stream);
  if (dtype != SN_DTYPE_F32) return SN_E_UNSUPPORTED;
  if (!(flags & SN_FLAG_F32_LDS_RING))
    return SN_HEADS(classic, sn_mlp_forward_f32g)(blob, x, nullptr, n_rows, ld, sigma_only, 1, out, nullptr, nullptr, 0, (hipStream_t)stream);
  return SN_HEADS(classic, sn_mlp_forward_f32)(blob, x, nullptr, n_rows, ld, sigma_only, 1,
                                               out, nullptr, nullptr, 0, (hipStream_t)stream);
}

int sn_composite_forward(const float* raw, int has_rgb, const float* z_vals, const float* rays, const float* noise,
                         float noise_std, long n_rays, int n_samples, int white_back, float* rgb, float* depth,
                         float* weights, void* stream) {
  if (!raw || !z_vals || !rays || !weights || n_rays < 0 || n_samples < 1) return SN_E_BADARG;
  if (has_rgb && (!rgb || !depth)) return SN_E_BADARG;
  return sn_composite_forward_launch(raw, has_rgb, z_vals, rays, noise, noise_std, n_rays, n_samples, white_back, rgb,
                                     depth, weights, (hipStream_t)stream);
}

int sn_sample_pdf(const float* z_vals, const float* weights, const float* u, long n_rays, int n_samples,
                  int n_importance, float* z_fine, float* z_merged, void* stream) {
  if (!z_vals || !weights || !z_merged || n_rays < 0) return SN_E_BADARG;
  return sn_sample_pdf_launch(z_vals, weights, u, n_rays, n_samples, n_importance, z_fine, z_merged, (hipStream_t)stream);
}

int sn_sample_pdf_bins(const float* bins, const float* weights, const float* u, long n_rays, int n_bins,
                       int n_importance, float eps, float* samples, void* stream) {
  if (!bins || !weights || !samples || n_rays < 0 || !(eps > 0.0f)) return SN_E_BADARG;
  return sn_sample_pdf_bins_launch(bins, weights, u, n_rays, n_bins, n_importance, eps, samples, (hipStream_t)stream);
}

}  // extern "C"
