/* sinnerf_hip.h -- C ABI of libsinnerf_hip.so: the MI355X (gfx950) volume-rendering hot path of SinNeRF.
 *
 * The reference (VITA-Group/SinNeRF) has no FFI: its hot path is Python calling ATen ops
 * (models/rendering.py::render_rays, models/nerf.py::{Embedding,NeRF}).  Each entry point below replaces the
 * group of reference lines cited next to it; sinnerf_amd/rendering.py (the drop-in render_rays) binds them
 * with ctypes -- see INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions: every pointer is a DEVICE pointer unless the name ends in _host; tensors are dense row-major
 * fp32; `stream` is a hipStream_t passed as void* (NULL = default stream).  Functions only enqueue work: no
 * allocation, no synchronisation, no global state.  Return 0 on success, a negative SN_E_* code for argument
 * errors, a positive value = hipError_t of a failed launch.
 */
#ifndef SINNERF_HIP_H
#define SINNERF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SN_ABI_VERSION 5   /* 2: sn_sample_pdf_bins gained eps, dtype carries SN_DTYPE_CLASSIC_HEADS, sn_mlp_forward flag bits
                            * 3: dtype carries SN_DTYPE_COMPILER_SCHEDULED and SN_DTYPE_EMB_BF16
                            * 4: SN_DTYPE_BF16X3 (inference and training entries, packers), sn_pack_table_entries_dtype
                            * 5: SN_DTYPE_F16 (inference entries, packer), SN_FLAG_F32_LDS_RING */

#define SN_DTYPE_F32 0  /* v_mfma_f32_32x32x2_f32, exact fp32                                          */
#define SN_DTYPE_BF16 1 /* v_mfma_f32_32x32x16_bf16, bf16 operands / fp32 accumulate                    */
#define SN_DTYPE_BF16_STATE 2 /* training entries only: SN_DTYPE_BF16 arithmetic AND acts / g_acts stored as bf16
                               * (same shapes; the float* parameters then point at bf16 arrays; emb stays fp32
                               * unless SN_DTYPE_EMB_BF16 is OR-ed in)                                             */
#define SN_DTYPE_BF16X3 3 /* sn_mlp_forward, sn_mlp_forward_embedded, sn_mlp_forward_train[_embedded], sn_mlp_backward_chain,
                           * sn_weight_grads (training state: see sn_mlp_forward_train) and the packer: fp32-LEVEL accuracy on the bf16
                           * MFMA -- every weight and activation as a (hi, lo) bf16 pair, W.x ~= Wh.xh + Wl.xh + Wh.xl with fp32
                           * accumulation (3 bf16 MFMAs instead of 8 fp32 ones per 16 k; SURVEY §7 "3-term bf16 split"); exact
                           * embeddings and fp32 heads as SN_DTYPE_F32.  Measured against the fp32 bars of the parity tests.  */
/* OR-ed into `dtype` of the sn_mlp_* entry points: the network was built as NeRF(use_new_activation=False), the
 * constructor's default (models/nerf.py:47-50, :91-100) -- ReLU after dir_encoding, Sigmoid after rgb -- instead of the
 * ShiftedSoftplus / WidenedSigmoid heads both reference call sites ask for (models/nerf.py:81-90, models/activations.py).
 * Blobs, training state and every other entry point are the same for both.                                          */
#define SN_DTYPE_F16 4 /* ABI 5. sn_mlp_forward, sn_mlp_forward_embedded, sn_pack_weights only (INFERENCE): v_mfma_f32_32x32x16_f16,
                        * fp16 operands (11 significand bits instead of bf16's 8) / fp32 accumulate at the bf16 matrix rate -- the
                        * instruction streams of SN_DTYPE_BF16 with v_cvt_pk_f16_f32.  On the trained-weight fixtures its error is
                        * 8-17x below SN_DTYPE_BF16's (tests/test_trained_weights_gpu.py); activations beyond +-65504 overflow (the
                        * entry points do not check).  Blob layout and size = SN_DTYPE_BF16's. */
#define SN_DTYPE_CLASSIC_HEADS 0x100
/* OR-ed into `dtype` of sn_mlp_forward_train / sn_mlp_backward_chain: run the PREVIOUS generation of the kernel instead of the shipped one --
 * SN_DTYPE_BF16_STATE / SN_DTYPE_BF16X3: the compiler-scheduled kernels (csrc/sn_mlp_fwd_bf16.hip, csrc/sn_mlp_bwd_bf16.hip,
 * csrc/sn_mlp_{fwd,bwd}_bf16x3.hip) instead of the generated instruction streams (csrc/sn_mlp_*_t.hip); SN_DTYPE_F32 (round 6): the
 * LDS-ring kernels (csrc/sn_mlp_fwd.hip, csrc/sn_mlp_bwd.hip) instead of the fragments-from-L2 ones (csrc/sn_mlp_fwd_f32g.hip,
 * csrc/sn_mlp_bwd_f32g.hip).  Same arithmetic, same stored state bit for bit; kept for A/B timing and the bit-identity tests. */
#define SN_DTYPE_COMPILER_SCHEDULED 0x200
/* OR-ed into `dtype` (SN_DTYPE_BF16_STATE, hand-scheduled kernels) of sn_mlp_forward_train, sn_weight_grads and
 * sn_weight_grads_workspace_bytes -- all three or none: `emb` is a bf16 array (slot_rows, 128) holding the embedded inputs as the
 * MFMA operands the forward builds, in its K-slot order instead of the reference's column order: positions [0, 64) =
 * Embedding(xyz) (lane half h, slot e at 32 h + e), [64, 96) = Embedding(dir) (16 h + e), [96, 128) never written.  192 B per point
 * instead of 360 B written, half the bytes read by the three contractions that use it; sn_weight_grads returns the same gradients bit
 * for bit (it contracts the same bf16 values in the same order and permutes the 63 / 27 columns back).              */
#define SN_DTYPE_EMB_BF16 0x400

#define SN_E_BADARG (-1)
#define SN_E_TOOLARGE (-2)
#define SN_E_MISSING_RNG (-3)
#define SN_E_UNSUPPORTED (-4)
#define SN_E_BADSHAPE (-5)

/* sn_mlp_forward `flags` (sn_mlp_forward_embedded: reserved, pass 0):
 * bit 1: dtype SN_DTYPE_BF16 only -- run the compiler-scheduled kernel (csrc/sn_mlp_fwd_bf16.hip) instead of the
 * hand-scheduled one (csrc/sn_mlp_fwd_bf16_v3.hip); same arithmetic, kept for A/B timing and as the sigma-only / training form. */
#define SN_FLAG_BF16_COMPILER_SCHEDULED 2
/* bit 2: dtype SN_DTYPE_F32 only -- run the round-1..5 inference kernel (csrc/sn_mlp_fwd.hip: weights through an LDS ring filled by
 * LDS-DMA, one barrier per slab) instead of the round-6 one (csrc/sn_mlp_fwd_f32g.hip: A fragments straight from L2 into a register
 * ring, no VALU instruction in the trunk, no barrier); bit-identical results, kept for A/B timing.  Also honoured by
 * sn_mlp_forward_embedded. */
#define SN_FLAG_F32_LDS_RING 4

int sn_abi_version(void);
const char* sn_error_string(int code);

/* layout introspection (host only; mirrors csrc/sn_layout.h -- used by the CPU layout tests) */
int sn_layout_xyz_slot_col(int lane_half, int slot); /* reference Embedding(3,10) column, -1 = zero pad */
int sn_layout_dir_slot_col(int lane_half, int slot); /* reference Embedding(3,4) column, -1 = zero pad  */
int sn_layout_slab_k(int slab);
int sn_layout_n_slabs(void);

/* ---- weights -------------------------------------------------------------------------------------------
 * One NeRF MLP (models/nerf.py:66-103, NeRF(D=8,W=256,63,27,skips=[4],use_new_activation=True)) is consumed by
 * the kernels as a "packed blob": weights reordered into MFMA A-fragment order (csrc/sn_layout.h).
 * raw[24] = device pointers of the state_dict tensors in this order (names = nerf.py:75-103):
 *   xyz_encoding_{1..8}.0.weight/.bias (16), xyz_encoding_final.weight/.bias, dir_encoding.0.weight/.bias,
 *   sigma.weight/.bias, rgb.0.weight/.bias                                                              */
#define SN_N_RAW_TENSORS 24
long sn_packed_weights_bytes(int dtype);
long sn_pack_table_entries(void);           /* SN_DTYPE_F32 / SN_DTYPE_BF16 */
long sn_pack_table_entries_dtype(int dtype); /* any packable dtype (SN_DTYPE_BF16X3 has two entries per weight) */
/* fills table_host[2*entries] int32 (dst byte offset, src tensor<<20|offset or -1); upload it once */
int sn_build_pack_table(int dtype, int32_t* table_host);
int sn_pack_weights(const float* const* raw_host_array_of_device_ptrs, const int32_t* table, long n_entries,
                    void* blob, int dtype, void* stream);
/* transposed-weight blob consumed by sn_mlp_backward_chain (fp32; same table format, same packer) */
long sn_packed_weights_bytes_bwd(void);
long sn_pack_table_entries_bwd(void);
int sn_build_pack_table_bwd(int32_t* table_host);
/* ... and its bf16-operand form (sn_mlp_backward_chain with dtype SN_DTYPE_BF16; pack with sn_pack_weights(..., SN_DTYPE_BF16)) */
long sn_packed_weights_bytes_bwd_bf16(void);
long sn_pack_table_entries_bwd_bf16(void);
int sn_build_pack_table_bwd_bf16(int32_t* table_host);
/* ... and its bf16x3 form (sn_mlp_backward_chain with dtype SN_DTYPE_BF16X3; pack with sn_pack_weights(..., SN_DTYPE_BF16X3)) */
long sn_packed_weights_bytes_bwd_bf16x3(void);
long sn_pack_table_entries_bwd_bf16x3(void);
int sn_build_pack_table_bwd_bf16x3(int32_t* table_host);

/* ---- models/rendering.py:264-282  z_vals = near*(1-t)+far*t (or disparity), stratified perturb -----------
 * rays (n_rays,8) = [o(3), d(3), near, far] (rendering.py:257-258); perturb_rand (n_rays,n_samples) = the
 * torch.rand draw of :281 (required iff perturb > 0); z_vals out (n_rays,n_samples).                      */
int sn_sample_coarse(const float* rays, long n_rays, int n_samples, int use_disp, float perturb,
                     const float* perturb_rand, float* z_vals, void* stream);

/* ---- models/rendering.py:187-212 (closure `inference`, MLP part) + models/nerf.py:36-41,122-148 -----------
 * For every sample point p = (ray, i): xyz = o + d*z (rendering.py:284-285), Embedding(xyz) (63), Embedding(d)
 * (27), NeRF.forward.  out: (n_rays*n_samples, 4) = [rgb, raw sigma] (nerf.py:146), or (n_rays*n_samples)
 * raw sigma when sigma_only (nerf.py:136-138).  Replaces the chunk loop of rendering.py:196-206.           */
int sn_mlp_forward(const void* blob, int dtype, const float* rays, const float* z_vals, long n_rays, int n_samples,
                   int sigma_only, int flags, float* out, void* stream);

/* ---- training forward: sn_mlp_forward + the activations autograd would keep alive --------------
 * dtype SN_DTYPE_BF16 = mixed precision: bf16-operand contractions as in sn_mlp_forward, fp32 activations stored (the
 * values before their bf16 rounding) for the fp32 backward.
 * slot_rows = rows allocated per slot, >= n_points rounded up to a multiple of 128 (fp32) / 256 (bf16): the kernel stores
 * whole point tiles without a predicate (rows >= n_points receive finite copies of the last point; their gradients are zero, so
 * sn_dw_gemm, which walks whole 16-point chunks, ignores them).
 * acts (10, slot_rows, 256): slots 0..7 = outputs of xyz_encoding_1..8 (post-ReLU), 8 = xyz_encoding_final,
 * 9 = dir_encoding output (128 wide, leading dimension 256).  emb (slot_rows, 128): the kernel writes columns
 * [0,63) = Embedding(xyz) and [64,91) = Embedding(dir) in the reference's column order (nerf.py:36-41) of every row
 * of the whole point tiles; columns 63 and 91..127 are never written (nor read back by sn_weight_grads' results:
 * zero-fill them only if you contract emb yourself).
 * SN_DTYPE_BF16_STATE: the unused half of slot 9 (columns 128..255 of the bf16 array, 256 B per point) receives the ReLU
 * SIGN WORDS of xyz_encoding_1..8 -- for the 64 consecutive points a wave owns, the 32-bit word of (layer l, 32-feature
 * tile t) and lane L sits in row (first point + 8 l + t), bytes [256 + 4 L, 256 + 4 L + 4).  sn_mlp_backward_chain with
 * SN_DTYPE_BF16_STATE takes its ReLU masks from these words (and reads no other activation but slot 9's values): pass it
 * the acts array THIS entry wrote.
 * SN_DTYPE_BF16X3: buffers of the fp32 shapes and sizes (slot_rows a multiple of 128).  emb and slot 9 are fp32 exactly as above;
 * SLOTS 0..8 hold every value as the (hi, lo) bf16 PAIR the kernels compute with -- hi = RNE(x), lo = RNE(x - hi), x = hi + lo to
 * 2^-17 relative -- in the 1 KB an fp32 row takes: per 8 consecutive features 16 B of hi parts, then 16 B of lo parts (feature f of a
 * row: hi at byte 32 (f / 8) + 2 (f % 8), lo 16 B further).  sn_mlp_backward_chain writes slots 0..8 of g_acts the same way,
 * sn_weight_grads reads both (no conversion on its way to the MFMA).  The unused half of slot 9 receives ReLU sign words for the
 * chain as with SN_DTYPE_BF16_STATE, in this kernel's tile shape: for the 32 points a wave owns, layer l sits in rows
 * first point + 4 l + (lane >> 4), bytes [512 + 16 (lane & 15), + 16) -- one word per lane and pair of 32-feature tiles.      */
int sn_mlp_forward_train(const void* blob, int dtype, const float* rays, const float* z_vals, long n_rays, int n_samples,
                         float* out, float* acts, float* emb, long slot_rows, void* stream);

/* ---- the same for NeRF.forward(x) under autograd (models/nerf.py:105-148 is an ordinary differentiable nn.Module): x
 * (n_rows, ld >= 90) pre-embedded rows as for sn_mlp_forward_embedded; acts / slot_rows as above.  The `emb` matrix of the
 * weight-gradient contractions is a column re-layout of x the caller builds itself ([0,63) = x[:, 0:63], [64,91) =
 * x[:, 63:90], zeros elsewhere).  dtype: SN_DTYPE_F32 or SN_DTYPE_BF16_STATE.                                   */
int sn_mlp_forward_train_embedded(const void* blob, int dtype, const float* x, long n_rows, int ld, float* out,
                                  float* acts, long slot_rows, void* stream);

/* ---- backward of models/nerf.py:122-148 w.r.t. layer outputs (what loss.backward() at sinnerf.py:551 runs) ----
 * blob_bwd: transposed-weight blob (sn_build_pack_table_bwd).  out_raw / g_raw (n_points,4): forward output and its
 * gradient.  Writes g_acts (10, slot_rows, 256): slots 0..7 = dL/d(pre-activation) of xyz_encoding_1..8, 8 = of
 * xyz_encoding_final, 9 = of dir_encoding (128 wide); g_out (n_points,4) = dL/d(pre-activation) of rgb (3), sigma (1).
 * dtype SN_DTYPE_BF16: blob_bwd from the *_bwd_bf16 table, bf16-operand contractions, everything stored stays fp32.
 * dtype SN_DTYPE_BF16X3: blob_bwd from the *_bwd_bf16x3 table; fp32-level accuracy on the bf16 MFMA (3-term hi/lo split); acts as
 * sn_mlp_forward_train(SN_DTYPE_BF16X3) wrote them (masks from the sign words, h2 from slot 9), g_acts in the same layout: slots 0..8
 * as (hi, lo) pairs, slot 9 (128 columns + the head block) fp32 (slot_rows a multiple of 128).
 * dtype SN_DTYPE_BF16_STATE: acts / g_acts are bf16 arrays; the ReLU masks come from the sign words sn_mlp_forward_train
 * left in slot 9 (see there), gradients leave as whole 128-byte rows.
 * slot_rows as for sn_mlp_forward_train (>= n_points rounded up to 128, 256 for bf16; rows >= n_points of slots' 256
 * columns are written as zeros, the caller zero-fills the rest of the pad rows).  Weight gradients are the contractions  dW_l = g_l^T X_l  over points of these matrices with acts / emb.
 * PAIRING RULE (all three training entry points): the state arrays of the arithmetics have the same shapes and byte sizes but NOT the
 * same contents -- `acts` written by sn_mlp_forward_train(dtype D) may only be read by sn_mlp_backward_chain(D) and sn_weight_grads(D),
 * `g_acts` written by sn_mlp_backward_chain(D) only by sn_weight_grads(D), with D one of {SN_DTYPE_F32 (also read by SN_DTYPE_BF16),
 * SN_DTYPE_BF16_STATE, SN_DTYPE_BF16X3}.  The library cannot tell the layouts apart from the pointers: a mismatch (e.g. an x3 state
 * handed to the SN_DTYPE_F32 chain) yields wrong gradients, not an error code.  SN_DTYPE_BF16X3 additionally requires slot_rows % 128 == 0
 * (checked: SN_E_BADSHAPE).                                                                                                            */
int sn_mlp_backward_chain(const void* blob_bwd, int dtype, const float* acts, const float* out_raw, const float* g_raw,
                          long n_points, long slot_rows, float* g_acts, float* g_out, void* stream);

/* ---- weight gradients  dW[m,n] = sum_p G[p,m] X[p,n],  db[m] = sum_p G[p,m]  (autograd of every nn.Linear of
 * models/nerf.py:66-103: gW = g_y^T x, gb = sum g_y) as K-split MFMA contractions over all sample points.
 * tasks: DEVICE array of n_tasks 64-byte records (one workgroup each):
 *   { const float* a;  const float* b;  float* c;  float* bias_or_NULL;  int64 k0, k1;  int32 lda, ldb;  int32 ldc, variant; }
 * a = G + column offset (row-major [P][lda]), b = X + column offset ([P][ldb]), point range [k0,k1) with
 * (k1-k0) % 16 == 0 and every row readable (zero-padded G rows contribute nothing);
 * variant 0: M x N = 256x256, 1: 256x64, 2: 128x256, 3: 128x64.  c receives the PARTIAL M x N result of that K-range
 * (row-major, leading dimension ldc), bias the partial column sums of a; the caller sums the partials of a problem. */
int sn_dw_gemm(const void* tasks, int n_tasks, void* stream);

/* ---- ALL weight / bias gradients of one NeRF from the stored training state, as two launches with nothing built on the
 * host (capturable in a HIP graph): the 14 contractions above (K-split over ~one workgroup per CU by a per-mode cost model,
 * described to the kernel BY VALUE in its arguments), then one kernel that sums the K-split partials in a fixed order
 * (deterministic, no atomics) and writes the results in the parameters' own shapes -- what autograd hands to
 * nn.Linear.weight.grad / .bias.grad for models/nerf.py:66-103 (the cat of the skip / direction inputs, nerf.py:133,142,
 * becomes two column ranges of the same gradient).
 *   acts, emb: as written by sn_mlp_forward_train; g_acts: as written by sn_mlp_backward_chain (pad rows zero);
 *   slot_rows: a multiple of 16; dtype: SN_DTYPE_F32 (fp32 MFMAs), SN_DTYPE_BF16 (bf16 operands, fp32 state) or
 *   SN_DTYPE_BF16_STATE (acts / g_acts stored as bf16) or SN_DTYPE_BF16X3 (the x3 state; see the PAIRING RULE at sn_mlp_backward_chain:
 *   acts / g_acts must have been written with the SAME dtype);
 *   workspace: sn_weight_grads_workspace_bytes(slot_rows, dtype) bytes of DEVICE scratch (the K-split partials);
 *   grads: HOST array of SN_N_RAW_TENSORS device pointers in the order of sn_pack_weights' `raw` (NULL = not wanted);
 *   accumulate != 0: grads[i] += result (autograd's accumulation into an existing .grad), else grads[i] = result.        */
long sn_weight_grads_workspace_bytes(long slot_rows, int dtype);
int sn_weight_grads(const void* acts, const float* emb, const void* g_acts, long slot_rows, int dtype, void* workspace,
                    float* const* grads_host_array_of_device_ptrs, int accumulate, void* stream);

/* ---- backward of models/rendering.py:215-246 w.r.t. raw=[rgb,sigma]; g_rgb (n_rays,3), g_depth (n_rays),
 * g_weights (n_rays,n_samples) are upstream gradients (any may be NULL = zeros); g_raw (n_rays,n_samples,4).  */
int sn_composite_backward(const float* raw, const float* z_vals, const float* rays, const float* noise, float noise_std,
                          long n_rays, int n_samples, int white_back, const float* g_rgb, const float* g_depth,
                          const float* g_weights, float* g_raw, void* stream);

/* ---- models/nerf.py:105-148  NeRF.forward(x, sigma_only) on an already embedded matrix --------------------
 * x (n_rows, ld) with columns [0,63) = embedded xyz and [63,90) = embedded dir (ignored when sigma_only).   */
int sn_mlp_forward_embedded(const void* blob, int dtype, const float* x, long n_rows, int ld, int sigma_only,
                            int flags, float* out, void* stream);

/* ---- models/rendering.py:215-246  deltas, noise, alpha, cumprod transmittance, weights, rgb/depth/white_back
 * raw: (n_rays,n_samples,4) if has_rgb else (n_rays,n_samples) raw sigma (weights_only, rendering.py:238-239)
 * noise: (n_rays,n_samples) torch.randn draw of :224 or NULL (= zeros).  rgb (n_rays,3), depth (n_rays) are
 * written only when has_rgb; weights (n_rays,n_samples) always.                                           */
int sn_composite_forward(const float* raw, int has_rgb, const float* z_vals, const float* rays, const float* noise,
                         float noise_std, long n_rays, int n_samples, int white_back, float* rgb, float* depth,
                         float* weights, void* stream);

/* ---- models/rendering.py:15-61 (sample_pdf) + :310-315 (mid points, detach, cat + sort) -------------------
 * z_vals/weights (n_rays,n_samples) coarse depths / compositing weights; u (n_rays,n_importance) = torch.rand
 * draw of :43 or NULL for det=True (linspace, :40).  z_fine (n_rays,n_importance) optional (may be NULL),
 * z_merged (n_rays, n_samples+n_importance) ascending.                                                     */
int sn_sample_pdf(const float* z_vals, const float* weights, const float* u, long n_rays, int n_samples,
                  int n_importance, float* z_fine, float* z_merged, void* stream);

/* ---- models/rendering.py:15-61  sample_pdf(bins, weights, N_importance, det, eps) exactly as the reference exposes
 * it: bins (n_rays, n_bins+1), weights (n_rays, n_bins) -> samples (n_rays, n_importance); u as above; eps > 0 (the
 * reference default 1e-5 is what render_rays uses, rendering.py:311).                                          */
int sn_sample_pdf_bins(const float* bins, const float* weights, const float* u, long n_rays, int n_bins,
                       int n_importance, float eps, float* samples, void* stream);

/* ==== "next" rows: the steps immediately before / after the hot path ========================== */

/* ---- datasets/ray_utils.py:86-133 (get_ray_directions + get_rays) + the [o, d, near, far] packing of the datasets
 * (e.g. blender_ray_patch_1image_rot3d.py:201-211; strided patches :487-498): rays generated on the GPU.
 * c2w: 12 floats (3x4 row-major, DEVICE).  Pixel (x0 + ix*stride_x, y0 + iy*stride_y), ix < patch_w, iy < patch_h,
 * written row-major over (iy, ix) into rays (patch_w*patch_h, 8).  Full frame = (0, 0, 1, 1, W, H).             */
int sn_generate_rays(const float* c2w, int H, int W, float focal, float near, float far, int x0, int y0, int stride_x,
                     int stride_y, int patch_w, int patch_h, float* rays, void* stream);

/* ---- utils/__init__.py:19-21 (torch.optim.Adam, eps=1e-8) over one flat fp32 buffer (the all-reduce buffer).
 * step = 1-based iteration count (bias correction).                                                             */
int sn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, void* stream);

/* ---- losses on the rendered rays, forward + gradients in two launches: MSELoss (losses.py:12-22: nn.MSELoss 'mean' of
 * rgb_coarse and rgb_fine against the targets), SL1Loss (models/sinnerf.py:32-42: nn.SmoothL1Loss 'mean', beta 1, of
 * depth_coarse and depth_fine -- call sites :310-319) and psnr (metrics.py:5-15).
 *   rgb_* (n,3), depth_* (n), rgb_gt (n,3), depth_gt (n): DEVICE fp32; any prediction and either target may be NULL
 *   (its terms are then 0 and its gradient is not written).
 *   mask_mode 0: every depth element counts (useMask=False); 1: depth_gt > 0 (mask=None, useMask=True); 2: mask (n) uint8.
 *   out[8] = mse_coarse, mse_fine, sl1_coarse, sl1_fine, total = w_rgb (mse_c + mse_f) + w_depth (sl1_c + sl1_f),
 *            psnr_coarse, psnr_fine, number of depth elements counted.
 *   g_*: gradients of `total` w.r.t. the matching prediction (NULL = not wanted).
 *   workspace: sn_render_loss_workspace_bytes() bytes of DEVICE scratch (per-block partial sums; deterministic order). */
long sn_render_loss_workspace_bytes(void);
int sn_render_loss(const float* rgb_coarse, const float* rgb_fine, const float* depth_coarse, const float* depth_fine,
                   const float* rgb_gt, const float* depth_gt, const unsigned char* mask, int mask_mode, long n,
                   float w_rgb, float w_depth, float* g_rgb_coarse, float* g_rgb_fine, float* g_depth_coarse,
                   float* g_depth_fine, void* workspace, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
