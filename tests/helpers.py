"""Shared helpers for the parity tests (tolerance policy of SURVEY.md §8d / BASELINE north_star)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp32 bar (north_star: "within 1e-3 rel on fp32"):  |new-ref| <= 1e-3*|ref| + 1e-5 on rgb_*/depth_*
# (the absolute floor only matters for outputs that are ~0, e.g. depth of an empty ray, where a
# relative error is meaningless).  opacity_* (per-sample compositing weights, never consumed by the
# reference's train/eval code) : abs 1e-4 on >= 99.9 % of the entries and never above 2e-2 -- a fine
# sample that sits on a cdf knot may legitimately land in the neighbouring bin (see well_conditioned()).
REL_TOL = 1e-3
ABS_FLOOR = 1e-5
OPACITY_ABS_TOL = 1e-4
OPACITY_OUTLIER_FRAC = 1e-3
OPACITY_OUTLIER_MAX = 2e-2


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    rng = {k[4:]: z[k] for k in z.files if k.startswith("rng_")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    return z["rays"], meta, rng, out


def max_rel(new, ref):
    new, ref = np.asarray(new, np.float64), np.asarray(ref, np.float64)
    return float(np.max(np.abs(new - ref) / (np.abs(ref) + 1e-6))) if ref.size else 0.0


def max_abs(new, ref):
    new, ref = np.asarray(new, np.float64), np.asarray(ref, np.float64)
    return float(np.max(np.abs(new - ref))) if ref.size else 0.0


def check_render(res, ref, rel=REL_TOL, opa=OPACITY_ABS_TOL, tag="", floor=ABS_FLOOR):
    assert set(res.keys()) >= set(ref.keys()), (tag, sorted(res.keys()), sorted(ref.keys()))
    for k, v in ref.items():
        got = np.asarray(res[k])
        assert got.shape == v.shape, (tag, k, got.shape, v.shape)
        assert np.isfinite(got).all(), (tag, k)
        if k.startswith("opacity"):
            err = np.abs(got.astype(np.float64) - v)
            frac = float((err > opa).mean())
            assert frac <= OPACITY_OUTLIER_FRAC, f"{tag}:{k} {frac:.2e} of entries off by > {opa}"
            assert err.max() <= OPACITY_OUTLIER_MAX, f"{tag}:{k} max abs err {err.max():.3e}"
        else:
            err = np.abs(got.astype(np.float64) - v)
            bound = rel * np.abs(v.astype(np.float64)) + floor
            worst = float((err / bound).max())
            assert worst <= 1.0, f"{tag}:{k} err/bound = {worst:.3f} (max abs err {err.max():.3e})"


RENDER_CASES = ["render_lego_eval_teacher", "render_lego_eval_rawinit", "render_llff_eval_128",
                "render_lego_train_teacher", "render_lego_testtime", "render_llff_disp_coarse_only",
                "render_ragged_small"]


def well_conditioned(bins, weights, u, margin=1e-6, eps=1e-5):
    """Mask of (ray, sample) entries of ``sample_pdf`` whose result is a continuous function of the
    cdf: u is not within ``margin`` of any cdf knot.  (At a knot the searchsorted decision of
    rendering.py:46 flips on the last bit of the cdf and, when the adjacent bin mass is < eps,
    rendering.py:56 turns that into a jump of up to one bin width.)"""
    w = weights.astype(np.float64) + eps
    cdf = np.concatenate([np.zeros((w.shape[0], 1)), np.cumsum(w / w.sum(-1, keepdims=True), -1)], -1)
    dist = np.abs(cdf[:, None, 1:] - u.astype(np.float64)[:, :, None]).min(-1)   # cdf[0] == 0 exactly
    return dist > margin


def sample_pdf_tol(bins, weights, u, eps=1e-5, cdf_noise=4e-7, floor=2e-6):
    """Per-entry absolute tolerance for ``sample_pdf`` outputs: fp32 rounding noise of the cdf
    (~1 ulp at 1.0) is amplified by width/denom of the bin the sample falls into (rendering.py:59-60)."""
    w = weights.astype(np.float64) + eps
    cdf = np.concatenate([np.zeros((w.shape[0], 1)), np.cumsum(w / w.sum(-1, keepdims=True), -1)], -1)
    m = weights.shape[1]
    inds = (cdf[:, None, :] <= u.astype(np.float64)[:, :, None]).sum(-1)
    below, above = np.maximum(inds - 1, 0), np.minimum(inds, m)
    denom = np.take_along_axis(cdf, above, 1) - np.take_along_axis(cdf, below, 1)
    width = np.take_along_axis(bins.astype(np.float64), above, 1) - np.take_along_axis(bins.astype(np.float64), below, 1)
    return floor + cdf_noise / np.maximum(denom, eps) * np.abs(width)


# ------------------------------------------------------------------------------------------------------------------
# raw sn_dw_gemm task tables (test / tool infrastructure)
_VARIANT_MN = {0: (256, 256), 1: (256, 64), 2: (128, 256), 3: (128, 64), 4: (32, 256), 5: (32, 128)}
# cost of one point of a K-range on one CU, in cycles: max(MFMA issue time of the wave block, tile bytes / ~8 B/clk of
# per-CU streaming bandwidth) -- the narrow problems are DMA-bound, not MFMA-bound (measured: splitting by FLOPs alone left
# the 32x128 problem streaming 168 MB through a single CU, 2.5x the kernel time of the balanced split)
_VARIANT_COST = {0: 512, 1: 161, 2: 260, 3: 95, 4: 101, 5: 59}      # measured per-point times (tools/dw_time.py), variant 0 = 512
# bf16-operand mode: 8x less MFMA time, every variant is bound by its per-CU DMA stream -- measured per-point times again
_VARIANT_COST_BF16 = {0: 512, 1: 189, 2: 226, 3: 126, 4: 138, 5: 125}
# ... and with G / the activations stored as bf16 (gather-bound inner loop, half the bytes)
_VARIANT_COST_BF16_STATE = {0: 512, 1: 343, 2: 348, 3: 226, 4: 296, 5: 190}
_KB = 16                      # csrc/sn_dw.hip: points per staged chunk
_TARGET_WGS = 256             # exactly one workgroup per CU per launch


def dw_tasks(acts, emb, G, bf16=False):
    """Host-built task table for the raw ``sn_dw_gemm`` entry (the low-level ABI the weight-gradient tests and tools/dw_time.py
    exercise; the product path uses ``sn_weight_grads``, whose plan is built inside the library): the 14 contractions
    dW = G^T X of a network, K-split over ~one workgroup per CU in proportion to their cost.
    Returns (rows: list of 8-int64 task records, outs: [(key, partial dW, partial db)])."""
    import numpy as np
    import torch
    P = acts.shape[1]                                        # padded to a multiple of 16 (pad rows of G are zero)
    dev = acts.device
    # (key, A tensor, A col, lda, B tensor, B col, ldb, variant, want_bias)
    probs = []
    for i in range(8):                                       # xyz_encoding_{i+1}
        if i == 0:
            probs.append((("w", 0), G[0], 0, 256, emb, 0, 128, 1, True))
        else:
            probs.append((("w", i), G[i], 0, 256, acts[i - 1], 0, 256, 0, True))
            if i == 4:                                       # skip: cat([input_xyz, h4])  nerf.py:133
                probs.append((("w4e", 4), G[4], 0, 256, emb, 0, 128, 1, False))
    probs.append((("w", 8), G[8], 0, 256, acts[7], 0, 256, 0, True))          # xyz_encoding_final
    probs.append((("w", 9), G[9], 0, 256, acts[8], 0, 256, 2, True))          # dir_encoding[:, :256]
    probs.append((("w9e", 9), G[9], 0, 256, emb, 64, 128, 3, False))          # dir_encoding[:, 256:]
    # rows 0..2 = g_y of rgb, row 3 = g_y of sigma (zero-padded 32-wide block at G[9][:, 128:160], sn_mlp_bwd.hip)
    probs.append((("sig", 10), G[9], 128, 256, acts[7], 0, 256, 4, False))    # sigma  (nerf.py:136)
    probs.append((("rgb", 11), G[9], 128, 256, acts[9], 0, 256, 5, True))     # rgb    (nerf.py:144)
    cost = (_VARIANT_COST_BF16_STATE if G.dtype == torch.bfloat16 else _VARIANT_COST_BF16) if bf16 else _VARIANT_COST
    # 0x100: bf16 operands; 0x200: G and the activations are STORED as bf16 (emb stays fp32)
    state16 = G.dtype == torch.bfloat16
    assert (not state16) or (bf16 and acts.dtype == torch.bfloat16 and emb.dtype == torch.float32)
    flags = (0x100 if bf16 else 0) | (0x200 if state16 else 0)
    work = [cost[p[7]] for p in probs]
    tot = float(sum(work))
    max_split = max(1, P // (4 * _KB))
    # K-splits proportional to the work of a problem, summing to _TARGET_WGS (largest remainders get the slack)
    ideal = [_TARGET_WGS * w / tot for w in work]
    splits = [max(1, int(x)) for x in ideal]
    for j in sorted(range(len(work)), key=lambda j: ideal[j] - int(ideal[j]), reverse=True):
        if sum(splits) >= _TARGET_WGS:
            break
        splits[j] += 1
    rows, outs = [], []
    for pr, ns in zip(probs, splits):
        key, A, ac, lda, B, bc, ldb, var, want_b = pr
        M, N = _VARIANT_MN[var]
        ns = int(min(max_split, ns))
        per = -(-P // ns)
        per = -(-per // _KB) * _KB
        ns = -(-P // per)
        cpart = torch.empty((ns, M, N), dtype=torch.float32, device=dev)
        bpart = torch.empty((ns, M), dtype=torch.float32, device=dev) if want_b else None
        outs.append((key, cpart, bpart))
        a_ptr, b_ptr = A.data_ptr() + ac * A.element_size(), B.data_ptr() + bc * B.element_size()
        for j in range(ns):
            rows.append((a_ptr, b_ptr, cpart.data_ptr() + j * M * N * 4,
                         (bpart.data_ptr() + j * M * 4) if want_b else 0,
                         j * per, min(P, (j + 1) * per), lda | (ldb << 32), N | ((var | flags) << 32)))
    return rows, outs


# ---- bf16x3 training state (include/sinnerf_hip.h SN_DTYPE_BF16X3, csrc/sn_layout.h "x3 state") --------------------------------
# Slots 0..8 of `acts` / `G` hold every value as the (hi, lo) bf16 pair the kernels compute with: per 8 features 16 B of hi parts, then
# 16 B of lo parts (a row of 256 features is 1 KB like an fp32 row).  Slot 9 stays fp32.
def x3_state_decode(a):
    """(..., 256) float32 array holding split rows -> the fp32 values hi + lo"""
    u = np.ascontiguousarray(a).view(np.uint16).reshape(a.shape[:-1] + (32, 2, 8)).astype(np.uint32) << 16
    f = u.view(np.float32)
    return (f[..., 0, :] + f[..., 1, :]).reshape(a.shape)


def x3_state_encode(x):
    """fp32 values (..., 256) -> split rows (as float32 bit patterns): hi = RNE bf16(x), lo = RNE bf16(x - hi)"""
    def bf16_bits(v):
        b = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
        return (((b + 0x7FFF + ((b >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = bf16_bits(x)
    lo = bf16_bits(x - (hi.astype(np.uint32) << 16).view(np.float32))
    out = np.stack([hi.reshape(x.shape[:-1] + (32, 8)), lo.reshape(x.shape[:-1] + (32, 8))], axis=-2)
    return np.ascontiguousarray(out).reshape(x.shape[:-1] + (512,)).view(np.float32)


def x3_state_to_fp32(state):
    """(10, rows, 256) bf16x3 state -> the fp32 state an SN_DTYPE_F32 kernel reads (slots 0..8 decoded, slot 9 as is)"""
    out = np.array(state, copy=True)
    out[:9] = x3_state_decode(state[:9])
    return out
