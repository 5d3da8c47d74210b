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
