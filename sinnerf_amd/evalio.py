"""Eval-side callers and data formats around the hot path (host code only):

* ``batched_inference``   eval.py:83-115: the ray-chunk driver of the eval script (chunk fixed to 1024*32*16 rays there,
                          perturb = 0, noise_std = 0, ``test_time=False``)
* ``save_pfm/read_pfm``   datasets/depth_utils.py:6-74: the PFM depth-map format ``eval.py:170-174`` writes
* ``to_uint8`` / ``save_png``  eval.py:182-184: ``(img_pred * 255).astype(np.uint8)`` + an 8-bit RGB PNG (imageio is not
                          available offline: a minimal stdlib-zlib PNG encoder, any PNG reader opens the result)
* ``render_frame``        eval.py:152-189 for one pose: GPU ray generation -> batched_inference -> image / depth arrays
"""
import re
import struct
import sys
import zlib
from collections import defaultdict

import numpy as np
import torch

from .rendering import render_rays

EVAL_CHUNK = 1024 * 32 * 16          # eval.py:91


@torch.no_grad()
def batched_inference(models, embeddings, rays, N_samples, N_importance, use_disp, chunk, white_back):
    """eval.py:83-115 (``chunk`` is accepted and overridden by EVAL_CHUNK exactly as there)."""
    chunk = EVAL_CHUNK
    results = defaultdict(list)
    for i in range(0, rays.shape[0], chunk):
        out = render_rays(models, embeddings, rays[i:i + chunk], N_samples, use_disp, 0, 0, N_importance, chunk, white_back,
                          test_time=False)
        for k, v in out.items():
            results[k] += [v]
    return {k: torch.cat(v, 0) for k, v in results.items()}


def read_pfm(filename):
    """datasets/depth_utils.py:6-43 -> (array flipped to top-down rows, scale)."""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        color = header == "PF"
        dim = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not dim:
            raise Exception("Malformed PFM header.")
        width, height = map(int, dim.groups())
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.frombuffer(f.read(), endian + "f4")
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(np.reshape(data, shape)), abs(scale)


def save_pfm(filename, image, scale=1):
    """datasets/depth_utils.py:46-74: rows bottom-up, scale sign = endianness, '%f' scale line."""
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    image = np.flipud(image)
    endian = image.dtype.byteorder
    if endian == "<" or (endian == "=" and sys.byteorder == "little"):
        scale = -scale
    with open(filename, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write("{} {}\n".format(image.shape[1], image.shape[0]).encode("utf-8"))
        f.write(("%f\n" % scale).encode("utf-8"))
        f.write(np.ascontiguousarray(image).tobytes())


def to_uint8(img):
    """eval.py:182: ``(img_pred * 255).astype(np.uint8)`` (truncation; inputs lie in [0, 1] up to WidenedSigmoid's 1e-3)."""
    return (np.asarray(img) * 255).astype(np.uint8)


def save_png(filename, img_u8):
    """8-bit RGB / grey PNG, filter 0 on every row."""
    img_u8 = np.ascontiguousarray(img_u8)
    if img_u8.dtype != np.uint8 or img_u8.ndim not in (2, 3) or (img_u8.ndim == 3 and img_u8.shape[2] not in (1, 3)):
        raise ValueError("save_png: uint8 (H, W), (H, W, 1) or (H, W, 3) expected")
    h, w = img_u8.shape[:2]
    ch = 1 if img_u8.ndim == 2 else img_u8.shape[2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img_u8.reshape(h, w * ch)], 1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(filename, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if ch == 3 else 0, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def render_frame(models, embeddings, c2w, H, W, focal, near, far, N_samples=64, N_importance=64, use_disp=False,
                 white_back=True):
    """One eval pose (eval.py:152-189): rays generated on the GPU, rendered, returned as
    (img (H, W, 3) float32, depth (H, W) float32 with NaNs zeroed as eval.py:168, results dict)."""
    from .ray_utils import get_rays
    dev = next(models[0].parameters()).device
    c2w = torch.as_tensor(c2w, dtype=torch.float32, device=dev)
    rays = get_rays(H, W, focal, c2w, near, far)
    res = batched_inference(models, embeddings, rays, N_samples, N_importance, use_disp, EVAL_CHUNK, white_back)
    key = "fine" if "rgb_fine" in res else "coarse"
    img = res["rgb_" + key].view(H, W, 3).cpu().numpy()
    depth = np.nan_to_num(res["depth_" + key].view(H, W).cpu().numpy())
    return img, depth, res
