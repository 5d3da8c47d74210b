"""Drop-in ``Embedding`` / ``NeRF`` modules (reference: ``models/nerf.py:7-41, 46-148``).

Same constructor arguments, attribute names and ``state_dict`` keys as the reference
(``xyz_encoding_{1..8}.0.{weight,bias}``, ``xyz_encoding_final.*``, ``dir_encoding.0.*``, ``sigma.*``,
``rgb.0.*`` -- the checkpoint contract of ``train.py:25-32`` / ``utils/__init__.py:60-83``), so reference
checkpoints load unchanged and ``.parameters()`` feed the reference optimisers / DDP.  The arithmetic runs in
``libsinnerf_hip.so``: ``NeRF.forward`` calls the fused MFMA kernel on a pre-embedded matrix, and
``render_rays`` reads ``NeRF.packed()`` (weights re-ordered into MFMA fragment order) directly.

There is no CPU fallback: calling these modules on CPU tensors raises.
"""
import ctypes

import torch
from torch import nn

from . import _lib

_DTYPES = {"fp32": _lib.SN_DTYPE_F32, "float32": _lib.SN_DTYPE_F32, torch.float32: _lib.SN_DTYPE_F32,
           "bf16": _lib.SN_DTYPE_BF16, "bfloat16": _lib.SN_DTYPE_BF16, torch.bfloat16: _lib.SN_DTYPE_BF16,
           # fp32-level accuracy on the bf16 matrix cores (3-term hi/lo split, csrc/sn_mlp_{fwd,bwd}_bf16x3.hip): inference, and
           # under autograd the forward, the backward chain and the weight gradients (training state: (hi, lo) bf16 pairs in fp32-sized buffers)
           "bf16x3": _lib.SN_DTYPE_BF16X3,
           # fp16 operands / fp32 accumulate at the bf16 rate (v_mfma_f32_32x32x16_f16): INFERENCE only -- 11 significand bits instead of 8,
           # 8-17x closer to the fp32 render than "bf16" on the trained-weight fixtures; activations beyond +-65504 overflow
           "fp16": _lib.SN_DTYPE_F16, "float16": _lib.SN_DTYPE_F16, torch.float16: _lib.SN_DTYPE_F16}


def dtype_code(dtype):
    try:
        return _DTYPES[dtype]
    except KeyError:
        raise ValueError(f"unsupported compute dtype {dtype!r} (use 'fp32', 'bf16', 'bf16x3' or -- inference only -- 'fp16')")


class Embedding(nn.Module):
    """``Embedding(in_channels, N_freqs, logscale=True)`` -- reference ``models/nerf.py:7-41``.

    x -> (x, sin(2^k x), cos(2^k x), ...).  Inside ``render_rays`` the embedding is fused into the MLP kernel
    (never materialised); this module exists for API compatibility and stand-alone use.  Its forward is plain
    torch elementwise code on whatever device ``x`` lives on (it is not part of the accelerated hot path).
    """

    def __init__(self, in_channels, N_freqs, logscale=True):
        super().__init__()
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.funcs = [torch.sin, torch.cos]
        self.out_channels = in_channels * (len(self.funcs) * N_freqs + 1)
        if logscale:
            self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs)
        else:
            self.freq_bands = torch.linspace(1, 2 ** (N_freqs - 1), N_freqs)
        self.logscale = logscale

    def forward(self, x):
        out = [x]
        for freq in self.freq_bands:
            for func in self.funcs:
                out.append(func(freq * x))
        return torch.cat(out, -1)


_PACK_TABLES = {}        # (device, dtype_code) -> int32 device tensor


def _pack_table(device, code):
    key = (str(device), code)
    if key not in _PACK_TABLES:
        n = _lib.lib.sn_pack_table_entries_dtype(code)
        host = torch.empty((n, 2), dtype=torch.int32)
        _lib.check(_lib.lib.sn_build_pack_table(code, ctypes.c_void_p(host.data_ptr())), "sn_build_pack_table")
        _PACK_TABLES[key] = host.to(device)
    return _PACK_TABLES[key]


def _pack_table_bwd(device, code):
    """(table, blob bytes) of the transposed-weight blob of ``sn_mlp_backward_chain``"""
    bf16 = code == _lib.SN_DTYPE_BF16
    n_entries, build, n_bytes = ((_lib.lib.sn_pack_table_entries_bwd_bf16, _lib.lib.sn_build_pack_table_bwd_bf16,
                                  _lib.lib.sn_packed_weights_bytes_bwd_bf16) if bf16 else
                                 (_lib.lib.sn_pack_table_entries_bwd_bf16x3, _lib.lib.sn_build_pack_table_bwd_bf16x3,
                                  _lib.lib.sn_packed_weights_bytes_bwd_bf16x3) if code == _lib.SN_DTYPE_BF16X3 else
                                 (_lib.lib.sn_pack_table_entries_bwd, _lib.lib.sn_build_pack_table_bwd,
                                  _lib.lib.sn_packed_weights_bytes_bwd))
    key = (str(device), "bwd", code)
    if key not in _PACK_TABLES:
        host = torch.empty((n_entries(), 2), dtype=torch.int32)
        _lib.check(build(ctypes.c_void_p(host.data_ptr())), "sn_build_pack_table_bwd")
        _PACK_TABLES[key] = host.to(device)
    return _PACK_TABLES[key], int(n_bytes())


class NeRF(nn.Module):
    """``NeRF(D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4], use_new_activation=False)``.

    Reference ``models/nerf.py:46-148``.  The HIP kernels implement the layer configuration both reference call sites
    construct (``sinnerf.py:137,140``, ``eval.py:136-137``): D=8, W=256, 63/27 inputs, skips=[4].  Any other configuration
    the reference's constructor accepts raises ``NotImplementedError`` here, at construction: there is no torch-op second
    backend in this package.  Both head variants of ``nerf.py:77-100`` are built: ``use_new_activation=True``
    (ShiftedSoftplus / WidenedSigmoid, what SinNeRF uses) and the constructor's default ``False`` (ReLU / Sigmoid).
    """

    def __init__(self, D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4], use_new_activation=False,
                 compute_dtype="fp32"):
        super().__init__()
        if (D, W, in_channels_xyz, in_channels_dir, list(skips)) != (8, 256, 63, 27, [4]):
            raise NotImplementedError(
                "sinnerf_amd.NeRF: the HIP kernels exist for NeRF(D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4]) "
                "(models/sinnerf.py:137,140, eval.py:136-137); got D=%r, W=%r, in_channels_xyz=%r, in_channels_dir=%r, skips=%r.  "
                "There is no torch-op fallback for other layer configurations."
                % (D, W, in_channels_xyz, in_channels_dir, list(skips)))
        dtype_code(compute_dtype)                                    # unknown arithmetic: ValueError here, not at the first render
        self.use_new_activation = bool(use_new_activation)
        self.D, self.W = D, W
        self.in_channels_xyz, self.in_channels_dir = in_channels_xyz, in_channels_dir
        self.skips = skips
        self.compute_dtype = compute_dtype
        for i in range(D):                                           # nerf.py:66-75
            if i == 0:
                layer = nn.Linear(in_channels_xyz, W)
            elif i in skips:
                layer = nn.Linear(W + in_channels_xyz, W)
            else:
                layer = nn.Linear(W, W)
            setattr(self, f"xyz_encoding_{i+1}", nn.Sequential(layer, nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)                    # nerf.py:76
        # nn.Identity stands in for ShiftedSoftplus / WidenedSigmoid resp. ReLU / Sigmoid (parameter-free, applied
        # in-kernel): keeps the state_dict keys "dir_encoding.0.*" / "rgb.0.*" of nerf.py:81-100.
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), nn.Identity())
        self.sigma = nn.Linear(W, 1)
        self.rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Identity())
        self._packed = {}          # dtype_code -> (blob tensor, version signature)
        self._pack_generation = 0  # bumped by invalidate_packed(): writes through .data do not bump Parameter._version

    def kernel_dtype(self, code=None):
        """The ``dtype`` argument of the ``sn_mlp_*`` entry points for this network: the arithmetic code (default: of
        ``compute_dtype``) plus the head-variant flag."""
        code = dtype_code(self.compute_dtype) if code is None else code
        return code if self.use_new_activation else code | _lib.SN_DTYPE_CLASSIC_HEADS

    # ---- packed weights -------------------------------------------------------------------------------
    def raw_tensors(self):
        """The 24 parameter tensors in the order of ``include/sinnerf_hip.h`` (= state_dict order)."""
        out = []
        for i in range(self.D):
            lin = getattr(self, f"xyz_encoding_{i+1}")[0]
            out += [lin.weight, lin.bias]
        out += [self.xyz_encoding_final.weight, self.xyz_encoding_final.bias,
                self.dir_encoding[0].weight, self.dir_encoding[0].bias,
                self.sigma.weight, self.sigma.bias, self.rgb[0].weight, self.rgb[0].bias]
        return out

    def _signature(self, raws):
        return (self._pack_generation,) + tuple((t.data_ptr(), t._version) for t in raws)

    def invalidate_packed(self):
        """Mark the MFMA-packed weight blobs stale.  ``packed()`` notices in-place updates of the parameters through
        ``Parameter._version`` (optimizer steps, ``load_state_dict``, ``p.mul_()`` under ``no_grad``) and replaced storage
        through ``data_ptr``; a write THROUGH ``p.data`` (``dist.broadcast(p.data)``, ``p.data.copy_()``, EMA / clipping code,
        a fused optimiser writing a flat buffer the parameters are views of) bumps neither -- call this after such a write.
        ``parallel.broadcast_parameters`` and ``optim.FlatAdam`` do.  The blob tensors are kept and re-filled in place."""
        self._pack_generation += 1

    def packed(self, dtype=None):
        """uint8 device tensor holding the weights in MFMA-fragment order (``csrc/sn_layout.h``); rebuilt
        (one small gather kernel) whenever a parameter was modified in place or replaced."""
        code = dtype_code(dtype if dtype is not None else self.compute_dtype)
        raws = self.raw_tensors()
        dev = raws[0].device
        if dev.type != "cuda":
            raise RuntimeError("sinnerf_amd.NeRF: parameters must live on a ROCm device (no CPU fallback)")
        sig = self._signature(raws)
        hit = self._packed.get(code)
        if hit is not None and hit[1] == sig and hit[0].device == dev:
            return hit[0]
        for t in raws:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("sinnerf_amd.NeRF: parameters must be contiguous float32 master weights")
        if self._training_pack(raws) and code != _lib.SN_DTYPE_F16:      # (fp16 operands: inference only, no transposed blob)
            return self._pack_both(code, raws, dev, sig)[0]
        blob = hit[0] if (hit is not None and hit[0].device == dev) else \
            torch.empty(_lib.lib.sn_packed_weights_bytes(code), dtype=torch.uint8, device=dev)
        arr = (ctypes.c_void_p * _lib.N_RAW_TENSORS)(*[t.data_ptr() for t in raws])
        with torch.cuda.device(dev):
            table = _pack_table(dev, code)
            _lib.check(_lib.lib.sn_pack_weights(arr, _lib.ptr(table), table.shape[0], _lib.ptr(blob), code,
                                                _lib.stream_ptr()), "sn_pack_weights")
        self._packed[code] = (blob, sig)
        return blob

    @staticmethod
    def _training_pack(raws):
        # (not torch.is_grad_enabled(): the blobs are requested inside autograd.Function.forward, where grad mode is off)
        return any(t.requires_grad for t in raws)

    def _pack_both(self, code, raws, dev, sig):
        """Training: the forward blob and the transposed blob of the backward chain are re-packed after every optimizer step --
        ONE gather launch for both (the two tables concatenated, the second one's destinations shifted behind the first blob)
        instead of two launch-latency-bound ones per network."""
        fwd_t = _pack_table(dev, code)
        bwd_t, nb = _pack_table_bwd(dev, code)
        nf = int(_lib.lib.sn_packed_weights_bytes(code))
        off = (nf + 255) // 256 * 256
        key = (str(dev), "both", code)
        if key not in _PACK_TABLES:
            shifted = bwd_t.clone()
            shifted[:, 0] += off
            _PACK_TABLES[key] = torch.cat([fwd_t, shifted], 0).contiguous()
        table = _PACK_TABLES[key]
        hit = self._packed.get(("both", code))
        blob = hit if (hit is not None and hit.device == dev) else torch.empty(off + nb, dtype=torch.uint8, device=dev)
        arr = (ctypes.c_void_p * _lib.N_RAW_TENSORS)(*[t.data_ptr() for t in raws])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.sn_pack_weights(arr, _lib.ptr(table), table.shape[0], _lib.ptr(blob), code, _lib.stream_ptr()),
                       "sn_pack_weights")
        self._packed[("both", code)] = blob
        self._packed[code] = (blob[:nf], sig)
        self._packed[("bwd", code)] = (blob[off:off + nb], sig)
        return self._packed[code][0], self._packed[("bwd", code)][0]

    def packed_bwd(self, dtype="fp32"):
        """Transposed-weight blob for ``sn_mlp_backward_chain`` (fp32, or bf16 operands for the mixed-precision chain),
        cached like ``packed()``."""
        code = dtype_code(dtype)
        raws = self.raw_tensors()
        dev = raws[0].device
        if dev.type != "cuda":
            raise RuntimeError("sinnerf_amd.NeRF: parameters must live on a ROCm device (no CPU fallback)")
        sig = self._signature(raws)
        slot = ("bwd", code)
        hit = self._packed.get(slot)
        if hit is not None and hit[1] == sig and hit[0].device == dev:
            return hit[0]
        if self._training_pack(raws):
            return self._pack_both(code, raws, dev, sig)[1]
        table, n_bytes = _pack_table_bwd(dev, code)
        blob = hit[0] if (hit is not None and hit[0].device == dev) else \
            torch.empty(n_bytes, dtype=torch.uint8, device=dev)
        arr = (ctypes.c_void_p * _lib.N_RAW_TENSORS)(*[t.data_ptr() for t in raws])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.sn_pack_weights(arr, _lib.ptr(table), table.shape[0], _lib.ptr(blob), code,
                                                _lib.stream_ptr()), "sn_pack_weights")
        self._packed[slot] = (blob, sig)
        return blob

    # ---- nerf.py:105-148 ------------------------------------------------------------------------------
    def forward(self, x, sigma_only=False):
        """x: (B, 63[+27]) embedded input -> (B,4) [rgb, sigma] or (B,1) sigma (``nerf.py:105-148``)."""
        if not x.is_cuda:
            raise RuntimeError("sinnerf_amd.NeRF.forward: CUDA/ROCm tensors only (no CPU fallback)")
        need = self.in_channels_xyz if sigma_only else self.in_channels_xyz + self.in_channels_dir
        if x.dim() != 2 or x.shape[1] != need:
            raise RuntimeError(f"expected input of shape (B, {need}), got {tuple(x.shape)}")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .autograd import mlp_embedded_autograd
            return mlp_embedded_autograd(self, x, sigma_only)
        return self._forward_nograd(x, sigma_only)

    def _forward_nograd(self, x, sigma_only):
        x = x.contiguous().float()
        code = dtype_code(self.compute_dtype)
        out = torch.empty((x.shape[0], 1 if sigma_only else 4), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib.sn_mlp_forward_embedded(_lib.ptr(self.packed()), self.kernel_dtype(code), _lib.ptr(x), x.shape[0],
                                                        x.shape[1], int(sigma_only), 0, _lib.ptr(out),
                                                        _lib.stream_ptr()), "sn_mlp_forward_embedded")
        return out
