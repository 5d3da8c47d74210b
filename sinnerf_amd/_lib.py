"""ctypes binding of ``csrc/libsinnerf_hip.so`` (C ABI: ``include/sinnerf_hip.h``).

The product path has NO CPU fallback: if the shared library is missing or a symbol cannot be resolved the
import of this module raises, and every operator raises on non-CUDA tensors.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SINNERF_HIP_LIB: developer override used by tools/ to time experimental builds of the same ABI
LIB_PATH = os.environ.get("SINNERF_HIP_LIB") or os.path.join(_HERE, "csrc", "libsinnerf_hip.so")

SN_DTYPE_F32 = 0
SN_DTYPE_BF16 = 1
SN_DTYPE_BF16_STATE = 2
SN_DTYPE_BF16X3 = 3                 # inference + packer: fp32-level accuracy on the bf16 MFMA (3-term hi/lo split)
SN_DTYPE_F16 = 4                    # inference + packer: fp16 operands on the bf16 kernels' instruction streams (ABI 5)
SN_DTYPE_CLASSIC_HEADS = 0x100      # OR-ed into dtype: NeRF(use_new_activation=False) heads (include/sinnerf_hip.h)
SN_FLAG_F32_LDS_RING = 4             # sn_mlp_forward flags: the LDS-ring fp32 inference kernel of rounds 1-5 (A/B; default = csrc/sn_mlp_fwd_f32g.hip)
SN_DTYPE_COMPILER_SCHEDULED = 0x200 # OR-ed into dtype of the bf16-state training entries: the compiler-scheduled kernels (A/B, tests)
N_RAW_TENSORS = 24
SN_DTYPE_EMB_BF16 = 0x400           # ... emb stored as bf16 in K-slot order (hand-scheduled bf16-state kernels only)
ABI_VERSION = 5                     # == SN_ABI_VERSION of include/sinnerf_hip.h this binding was written against

c_fp = ctypes.c_void_p      # device float*
c_vp = ctypes.c_void_p
_long, _int, _float = ctypes.c_long, ctypes.c_int, ctypes.c_float

# name -> (restype, argtypes); mirrors include/sinnerf_hip.h line by line
SIGNATURES = {
    "sn_abi_version": (_int, []),
    "sn_error_string": (ctypes.c_char_p, [_int]),
    "sn_layout_xyz_slot_col": (_int, [_int, _int]),
    "sn_layout_dir_slot_col": (_int, [_int, _int]),
    "sn_layout_slab_k": (_int, [_int]),
    "sn_layout_n_slabs": (_int, []),
    "sn_packed_weights_bytes": (_long, [_int]),
    "sn_pack_table_entries": (_long, []),
    "sn_pack_table_entries_dtype": (_long, [_int]),
    "sn_build_pack_table": (_int, [_int, c_vp]),
    "sn_pack_weights": (_int, [ctypes.POINTER(c_vp), c_vp, _long, c_vp, _int, c_vp]),
    "sn_packed_weights_bytes_bwd": (_long, []),
    "sn_pack_table_entries_bwd": (_long, []),
    "sn_build_pack_table_bwd": (_int, [c_vp]),
    "sn_packed_weights_bytes_bwd_bf16": (_long, []),
    "sn_pack_table_entries_bwd_bf16": (_long, []),
    "sn_build_pack_table_bwd_bf16": (_int, [c_vp]),
    "sn_packed_weights_bytes_bwd_bf16x3": (_long, []),
    "sn_pack_table_entries_bwd_bf16x3": (_long, []),
    "sn_build_pack_table_bwd_bf16x3": (_int, [c_vp]),
    "sn_sample_coarse": (_int, [c_fp, _long, _int, _int, _float, c_fp, c_fp, c_vp]),
    "sn_mlp_forward": (_int, [c_vp, _int, c_fp, c_fp, _long, _int, _int, _int, c_fp, c_vp]),
    "sn_mlp_forward_train": (_int, [c_vp, _int, c_fp, c_fp, _long, _int, c_fp, c_fp, c_fp, _long, c_vp]),
    "sn_mlp_forward_train_embedded": (_int, [c_vp, _int, c_fp, _long, _int, c_fp, c_fp, _long, c_vp]),
    "sn_mlp_backward_chain": (_int, [c_vp, _int, c_fp, c_fp, c_fp, _long, _long, c_fp, c_fp, c_vp]),
    "sn_dw_gemm": (_int, [c_vp, _int, c_vp]),
    "sn_weight_grads_workspace_bytes": (_long, [_long, _int]),
    "sn_weight_grads": (_int, [c_vp, c_fp, c_vp, _long, _int, c_vp, ctypes.POINTER(c_vp), _int, c_vp]),
    "sn_generate_rays": (_int, [c_fp, _int, _int, _float, _float, _float, _int, _int, _int, _int, _int, _int, c_fp, c_vp]),
    "sn_adam_step": (_int, [c_fp, c_fp, c_fp, c_fp, _long, _float, _float, _float, _float, _float, _int, c_vp]),
    "sn_render_loss_workspace_bytes": (_long, []),
    "sn_render_loss": (_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_vp, _int, _long, _float, _float, c_fp, c_fp, c_fp, c_fp,
                              c_vp, c_fp, c_vp]),
    "sn_composite_backward": (_int, [c_fp, c_fp, c_fp, c_fp, _float, _long, _int, _int, c_fp, c_fp, c_fp, c_fp, c_vp]),
    "sn_mlp_forward_embedded": (_int, [c_vp, _int, c_fp, _long, _int, _int, _int, c_fp, c_vp]),
    "sn_composite_forward": (_int, [c_fp, _int, c_fp, c_fp, c_fp, _float, _long, _int, _int, c_fp, c_fp, c_fp, c_vp]),
    "sn_sample_pdf": (_int, [c_fp, c_fp, c_fp, _long, _int, _int, c_fp, c_fp, c_vp]),
    "sn_sample_pdf_bins": (_int, [c_fp, c_fp, c_fp, _long, _int, _int, _float, c_fp, c_vp]),
}


class SinnerfHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C sinnerf_amd/csrc`). There is no CPU fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError -> loud failure on a stale .so
        fn.restype, fn.argtypes = res, args
    if lib.sn_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.sn_abi_version()} != {ABI_VERSION}; rebuild the extension")
    return lib


lib = _load()


def check(code, what):
    if code != 0:
        msg = lib.sn_error_string(int(code))
        raise SinnerfHipError(f"{what} failed: [{code}] {msg.decode() if msg else '?'}")


def ptr(t):
    """device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "sinnerf_amd kernels need contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
