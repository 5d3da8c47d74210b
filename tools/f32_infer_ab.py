#!/usr/bin/env python3
"""fp32 inference MLP launch, fine-pass shape of the bench (160 000 rays x 128 samples = 20.48 M points): the round-6 kernel
(csrc/sn_mlp_fwd_f32g.hip: fragments straight from L2, VALU-free trunk, no barrier) against the LDS-ring kernel of rounds 1-5
(csrc/sn_mlp_fwd.hip, SN_FLAG_F32_LDS_RING) -- time, fraction of the 157.3 TF fp32 MFMA peak, and BIT IDENTITY of the outputs
(full + sigma-only + pre-embedded rows).  usage: f32_infer_ab.py [reps]     (run under rocprofv3 --pmc for cycles / MFMA-busy)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O                      # noqa: E402  (input generator only)
import sinnerf_amd                                     # noqa: E402
from sinnerf_amd import rendering, _lib                # noqa: E402

RING = 4                                               # SN_FLAG_F32_LDS_RING
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
FULL_ONLY = bool(os.environ.get("F32_AB_FULL_ONLY"))     # timing variants (tools/build_variant_f32g.sh) only carry the frame render's kernel
dev = torch.device("cuda:0")
rays = torch.from_numpy(O.lego_rays(400, 400, 0)).to(dev)
torch.manual_seed(0)
z = torch.sort(torch.rand((rays.shape[0], 128), device=dev) * 4 + 2, -1)[0].contiguous()
m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype="fp32")
m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
m = m.to(dev).eval()
pts = rays.shape[0] * 128
outs = {}
with torch.no_grad():
    for name, flags in (("f32g (round 6)", 0), ("LDS ring (rounds 1-5)", RING), ("f32g (round 6) again", 0)):
        for sigma_only in ((False,) if FULL_ONLY else (False, True)):
            for _ in range(1):
                o = rendering._mlp(m, rays, z, sigma_only, flags)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                o = rendering._mlp(m, rays, z, sigma_only, flags)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            flop = (1186816 if not sigma_only else 2 * (63 * 256 + 3 * 256 * 256 + 319 * 256 + 3 * 256 * 256 + 256)) * pts
            outs[(flags, sigma_only)] = o
            print("%-24s %-10s %8.3f ms  %6.1f TFLOP/s  %.4f of 157.3 TF" % (name, "sigma-only" if sigma_only else "full", ms, flop / ms / 1e9,
                                                                         flop / ms / 1e9 / 157.3))
    for so in ((False,) if FULL_ONLY else (False, True)):
        same = torch.equal(outs[(0, so)], outs[(RING, so)])
        print("bit-identical outputs (%s): %s" % ("sigma-only" if so else "full", same))
        assert same
    if FULL_ONLY:
        sys.exit(0)
    # pre-embedded rows through NeRF.forward's entry (sn_mlp_forward_embedded)
    x = torch.randn(5000, 90, device=dev)
    a = torch.empty(5000, 4, device=dev); b = torch.empty(5000, 4, device=dev)
    for o, flags in ((a, 0), (b, RING)):
        _lib.check(_lib.lib.sn_mlp_forward_embedded(_lib.ptr(m.packed()), m.kernel_dtype(), _lib.ptr(x), 5000, 90, 0, flags,
                                                    _lib.ptr(o), _lib.stream_ptr()), "sn_mlp_forward_embedded")
    torch.cuda.synchronize()
    print("bit-identical outputs (pre-embedded rows):", torch.equal(a, b))
    assert torch.equal(a, b)
