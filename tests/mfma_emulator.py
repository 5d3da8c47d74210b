"""Register-level numpy emulation of the fused MLP kernel's dataflow (csrc/sn_mlp_fwd.hip), CPU only.

It follows the kernel slab by slab with the documented gfx950 MFMA semantics
(/opt/skills/guides/cdna_hip_programming.md §3):
  v_mfma_f32_32x32x2_f32  A: lane l holds A[i=l&31][k=l>>5] ; B: lane l holds B[k=l>>5][j=l&31]
  C/D: lane l, register r  <->  D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
and consumes the packed blob produced by the library's own pack table, so it validates sn_layout.h (slot maps,
K permutation, slab order, bias order) and the kernel's schedule against the oracle without a GPU.
"""
import ctypes

import numpy as np

LANES = np.arange(64)
J, H = LANES & 31, LANES >> 5


def acc_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma_f32_32x32x2(a, b, acc):
    """a, b: (64,) per-lane operand registers; acc: (64,16)."""
    A = np.zeros((32, 2), np.float32); A[J, H] = a
    B = np.zeros((2, 32), np.float32); B[H, J] = b
    D = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    for r in range(16):
        acc[:, r] += D[acc_row(r, H), J]


def fma32(a, b, c):
    """fp32 fused multiply-add (product exact in float64, one rounding of the sum to fp32 up to double rounding)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def pack_blob(lib, params, dtype_code=0):
    """Run the library's pack table on host numpy arrays (what pack_kernel does on the device)."""
    order = ([f"xyz_encoding_{i+1}.0" for i in range(8)] + ["xyz_encoding_final", "dir_encoding.0", "sigma", "rgb.0"])
    raws = []
    for k in order:
        raws += [params[k + ".weight"].reshape(-1), params[k + ".bias"].reshape(-1)]
    n = lib.sn_pack_table_entries()
    table = np.empty((n, 2), np.int32)
    assert lib.sn_build_pack_table(dtype_code, ctypes.c_void_p(table.ctypes.data)) == 0
    nbytes = lib.sn_packed_weights_bytes(dtype_code)
    assert dtype_code == 0
    blob = np.zeros(nbytes // 4, np.float32)
    dst, src = table[:, 0] // 4, table[:, 1]
    vals = np.zeros(n, np.float32)
    ok = src >= 0
    tid, off = (src[ok] >> 20) & 0x3FF, src[ok] & 0xFFFFF
    flat = np.concatenate(raws)
    starts = np.cumsum([0] + [r.size for r in raws])[:-1]
    vals[ok] = flat[starts[tid] + off]
    assert len(np.unique(dst)) == n            # every blob element written exactly once
    blob[dst] = vals
    return blob, table


def emulate_tile(lib, blob, x_embedded, sigma_only=False):
    """x_embedded: (32, 90) rows = points of one tile.  Returns (32,4) or (32,).  Mirrors csrc/sn_mlp_fwd.hip:
    76 MFMA slabs (L0, L1-7, FIN, DIR) + the sigma / rgb heads applied from the aux table in K-slot order."""
    from oracle.oracle_np import shifted_softplus, widened_sigmoid
    n_slabs = lib.sn_layout_n_slabs()
    slab_k = [lib.sn_layout_slab_k(s) for s in range(n_slabs)]
    slab_off = np.cumsum([0] + [32 * k for k in slab_k])
    tail = blob[slab_off[-1]:]
    bias = tail[:n_slabs * 32].reshape(n_slabs, 2, 16)
    aux = tail[n_slabs * 32:]
    sig_w, rgb_w, head_b = aux[:256].reshape(2, 128), aux[256:640].reshape(3, 2, 64), aux[640:644]

    xe = np.zeros((64, 32), np.float32)
    for e in range(32):
        for h in (0, 1):
            c = lib.sn_layout_xyz_slot_col(h, e)
            if c >= 0:
                xe[H == h, e] = x_embedded[:, c]
    de = np.zeros((64, 16), np.float32)
    for e in range(16):
        for h in (0, 1):
            c = lib.sn_layout_dir_slot_col(h, e)
            if c >= 0:
                de[H == h, e] = x_embedded[:, 63 + c]

    def run_slab(s, segments):
        acc = bias[s][H].copy()                                     # load_bias(): [h][r]
        frags = blob[slab_off[s]:slab_off[s + 1]].reshape(-1, 64, 4)   # [group][lane][4]
        g = 0
        for b in segments:
            for q in range(0, b.shape[1], 4):
                for jj in range(4):
                    mfma_f32_32x32x2(frags[g, :, jj], b[:, q + jj], acc)
                g += 1
        assert g == frags.shape[0]
        return acc

    def head(w_hq, b):                                              # VALU head: per-half partial dot + cross-half add
        part = (w_hq[H] * b).sum(1, dtype=np.float64)
        return (part[:32] + part[32:]).astype(np.float32)

    s = 0
    nxt = np.zeros((64, 128), np.float32)
    for t in range(8):
        nxt[:, 16 * t:16 * t + 16] = np.maximum(run_slab(s, [xe]), 0); s += 1
    hid = nxt.copy()
    for l in range(1, 8):
        for t in range(8):
            nxt[:, 16 * t:16 * t + 16] = np.maximum(run_slab(s, [xe, hid] if l == 4 else [hid]), 0); s += 1
        hid = nxt.copy()
    sigma = head(sig_w, hid) + head_b[0]
    if sigma_only:
        return sigma
    for t in range(8):
        nxt[:, 16 * t:16 * t + 16] = run_slab(s, [hid]); s += 1
    hid = nxt.copy()
    h2 = np.zeros((64, 64), np.float32)
    for t in range(4):
        h2[:, 16 * t:16 * t + 16] = shifted_softplus(run_slab(s, [hid, de])); s += 1
    assert s == n_slabs
    rgb = np.stack([widened_sigmoid(head(rgb_w[c], h2) + head_b[1 + c]) for c in range(3)], 1)
    return np.concatenate([rgb, sigma[:, None]], 1)


def pack_blob_bwd(lib, params):
    order = ([f"xyz_encoding_{i+1}.0" for i in range(8)] + ["xyz_encoding_final", "dir_encoding.0", "sigma", "rgb.0"])
    raws = []
    for k in order:
        raws += [params[k + ".weight"].reshape(-1), params[k + ".bias"].reshape(-1)]
    n = lib.sn_pack_table_entries_bwd()
    table = np.empty((n, 2), np.int32)
    assert lib.sn_build_pack_table_bwd(ctypes.c_void_p(table.ctypes.data)) == 0
    blob = np.zeros(lib.sn_packed_weights_bytes_bwd() // 4, np.float32)
    dst, src = table[:, 0] // 4, table[:, 1]
    ok = src >= 0
    flat = np.concatenate(raws)
    starts = np.cumsum([0] + [r.size for r in raws])[:-1]
    vals = np.zeros(n, np.float32)
    vals[ok] = flat[starts[(src[ok] >> 20) & 0x3FF] + (src[ok] & 0xFFFFF)]
    assert len(np.unique(dst)) == n
    blob[dst] = vals
    return blob


def emulate_bwd_tile(blob, acts, out_raw, g_raw):
    """csrc/sn_mlp_bwd.hip for one 32-point tile.  acts: dict slot -> (32,256) forward activations (h1..h8, final, h2);
    out_raw, g_raw: (32,4).  Returns G: dict slot -> (32,256) and g_out (32,4)."""
    ks = [128] * 8 + [256] * 64
    off = np.cumsum([0] + [32 * k for k in ks])
    aux = blob[off[-1] + 64:off[-1] + 64 + 640]
    assert not blob[off[-1]:off[-1] + 64].any()                 # the slab pipeline's zero "bias" slot
    rgbT = aux[:384].reshape(3, 2, 64)                          # [c][h][K-slot]
    sigT = aux[384:].reshape(2, 128)

    def tile_from_acc(acc, width_tile=None):
        # accumulator layout -> (32 points, 32 features)
        out = np.zeros((32, 32), np.float32)
        for r in range(16):
            out[J, acc_row(r, H)] = acc[:, r]
        return out

    def act_tile(slot, t):                 # (64 lanes,16 regs) values of acts[slot][:, 32t + row(r,h)]
        a = acts[slot]
        v = np.zeros((64, 16), np.float32)
        for r in range(16):
            v[:, r] = a[J, 32 * t + acc_row(r, H)]
        return v

    def run_slab(s, segments):
        acc = np.zeros((64, 16), np.float32)
        frags = blob[off[s]:off[s + 1]].reshape(-1, 64, 4)
        g = 0
        for b in segments:
            for q in range(0, b.shape[1], 4):
                for jj in range(4):
                    mfma_f32_32x32x2(frags[g, :, jj], b[:, q + jj], acc)
                g += 1
        assert g == frags.shape[0]
        return acc

    k = np.float32(0.5 * 1.002 * 0.5)
    t3 = (2 * out_raw[:, :3] - 1) / np.float32(1.002)
    gy3 = g_raw[:, :3] * k * (1 - t3 * t3)
    g_out = np.concatenate([gy3, g_raw[:, 3:4]], 1)
    G = {}
    s = 0
    g2 = np.zeros((64, 64), np.float32); G[9] = np.zeros((32, 256), np.float32)
    for t in range(4):                     # rgb.0^T on the VALU: fma chain over the three colour rows
        w = rgbT[:, H, 16 * t:16 * t + 16]                      # (3, 64 lanes, 16)
        acc = w[0] * gy3[J, 0:1]
        acc = fma32(w[1], gy3[J, 1:2], acc)
        acc = fma32(w[2], gy3[J, 2:3], acc)
        g2[:, 16 * t:16 * t + 16] = acc * (1 - np.exp(-act_tile(9, t)))
        G[9][:, 32 * t:32 * t + 32] = tile_from_acc(g2[:, 16 * t:16 * t + 16])
    gh = np.zeros((64, 128), np.float32); G[8] = np.zeros((32, 256), np.float32)
    for t in range(8):
        acc = run_slab(s, [g2]); s += 1
        gh[:, 16 * t:16 * t + 16] = acc
        G[8][:, 32 * t:32 * t + 32] = tile_from_acc(acc)
    nxt = np.zeros_like(gh); G[7] = np.zeros((32, 256), np.float32)
    for t in range(8):
        acc = run_slab(s, [gh]); s += 1
        acc = fma32(sigT[H, 16 * t:16 * t + 16], g_raw[J, 3:4], acc)        # sigma^T on the VALU, after the MFMA sum
        nxt[:, 16 * t:16 * t + 16] = np.where(act_tile(7, t) > 0, acc, 0)
        G[7][:, 32 * t:32 * t + 32] = tile_from_acc(nxt[:, 16 * t:16 * t + 16])
    gh = nxt.copy()
    for li in range(7, 0, -1):
        G[li - 1] = np.zeros((32, 256), np.float32)
        for t in range(8):
            acc = run_slab(s, [gh]); s += 1
            nxt[:, 16 * t:16 * t + 16] = np.where(act_tile(li - 1, t) > 0, acc, 0)
            G[li - 1][:, 32 * t:32 * t + 32] = tile_from_acc(nxt[:, 16 * t:16 * t + 16])
        gh = nxt.copy()
    assert s == 72
    return G, g_out
