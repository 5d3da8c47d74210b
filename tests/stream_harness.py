"""CPU harness around tools/gcn_sim.py for the generated instruction streams of the bf16 MLP kernels: builds the packed
weight blob with the library's own pack table (host code, no GPU), the LDS image and the register inputs of one workgroup
(4 waves x 2 point tiles x 32 points) exactly as the kernel sources set them up, runs the generated asm statement in the
simulator and hands back registers / memory for comparison with the numpy oracle."""
import ctypes
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gcn_sim as G                                   # noqa: E402
from oracle import oracle_np as O                     # noqa: E402

LANE = np.arange(64)
LJ, LH = LANE & 31, LANE >> 5
ORDER = [f"xyz_encoding_{i+1}.0" for i in range(8)] + ["xyz_encoding_final", "dir_encoding.0", "sigma", "rgb.0"]
BLOB_BASE, ACTS_BASE, EMB_BASE = 0x10000000, 0x40000000, 0x7000000000
V3_SLOT_BYTES, V3_SLOTS = 20480, 7
V3_TAIL_OFF = V3_SLOT_BYTES * V3_SLOTS
BIAS_FLOATS, AUX_SIGW, TAIL_FLOATS = 76 * 32, 0, 76 * 32 + 648


def load_tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def acc_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def hid_slot_feature(q, h):
    return 32 * (q >> 4) + acc_row(q & 15, h)


def apply_table(table, raws, nbytes, weight_bytes):
    """what pack_kernel does (csrc/sn_api.hip): blob[dst] = raw[src], weights rounded to bf16 RNE when weight_bytes == 2,
    entries flagged fp32 (biases, aux table) stored as fp32"""
    blob = np.zeros(nbytes, np.uint8)
    flat = np.concatenate([r.reshape(-1).astype(np.float32) for r in raws])
    starts = np.cumsum([0] + [r.size for r in raws])[:-1]
    dst, src = table[:, 0].astype(np.int64), table[:, 1].astype(np.int64)
    val = np.zeros(len(dst), np.float32)
    ok = src >= 0
    tid, off = (src[ok] >> 20) & 0xFF, src[ok] & 0xFFFFF
    val[ok] = flat[starts[tid] + off]
    is_lo = np.zeros(len(dst), bool)
    is_lo[ok] = (src[ok] & (1 << 29)) != 0                      # DT_BF16X3: the lo part, RNE(w - float(RNE(w)))  (sn_layout.h SRC_LO_FLAG)
    val[is_lo] = val[is_lo] - G.bf16_to_f32(G.bf16_rne(val[is_lo]))
    as_f32 = (src == -2) | (ok & ((src & (1 << 30)) != 0))
    if weight_bytes == 4:
        as_f32[:] = True
    b32 = val.view(np.uint32)
    for k in range(4):
        blob[dst[as_f32] + k] = ((b32[as_f32] >> (8 * k)) & 0xFF).astype(np.uint8)
    w = ~as_f32
    b16 = G.bf16_rne(val[w])
    blob[dst[w]] = (b16 & 0xFF).astype(np.uint8)
    blob[dst[w] + 1] = (b16 >> 8).astype(np.uint8)
    return blob


def forward_blob_bf16(params):
    from sinnerf_amd import _lib
    lib = _lib.lib
    raws = []
    for k in ORDER:
        raws += [params[k + ".weight"], params[k + ".bias"]]
    n = lib.sn_pack_table_entries()
    table = np.empty((n, 2), np.int32)
    assert lib.sn_build_pack_table(1, ctypes.c_void_p(table.ctypes.data)) == 0
    return apply_table(table, raws, lib.sn_packed_weights_bytes(1), 2)


def slab_k(s):
    return 64 if s < 8 else 256 if s < 32 else 320 if s < 40 else 256 if s < 72 else 288


def slab_byte_offset(s):
    return sum(slab_k(i) * 64 for i in range(s))


def xyz_lane_slots(x_emb):
    """(64 points of a wave: [pt][32], 63) embedded xyz -> per point tile the (64 lanes, 32 slots) fp32 values a lane computes"""
    from sinnerf_amd import _lib
    out = np.zeros((2, 64, 32), np.float32)
    for pt in range(2):
        for e in range(32):
            for h in (0, 1):
                c = _lib.lib.sn_layout_xyz_slot_col(h, e)
                if c >= 0:
                    out[pt, LH == h, e] = x_emb[pt * 32 + LJ[LH == h], c]
    return out


def pack8(f):
    """(64, 8) fp32 -> (4, 64) packed bf16 pairs (v_cvt_pk_bf16_f32: low = first)"""
    b = G.bf16_rne(f)
    return np.stack([b[:, 2 * i] | (b[:, 2 * i + 1] << 16) for i in range(4)], 0).astype(np.uint32)


class TrunkRun:
    """one workgroup pass of the generated trunk statement"""

    # register binding of the statement's named operands (anything below the statement's own range v128..)
    BIND = dict(va0="v32", vb="v33", vs="v34", goff="v35", sg0="v36", sg1="v37", blob="s[4:5]", wv1k="s6", em="s[8:9]",
                **{"xe%d" % i: "v[%d:%d]" % (4 * i, 4 * i + 3) for i in range(8)})

    def __init__(self, params, xyz_points, seed=0, extra_bind=None):
        """xyz_points: (256, 3) sample positions of the workgroup's point tile (wave w, point tile pt, point j)"""
        self.params = params
        self.blob = forward_blob_bf16(params)
        self.x_emb = O.embedding(xyz_points.astype(np.float32), 10)              # (256, 63)
        wg = G.Workgroup(4)
        wg.mem.add("blob", BLOB_BASE, data=self.blob.tobytes(), writable=False)
        # LDS as the kernel prologue leaves it: slabs 0..4 in ring slots 0..4, bias + aux table behind the ring
        for s in range(5):
            o = slab_byte_offset(s)
            wg.lds.b[s * V3_SLOT_BYTES: s * V3_SLOT_BYTES + 4096] = self.blob[o:o + 4096]
        tail0 = slab_byte_offset(76)
        wg.lds.b[V3_TAIL_OFF: V3_TAIL_OFF + TAIL_FLOATS * 4] = self.blob[tail0: tail0 + TAIL_FLOATS * 4]
        for w, wave in enumerate(wg.waves):
            slots = xyz_lane_slots(self.x_emb[64 * w: 64 * w + 64])
            for ks in range(4):
                for pt in range(2):
                    i = ks * 2 + pt
                    wave.v[4 * i: 4 * i + 4] = pack8(slots[pt][:, 8 * ks: 8 * ks + 8])
            wave.v[32] = LANE * 16
            wave.v[33] = V3_TAIL_OFF + LH * 64
            wave.v[34] = V3_TAIL_OFF + (BIAS_FLOATS + AUX_SIGW + LH * 128) * 4
            wave.v[35] = (w * 64 + LANE) * 16 + 5 * 4096
            wave.v[36] = 0
            wave.v[37] = 0
            wave.s[4], wave.s[5] = BLOB_BASE & 0xFFFFFFFF, BLOB_BASE >> 32
            wave.s[6] = w * 1024
            wave.vm = [("store", None)] * 4          # the generator's entry assumption: at most 4 older pieces in flight
        self.wg = wg
        self.bind = dict(self.BIND, **(extra_bind or {}))

    def run(self, lines):
        self.wg.run(G.bind(lines, self.bind))
        return self

    def activation_set(self, st):
        """the 256-feature bf16 activation held in AGPR set st after the statement: (256 points, 256 features) fp32"""
        out = np.zeros((256, 256), np.float32)
        for w, wave in enumerate(self.wg.waves):
            for ks in range(16):
                for pt in range(2):
                    base = st * 128 + (ks * 2 + pt) * 4
                    for i in range(4):
                        word = wave.a[base + i]
                        for e, bits in enumerate((word & 0xFFFF, word >> 16)):
                            q = 8 * ks + 2 * i + e
                            for h in (0, 1):
                                sel = LH == h
                                out[64 * w + 32 * pt + LJ[sel], hid_slot_feature(q, h)] = G.bf16_to_f32(bits[sel])
        return out

    def sigma(self):
        """sigma head as the kernel finishes it: both lane halves' partial sums + bias"""
        tail0 = slab_byte_offset(76)
        aux = self.blob[tail0 + BIAS_FLOATS * 4: tail0 + TAIL_FLOATS * 4].view(np.float32)
        out = np.zeros(256, np.float32)
        for w, wave in enumerate(self.wg.waves):
            for pt, reg in ((0, 36), (1, 37)):
                sg = wave.v[reg].view(np.float32)
                out[64 * w + 32 * pt + np.arange(32)] = sg[:32] + sg[32:] + aux[640]
        return out


def oracle_trunk(params, x_emb):
    """bf16-operand forward of the trunk in numpy: returns (h8 after ReLU (fp32), final, sigma) with the kernel's roundings"""
    cache = {}
    x = np.concatenate([x_emb, np.zeros((x_emb.shape[0], 27), np.float32)], 1)
    with O.bf16_operands():
        O.nerf_forward(params, x, cache=cache)
    sig = cache["h8"].astype(np.float64) @ params["sigma.weight"].astype(np.float64).T + params["sigma.bias"]
    return cache, sig[:, 0].astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# STORE MODE (training forward, tools/gen_bf16_trunk.py store=1 / csrc/sn_mlp_fwd_bf16_t.hip)
T_SLOTS = 4
T_TAIL_OFF = T_SLOTS * V3_SLOT_BYTES                    # 81920
T_STAGE_OFF = 94464                                     # 256-byte aligned behind the bias / aux table
T_STAGE_WAVE = 9216
SLOT_ROWS = 256


class TrainTrunkRun(TrunkRun):
    BIND = dict(TrunkRun.BIND, stw="v38", str0="v39", vo="v41", sm="s10", aplo="s12", aphi="s13",
                sglo="s14", sghi="s15", srlo="s16", srhi="s17", vsk="s18")

    def __init__(self, params, xyz_points):
        self.params = params
        self.blob = forward_blob_bf16(params)
        self.x_emb = O.embedding(xyz_points.astype(np.float32), 10)
        wg = G.Workgroup(4)
        wg.mem.add("blob", BLOB_BASE, data=self.blob.tobytes(), writable=False)
        self.acts = wg.mem.add("acts", ACTS_BASE, nbytes=10 * SLOT_ROWS * 512)
        self.acts[:] = 0xEE
        for s in range(3):                               # slabs 0..2 staged by the previous tile's dir section / the prologue
            o = slab_byte_offset(s)
            wg.lds.b[s * V3_SLOT_BYTES: s * V3_SLOT_BYTES + 4096] = self.blob[o:o + 4096]
        tail0 = slab_byte_offset(76)
        wg.lds.b[T_TAIL_OFF: T_TAIL_OFF + TAIL_FLOATS * 4] = self.blob[tail0: tail0 + TAIL_FLOATS * 4]
        g, k = LANE >> 3, LANE & 7
        b3, hh, e = k >> 2, (k >> 1) & 1, k & 1
        for w, wave in enumerate(wg.waves):
            slots = xyz_lane_slots(self.x_emb[64 * w: 64 * w + 64])
            for ks in range(4):
                for pt in range(2):
                    i = ks * 2 + pt
                    wave.v[4 * i: 4 * i + 4] = pack8(slots[pt][:, 8 * ks: 8 * ks + 8])
            sb = T_STAGE_OFF + w * T_STAGE_WAVE
            wave.v[32] = LANE * 16
            wave.v[33] = T_TAIL_OFF + LH * 64
            wave.v[34] = T_TAIL_OFF + (BIAS_FLOATS + AUX_SIGW + LH * 128) * 4
            wave.v[35] = (w * 64 + LANE) * 16 + 3 * 4096
            wave.v[36] = 0
            wave.v[37] = 0
            wave.v[38] = sb + LJ * 32 + 16 * (LH ^ ((LJ >> 2) & 1))
            wave.v[39] = sb + b3 * 2304 + e * 1152 + g * 32 + 16 * (hh ^ ((g >> 2) & 1))
            wave.v[40] = wave.v[39]
            wave.v[41] = g * 512 + 16 * k
            wave.v[42] = LANE * 4
            wave.s[4], wave.s[5] = BLOB_BASE & 0xFFFFFFFF, BLOB_BASE >> 32
            wave.s[6] = w * 1024
            wave.s[10] = 0x80008000
            wave.s[18] = (4 * BIAS_FLOATS - 7 * T_TAIL_OFF) & 0xFFFFFFFF
            ap = ACTS_BASE + (64 * w) * 512
            sg = ACTS_BASE + ((9 * SLOT_ROWS + 64 * w) * 256 + 128) * 2
            wave.s[12], wave.s[13] = ap & 0xFFFFFFFF, ap >> 32
            wave.s[14], wave.s[15] = sg & 0xFFFFFFFF, sg >> 32
            wave.s[16], wave.s[17] = (SLOT_ROWS * 512) & 0xFFFFFFFF, (SLOT_ROWS * 512) >> 32
            wave.vm = [("store", None)] * 2
        self.wg = wg
        self.bind = dict(self.BIND)

    def stored(self, slot):
        """acts[slot] as fp32: (256 points, 256 features)"""
        a = self.acts.view(np.uint16).reshape(10, SLOT_ROWS, 256)[slot].astype(np.uint32)
        return G.bf16_to_f32(a)

    def sign_words(self):
        """(256 points -> wave, 64 rows = 8 layer + tile, 64 lanes) uint32"""
        a = self.acts.view(np.uint32).reshape(10, SLOT_ROWS, 128)[9][:, 64:]            # columns 128..255 of slot 9 as dwords
        return a.reshape(4, 64, 64)


# ---------------------------------------------------------------------------------------------------------------------
# BACKWARD CHAIN (tools/gen_bf16_chain.py / csrc/sn_mlp_bwd_bf16_t.hip)
C_SLOT_BYTES, C_SLOTS = 16384, 4
C_TAIL_OFF = C_SLOT_BYTES * C_SLOTS                     # 65536
C_STAGE_OFF = 77312                                     # behind the 11776-byte tail (zeros + aux), 256-byte aligned
BB_ZERO_FLOATS, BB_AUX_SIGT, BB_TAIL_FLOATS = 72 * 32, 384, 72 * 32 + 640
G_BASE = 0x50000000


def backward_blob_bf16(params):
    from sinnerf_amd import _lib
    lib = _lib.lib
    raws = []
    for k in ORDER:
        raws += [params[k + ".weight"], params[k + ".bias"]]
    n = lib.sn_pack_table_entries_bwd_bf16()
    table = np.empty((n, 2), np.int32)
    assert lib.sn_build_pack_table_bwd_bf16(ctypes.c_void_p(table.ctypes.data)) == 0
    return apply_table(table, raws, lib.sn_packed_weights_bytes_bwd_bf16(), 2)


def chain_slab_bytes(s):
    return (8 if s < 8 else 16) * 1024


class ChainRun:
    BIND = dict(va0="v32", stw="v38", str0="v39", vo="v41", goff="v35", gs0="v36", gs1="v37", vst="v34", blob="s[4:5]", wv1k="s6",
                c01="s10", gplo="s12", gphi="s13", sglo="s14", sghi="s15", srlo="s16", srhi="s17")

    def __init__(self, params, acts_bytes, g_y2, g_sigma):
        """acts_bytes: the forward's acts buffer (sign words in the upper half of slot 9); g_y2: (256, 128) fp32 pre-activation
        gradient of dir_encoding (the kernel's C++ prologue computes it; packed RNE into AGPR set 0); g_sigma: (256,)"""
        self.blob = backward_blob_bf16(params)
        wg = G.Workgroup(4)
        wg.mem.add("bblob", BLOB_BASE, data=self.blob.tobytes(), writable=False)
        wg.mem.add("acts", ACTS_BASE, data=acts_bytes.tobytes(), writable=False)
        self.G = wg.mem.add("G", G_BASE, nbytes=10 * SLOT_ROWS * 512)
        self.G[:] = 0xEE
        off = 0
        for s in range(3):
            wg.lds.b[s * C_SLOT_BYTES: s * C_SLOT_BYTES + chain_slab_bytes(s)] = self.blob[off: off + chain_slab_bytes(s)]
            off += chain_slab_bytes(s)
        tail0 = sum(chain_slab_bytes(s) for s in range(72))
        wg.lds.b[C_TAIL_OFF: C_TAIL_OFF + BB_TAIL_FLOATS * 4] = self.blob[tail0: tail0 + BB_TAIL_FLOATS * 4]
        g, k = LANE >> 3, LANE & 7
        b3, hh, e = k >> 2, (k >> 1) & 1, k & 1
        gb = G.bf16_rne(g_y2)                                                    # (256, 128) bf16 bits
        for w, wave in enumerate(wg.waves):
            for ks in range(8):
                for pt in range(2):
                    base = (ks * 2 + pt) * 4                                     # set 0
                    for i in range(4):
                        word = np.zeros(64, np.uint32)
                        for ee in range(2):
                            q = 8 * ks + 2 * i + ee
                            for h in (0, 1):
                                sel = LH == h
                                word[sel] |= gb[64 * w + 32 * pt + LJ[sel], hid_slot_feature(q, h)] << (16 * ee)
                        wave.a[base + i] = word
            sb = C_STAGE_OFF + w * T_STAGE_WAVE
            wave.v[32] = LANE * 16
            wave.v[34] = C_TAIL_OFF + (BB_ZERO_FLOATS + BB_AUX_SIGT + LH * 128) * 4
            wave.v[35] = (w * 64 + LANE) * 16 + sum(chain_slab_bytes(s) for s in range(3))
            for pt, reg in ((0, 36), (1, 37)):
                wave.v[reg] = np.asarray(g_sigma[64 * w + 32 * pt + LJ], np.float32).view(np.uint32)
            wave.v[38] = sb + LJ * 32 + 16 * (LH ^ ((LJ >> 2) & 1))
            wave.v[39] = sb + b3 * 2304 + e * 1152 + g * 32 + 16 * (hh ^ ((g >> 2) & 1))
            wave.v[41] = g * 512 + 16 * k
            wave.s[4], wave.s[5] = BLOB_BASE & 0xFFFFFFFF, BLOB_BASE >> 32
            wave.s[6] = w * 1024
            wave.s[10] = 0x00010001
            gp = G_BASE + (8 * SLOT_ROWS + 64 * w) * 512
            sg = ACTS_BASE + ((9 * SLOT_ROWS + 64 * w) * 256 + 128) * 2
            wave.s[12], wave.s[13] = gp & 0xFFFFFFFF, gp >> 32
            wave.s[14], wave.s[15] = sg & 0xFFFFFFFF, sg >> 32
            wave.s[16], wave.s[17] = (SLOT_ROWS * 512) & 0xFFFFFFFF, (SLOT_ROWS * 512) >> 32
            wave.vm = [("store", None)] * 4
        self.wg = wg
        self.bind = dict(self.BIND)

    def run(self, lines):
        self.wg.run(G.bind(lines, self.bind))
        return self

    def stored(self, slot):
        a = self.G.view(np.uint16).reshape(10, SLOT_ROWS, 256)[slot].astype(np.uint32)
        return G.bf16_to_f32(a)


# ---------------------------------------------------------------------------------------------------------------------
# bf16x3 TRAINING forward trunk (tools/gen_x3_trunk.py / csrc/sn_mlp_fwd_bf16x3_t.hip): one 32-point tile per wave, 3 ring slots
# that ROTATE from tile to tile (operands), tail at LDS offset 0, "x3 state" ((hi, lo) pairs) in acts
X3_TAIL_BYTES, X3_SLOT_BYTES = 12544, 40960
X3_STAGE_OFF = X3_TAIL_BYTES + 3 * X3_SLOT_BYTES               # 135424
X3_ROWS = 128


def forward_blob_bf16x3(params):
    from sinnerf_amd import _lib
    lib = _lib.lib
    raws = []
    for k in ORDER:
        raws += [params[k + ".weight"], params[k + ".bias"]]
    n = lib.sn_pack_table_entries_dtype(3)
    table = np.empty((n, 2), np.int32)
    assert lib.sn_build_pack_table(3, ctypes.c_void_p(table.ctypes.data)) == 0
    return apply_table(table, raws, lib.sn_packed_weights_bytes(3), 2)


def split8(f):
    """(64, 8) fp32 -> ((4, 64) packed hi pairs, (4, 64) packed lo pairs): csrc/sn_mlp_x3.h x3_split8"""
    hi = G.bf16_rne(f)
    lo = G.bf16_rne((f - G.bf16_to_f32(hi)).astype(np.float32))
    pk = lambda b: np.stack([b[:, 2 * i] | (b[:, 2 * i + 1] << 16) for i in range(4)], 0).astype(np.uint32)
    return pk(hi), pk(lo)


class X3TrunkRun:
    """one workgroup pass (4 waves x 32 points) of the generated bf16x3 training trunk, with the ring rotated by `rot` slots"""
    BIND = dict(vaA="v32", vaB="v33", vaC="v34", vb="v35", vs="v36", stw0="v37", stw1="v38", stw2="v39", stw3="v40", str="v41", vsg="v42",
                goff="v43", sg="v44", vo="v45", blob="s[4:5]", smA="s6", smB="s7", smC="s8", c01="s9",
                aplo="s12", aphi="s13", sglo="s14", sghi="s15", srlo="s16", srhi="s17",
                **{"xh%d" % i: "v[%d:%d]" % (4 * i, 4 * i + 3) for i in range(4)},
                **{"xl%d" % i: "v[%d:%d]" % (16 + 4 * i, 16 + 4 * i + 3) for i in range(4)})

    def __init__(self, params, xyz_points, rot=1):
        from sinnerf_amd import _lib
        self.params = params
        self.blob = forward_blob_bf16x3(params)
        self.x_emb = O.embedding(xyz_points.astype(np.float32), 10)              # (128, 63)
        wg = G.Workgroup(4)
        wg.mem.add("blob", BLOB_BASE, data=self.blob.tobytes(), writable=False)
        self.acts = wg.mem.add("acts", ACTS_BASE, nbytes=10 * X3_ROWS * 1024)
        self.acts[:] = 0xEE
        x3_off = lambda s: sum(slab_k(i) * 128 for i in range(s))
        slots = [(rot + k) % 3 for k in range(3)]
        for s in range(2):                               # slabs 0, 1 staged by the previous tile's dir section / the prologue
            o, n = x3_off(s), slab_k(s) * 128
            wg.lds.b[X3_TAIL_BYTES + slots[s] * X3_SLOT_BYTES: X3_TAIL_BYTES + slots[s] * X3_SLOT_BYTES + n] = self.blob[o:o + n]
        tail0 = x3_off(76)
        wg.lds.b[0: TAIL_FLOATS * 4] = self.blob[tail0: tail0 + TAIL_FLOATS * 4]
        g, k = LANE >> 3, LANE & 7
        for w, wave in enumerate(wg.waves):
            f = np.zeros((64, 32), np.float32)
            for e in range(32):
                for h in (0, 1):
                    c = _lib.lib.sn_layout_xyz_slot_col(h, e)
                    if c >= 0:
                        f[LH == h, e] = self.x_emb[32 * w + LJ[LH == h], c]
            for ks in range(4):
                hi, lo = split8(f[:, 8 * ks: 8 * ks + 8])
                wave.v[4 * ks: 4 * ks + 4] = hi
                wave.v[16 + 4 * ks: 16 + 4 * ks + 4] = lo
            sb = X3_STAGE_OFF + w * 4608                 # (a wave's trunk tile inside its own 4 608-byte dir-section tile)
            for i in range(3):
                wave.v[32 + i] = X3_TAIL_BYTES + slots[i] * X3_SLOT_BYTES + LANE * 16
                wave.s[6 + i] = X3_TAIL_BYTES + slots[i] * X3_SLOT_BYTES + w * 1024
            wave.v[35] = LH * 64
            wave.v[36] = 4 * (BIAS_FLOATS + AUX_SIGW + 128 * LH)
            stw0 = sb + LJ * 128 + 16 * (LH ^ (LJ & 7))
            for i in range(4):
                wave.v[37 + i] = stw0 ^ (32 * i)
            wave.v[41] = sb + g * 128 + 16 * (k ^ g)
            wave.v[42] = (LANE >> 4) * 1024 + (LANE & 15) * 16
            wave.v[43] = (w * 64 + LANE) * 16 + x3_off(2)
            wave.v[44] = 0
            wave.v[45] = g * 1024 + 16 * k
            wave.s[4], wave.s[5] = BLOB_BASE & 0xFFFFFFFF, BLOB_BASE >> 32
            wave.s[9] = 0x00010001
            ap = ACTS_BASE + (32 * w) * 1024
            sg = ACTS_BASE + ((9 * X3_ROWS + 32 * w) * 256 + 128) * 4
            wave.s[12], wave.s[13] = ap & 0xFFFFFFFF, ap >> 32
            wave.s[14], wave.s[15] = sg & 0xFFFFFFFF, sg >> 32
            wave.s[16], wave.s[17] = (X3_ROWS * 1024) & 0xFFFFFFFF, (X3_ROWS * 1024) >> 32
            wave.vm = [("store", None)] * 2              # the generator's entry assumption: at most the two pieces of slab 1 in flight
        self.wg = wg
        self.vo0 = [w_.v[45].copy() for w_ in wg.waves]

    def run(self, lines):
        self.wg.run(G.bind(lines, self.BIND))
        return self

    def state(self):
        """(10, 128 rows, 256) float32 view of acts (slots 0..8 still as (hi, lo) pairs: decode with tests.helpers.x3_state_decode)"""
        return self.acts.view(np.float32).reshape(10, X3_ROWS, 256)

    def agpr_set(self, st):
        """the activation held in AGPR set st as (128 points, 256 features) fp32 = hi + lo"""
        out = np.zeros((2, 128, 256), np.float32)
        for w, wave in enumerate(self.wg.waves):
            for part in range(2):
                for ks in range(16):
                    base = st * 128 + part * 64 + ks * 4
                    for i in range(4):
                        word = wave.a[base + i]
                        for e, bits in enumerate((word & 0xFFFF, word >> 16)):
                            q = 8 * ks + 2 * i + e
                            for h in (0, 1):
                                sel = LH == h
                                out[part, 32 * w + LJ[sel], hid_slot_feature(q, h)] = G.bf16_to_f32(bits[sel])
        return out[0] + out[1]

    def sigma(self):
        tail0 = sum(slab_k(i) * 128 for i in range(76))
        aux = self.blob[tail0 + BIAS_FLOATS * 4: tail0 + TAIL_FLOATS * 4].view(np.float32)
        out = np.zeros(128, np.float32)
        for w, wave in enumerate(self.wg.waves):
            sg = wave.v[44].view(np.float32)
            out[32 * w + np.arange(32)] = sg[:32] + sg[32:] + aux[640]
        return out


# ---- bf16x3 backward chain (tools/gen_x3_chain.py / csrc/sn_mlp_bwd_bf16x3_t.hip): 4 static slots of 32 KB, tail at LDS offset 0
XC_SLOT, XC_RING_OFF = 32768, 11776
XC_STAGE_OFF = XC_RING_OFF + 4 * XC_SLOT                          # 142848


def backward_blob_bf16x3(params):
    from sinnerf_amd import _lib
    lib = _lib.lib
    raws = []
    for k in ORDER:
        raws += [params[k + ".weight"], params[k + ".bias"]]
    n = lib.sn_pack_table_entries_bwd_bf16x3()
    table = np.empty((n, 2), np.int32)
    assert lib.sn_build_pack_table_bwd_bf16x3(ctypes.c_void_p(table.ctypes.data)) == 0
    return apply_table(table, raws, lib.sn_packed_weights_bytes_bwd_bf16x3(), 2)


def xc_slab_bytes(s):
    return (8 if s < 8 else 16) * 2048


class X3ChainRun:
    """one workgroup pass (4 waves x 32 points) of the generated bf16x3 backward-chain statement"""
    BIND = dict(vaA="v32", vaB="v33", vs="v34", goff="v35", gsig="v36", stw0="v37", stw1="v38", stw2="v39", stw3="v40", str="v41", vo="v42",
                blob="s[4:5]", sm="s6", gplo="s12", gphi="s13", srlo="s16", srhi="s17",
                **{"sw%d_%d" % (l, w): "v%d" % (64 + 4 * l + w) for l in range(8) for w in range(4)})

    def __init__(self, params, fwd_state, g_y2, g_sigma):
        """fwd_state: (10, 128, 256) float32 view of the forward's acts (sign words in slot 9, columns 128..191); g_y2: (128, 128) fp32
        pre-activation gradient of dir_encoding (the kernel's C++ prologue computes it and splits it into AGPR set 0); g_sigma: (128,)"""
        self.blob = backward_blob_bf16x3(params)
        wg = G.Workgroup(4)
        wg.mem.add("bblob", BLOB_BASE, data=self.blob.tobytes(), writable=False)
        self.G = wg.mem.add("G", G_BASE, nbytes=10 * X3_ROWS * 1024)
        self.G[:] = 0xEE
        off = 0
        for s in range(3):
            n = xc_slab_bytes(s)
            wg.lds.b[XC_RING_OFF + s * XC_SLOT: XC_RING_OFF + s * XC_SLOT + n] = self.blob[off: off + n]
            off += n
        tail0 = sum(xc_slab_bytes(s) for s in range(72))
        wg.lds.b[0: BB_TAIL_FLOATS * 4] = self.blob[tail0: tail0 + BB_TAIL_FLOATS * 4]
        g, k = LANE >> 3, LANE & 7
        words = np.ascontiguousarray(fwd_state[9]).view(np.uint32)[:, 128:192]        # (128 rows, 64 dwords)
        hi = G.bf16_rne(g_y2)
        lo = G.bf16_rne((g_y2 - G.bf16_to_f32(hi)).astype(np.float32))
        for w, wave in enumerate(wg.waves):
            for part, bits in ((0, hi), (1, lo)):
                for ks in range(8):
                    base = part * 64 + ks * 4                                         # set 0
                    for i in range(4):
                        word = np.zeros(64, np.uint32)
                        for ee in range(2):
                            q = 8 * ks + 2 * i + ee
                            for h in (0, 1):
                                sel = LH == h
                                word[sel] |= bits[32 * w + LJ[sel], hid_slot_feature(q, h)] << (16 * ee)
                        wave.a[base + i] = word
            for l in range(8):                                                        # the lane's four words of layer l
                rows = words[32 * w + 4 * l: 32 * w + 4 * l + 4].reshape(64, 4)       # [lane][tile pair]
                for q in range(4):
                    wave.v[64 + 4 * l + q] = rows[:, q]
            sb = XC_STAGE_OFF + w * 4608
            wave.v[32] = XC_RING_OFF + LANE * 16
            wave.v[33] = XC_RING_OFF + 2 * XC_SLOT + LANE * 16
            wave.v[34] = 4 * (BB_ZERO_FLOATS + BB_AUX_SIGT + 128 * LH)
            wave.v[35] = (w * 64 + LANE) * 16 + off
            wave.v[36] = np.asarray(g_sigma[32 * w + LJ], np.float32).view(np.uint32)
            stw0 = sb + LJ * 128 + 16 * (LH ^ (LJ & 7))
            for i in range(4):
                wave.v[37 + i] = stw0 ^ (32 * i)
            wave.v[41] = sb + g * 128 + 16 * (k ^ g)
            wave.v[42] = g * 1024 + 16 * k
            wave.s[4], wave.s[5] = BLOB_BASE & 0xFFFFFFFF, BLOB_BASE >> 32
            wave.s[6] = XC_RING_OFF + w * 1024
            gp = G_BASE + (8 * X3_ROWS + 32 * w) * 1024
            wave.s[12], wave.s[13] = gp & 0xFFFFFFFF, gp >> 32
            wave.s[16], wave.s[17] = (X3_ROWS * 1024) & 0xFFFFFFFF, (X3_ROWS * 1024) >> 32
            wave.vm = [("store", None)] * 8
        self.wg = wg
        self.goff0 = [w_.v[35].copy() for w_ in wg.waves]

    def run(self, lines):
        self.wg.run(G.bind(lines, self.BIND))
        return self

    def state(self):
        return self.G.view(np.float32).reshape(10, X3_ROWS, 256)
