#!/usr/bin/env python3
"""Generator of the hand-scheduled inner loops of the NARROW bf16-state weight-gradient problems
(sinnerf_amd/csrc/sn_dw_narrow_bf16.hip): dW[m, n] += sum over the 16 points of a staged chunk of G[p, m] X[p, n] for the
contractions of models/nerf.py:66-103 that are not 256 x 256 --

    variant 1  256 x 64   xyz_encoding_1 and the skip layer's xyz columns   A = G (bf16, 256 wide), B = embedded inputs (fp32, 64 wide)
    variant 2  128 x 256  dir_encoding[:, :256]                             A = G slot 9 (128 wide), B = xyz_encoding_final output
    variant 3  128 x 64   dir_encoding[:, 256:]                             A = G slot 9, B = embedded directions (fp32, 64 wide)
    variant 4   32 x 256  sigma                                             A = the 32-wide head block of G slot 9, B = h8
    variant 5   32 x 128  rgb                                               A = the same block, B = dir_encoding output (128 wide)
    variants 6, 7         variants 1, 3 with B = the embedded inputs stored as bf16 in K-slot order (64 wide)

Same plan as tools/gen_dw_bf16.py (one asm statement per PAIR of chunks: counted vmcnt + barrier on entry, the reads of chunk c+1
behind the MFMAs of chunk c, the DMA pieces of the chunks R-2 ahead, bias dot products), parameterised by the wave tiling
(MT x NT blocks of 32 x 32 per wave, WM x WN waves), the tile widths and the element type of the B tile.  The compiler-scheduled loop
these replace (sn_dw.hip run_task, two workgroups per CU) ran with its waves parked 74 % of the time, 12 SALU instructions per MFMA,
at 5.0 TB/s where the 256 x 256 kernel streams at 7.

  bf16 tiles   fragments by ds_read_b64_tr_b16 (two per 32 x 16 block), swizzled DMA image for tiles of >= 128 columns (sn_dw.hip)
  fp32 B tile  eight ds_read_b32 per block (points 8h .. 8h+7 of this lane's feature) + four v_cvt_pk_bf16_f32
  small tiles  a 32-wide A tile is 64 sixteen-byte pieces (wave 0 stages it), a 64-wide bf16 B tile 128 (waves 0, 1): the statement
               then exists in two forms (W0 / WX) whose DMA instruction counts, and therefore counted waits, differ

Registers of a statement: v[48:111] two fragment sets (A blocks at 48 + 32 s + 4 a, B blocks 16 further), v[112:119] address
temporaries, v[120:127] raw fp32 B words (the compiler keeps v0..v47: two workgroups share a CU, 128 VGPRs + 128 AGPRs per wave); accumulators a[16 (NT a + b) : +15]; bs0.. column sums.

usage: gen_dw_narrow.py out.inc
"""
import sys

F0, T0, RAW = 48, 112, 120                       # two waves per SIMD: a wave owns 256 of the 512 unified registers -- v0..v127 + a0..a127
KB = 16
LDS_BYTES = 81920                                   # per workgroup (two share a CU)

VARIANTS = {                                         # v: (MT, NT, WM, WN, B element bytes)
    1: (4, 1, 2, 2, 4),
    2: (2, 4, 2, 2, 2),
    3: (2, 1, 2, 2, 4),
    4: (1, 2, 1, 4, 2),
    5: (1, 1, 1, 4, 2),
    6: (4, 1, 2, 2, 2),                              # variants 1 / 3 with the embedded inputs stored as bf16 (SN_DTYPE_EMB_BF16):
    7: (2, 1, 2, 2, 2),                              # a 64-wide bf16 B tile is 128 pieces -- staged by waves 0 and 1
}


class Shape:
    def __init__(self, v):
        self.v = v
        self.MT, self.NT, self.WM, self.WN, self.EB = VARIANTS[v]
        self.WA, self.WB = self.WM * self.MT * 32, self.WN * self.NT * 32
        self.A_BYTES, self.B_BYTES = KB * self.WA * 2, KB * self.WB * self.EB
        self.BUF = self.A_BYTES + self.B_BYTES
        r = LDS_BYTES // self.BUF
        self.R = min(16, r - (r & 1))                # ring depth in chunks (even: chunks are consumed in pairs)
        self.nA_pieces, self.nB_pieces = self.A_BYTES // 16, self.B_BYTES // 16
        self.nA = (self.nA_pieces + 255) // 256       # DMA instructions per thread and chunk (wave 0 when the tile is < 256 pieces)
        self.nB = (self.nB_pieces + 255) // 256
        # a tile of fewer than 256 pieces is staged by its first waves only (64 pieces per wave): the statement then exists in two
        # forms -- W0 for the waves that stage both tiles, WX for the others -- whose DMA counts, and counted waits, differ
        self.a_waves = 4 if self.nA_pieces % 256 == 0 else self.nA_pieces // 64
        self.b_waves = 4 if self.nB_pieces % 256 == 0 else self.nB_pieces // 64
        assert self.a_waves in (1, 2, 4) and self.b_waves in (1, 2, 4) and (self.a_waves == 4 or self.b_waves == 4)
        self.two_forms = min(self.a_waves, self.b_waves) < 4

    def stages(self, full):
        """(stages A, stages B) of a wave of the full / the other class"""
        return (full or self.a_waves == 4, full or self.b_waves == 4)

    def n_dma(self, full):
        sa, sb = self.stages(full)
        return (self.nA if sa else 0) + (self.nB if sb else 0)


def fa(st, a): return F0 + 32 * st + 4 * a
def fb(st, b): return F0 + 32 * st + 16 + 4 * b
def acc(sh, a, b): return 16 * (sh.NT * a + b)


def addr_adds(sh, sl):
    out = ["v_add_u32 v%d, %%[%s], %%[ta%d]" % (T0 + i, sl, i) for i in range(sh.MT)]
    if sh.EB == 2:
        out += ["v_add_u32 v%d, %%[%s], %%[tb%d]" % (T0 + 4 + i, sl, i) for i in range(sh.NT)]
    else:
        out += ["v_add_u32 v%d, %%[%s], %%[tbf]" % (T0 + 4, sl)]
    return out


def reads(sh, st):
    """fragment reads of one chunk into set st (fp32 B: raw words, converted by cvts())"""
    out = []
    row4_a = 4 * sh.WA * 2
    for i in range(sh.MT):
        out.append("ds_read_b64_tr_b16 v[%d:%d], v%d" % (fa(st, i), fa(st, i) + 1, T0 + i))
        out.append("ds_read_b64_tr_b16 v[%d:%d], v%d offset:%d" % (fa(st, i) + 2, fa(st, i) + 3, T0 + i, row4_a))
    if sh.EB == 2:
        row4_b = 4 * sh.WB * 2
        for i in range(sh.NT):
            out.append("ds_read_b64_tr_b16 v[%d:%d], v%d" % (fb(st, i), fb(st, i) + 1, T0 + 4 + i))
            out.append("ds_read_b64_tr_b16 v[%d:%d], v%d offset:%d" % (fb(st, i) + 2, fb(st, i) + 3, T0 + 4 + i, row4_b))
    else:
        assert sh.NT == 1
        for jj in range(8):                           # point rows 8h + jj of this lane's feature
            out.append("ds_read_b32 v%d, v%d offset:%d" % (RAW + jj, T0 + 4, jj * sh.WB * 4))
    return out


def cvts(sh, st):
    if sh.EB == 2:
        return []
    return ["v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (fb(st, 0) + w, RAW + 2 * w, RAW + 2 * w + 1) for w in range(4)]


def half(sh, st, next_reads, g, wave0):
    """the MFMAs of one chunk on set st with everything else of this half behind them"""
    mf = []
    for a in range(sh.MT):
        for b in range(sh.NT):
            mf.append("v_mfma_f32_32x32x16_bf16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" % (
                acc(sh, a, b), acc(sh, a, b) + 15, fa(st, a), fa(st, a) + 3, fb(st, b), fb(st, b) + 3, acc(sh, a, b), acc(sh, a, b) + 15))
    fill = []
    if next_reads:
        fill += reads(sh, st ^ 1)
    fill += ["v_dot2c_f32_bf16 %%[bs%d], v%d, %%[one]" % (a, fa(st, a) + w) for a in range(sh.MT) for w in range(4)]
    if g is not None:
        pieces = []
        sa, sb = sh.stages(wave0)
        if sa:
            pieces += [("ga%d" % g, "oa%d" % k, k * 4096) for k in range(sh.nA)]
        if sb:
            pieces += [("gb%d" % g, "ob%d" % k, sh.A_BYTES + k * 4096) for k in range(sh.nB)]
        for base, off, lds in pieces:                 # m0 one instruction ahead of its use
            fill += ["s_add_u32 m0, %%[md%d], %d" % (g, lds), "s_nop 0", "global_load_lds_dwordx4 %%[%s], %%[%s] nt" % (off, base)]
    # deal the fill instructions evenly behind the MFMAs
    out, per = [], -(-len(fill) // max(1, len(mf)))
    for j, m in enumerate(mf):
        out.append(m)
        out.extend(fill[j * per:(j + 1) * per])
    out.extend(fill[len(mf) * per:])
    return out


def gen_pair(sh, wave0):
    n = sh.n_dma(wave0)
    out = ["s_waitcnt vmcnt(%d)" % ((sh.R - 4) * n), "s_barrier"]
    out += addr_adds(sh, "sl0") + reads(sh, 0) + ["s_waitcnt lgkmcnt(0)"] + cvts(sh, 0)
    out += addr_adds(sh, "sl1")
    if sh.EB == 4:
        out += ["s_nop 1"]                             # v_cvt_pk -> MFMA read: 2 wait states
    out += half(sh, 0, True, 0, wave0)
    out += ["s_waitcnt lgkmcnt(0)"] + cvts(sh, 1)
    if sh.EB == 4:
        out += ["s_nop 1"]
    out += half(sh, 1, False, 1, wave0)
    return out


def gen_group(sh, wave0, nch):
    """nch chunks per sync point (the shapes whose chunks are small: a workgroup's streaming rate follows the bytes it consumes per
    barrier).  Entry: the nch chunks of this group have landed -- R - 2 nch younger ones may be in flight; chunk k+1's fragments are
    read behind the MFMAs of chunk k; the DMA pieces of chunk k of the group R - nch ahead go out in half k."""
    assert sh.R % nch == 0 and sh.R >= 3 * nch
    n = sh.n_dma(wave0)
    assert (sh.R - 2 * nch) * n <= 63
    out = ["s_waitcnt vmcnt(%d)" % ((sh.R - 2 * nch) * n), "s_barrier"]
    out += addr_adds(sh, "sl0") + reads(sh, 0) + ["s_waitcnt lgkmcnt(0)"] + cvts(sh, 0)
    for k in range(nch):
        st = k & 1
        last = k + 1 == nch
        if not last:
            out += addr_adds(sh, "sl%d" % (k + 1))
        if sh.EB == 4:
            out += ["s_nop 1"]
        out += half(sh, st, not last, k, wave0)
        if not last:
            out += ["s_waitcnt lgkmcnt(0)"] + cvts(sh, st ^ 1)
    return out


QUAD = {5: 4, 7: 4}                                  # variant -> chunks per sync point (default 2: gen_pair)


def gen_tail(sh):
    out = ["s_waitcnt vmcnt(0)", "s_barrier"] + addr_adds(sh, "sl0") + reads(sh, 0) + ["s_waitcnt lgkmcnt(0)"] + cvts(sh, 0)
    if sh.EB == 4:
        out += ["s_nop 1"]
    return out + half(sh, 0, False, None, True)


def main():
    with open(sys.argv[1], "w") as f:
        f.write("// GENERATED by tools/gen_dw_narrow.py -- do not edit.\n")
        for v in sorted(VARIANTS):
            sh = Shape(v)
            forms = [("W0", True)] + ([("WX", False)] if sh.two_forms else [])
            for tag, w0 in forms:
                body = gen_pair(sh, w0)
                f.write("#define SN_DWN%d_PAIR_%s_ASM \\\n" % (v, tag))
                for l in body:
                    f.write('  "%s\\n\\t" \\\n' % l)
                f.write('  ""\n')
            if v in QUAD:
                for tag, w0 in forms:
                    f.write("#define SN_DWN%d_GROUP_%s_ASM \\\n" % (v, tag))
                    for l in gen_group(sh, w0, QUAD[v]):
                        f.write('  "%s\\n\\t" \\\n' % l)
                    f.write('  ""\n')
                f.write("#define SN_DWN%d_NCH %d\n" % (v, QUAD[v]))
            f.write("#define SN_DWN%d_TAIL_ASM \\\n" % v)
            for l in gen_tail(sh):
                f.write('  "%s\\n\\t" \\\n' % l)
            f.write('  ""\n')
            f.write("#define SN_DWN%d_ZERO_ASM \\\n" % v)
            for i in range(16 * sh.MT * sh.NT):
                f.write('  "v_accvgpr_write_b32 a%d, 0\\n\\t" \\\n' % i)
            f.write('  ""\n')
            f.write("#define SN_DWN%d_RING %d\n#define SN_DWN%d_BUF %d\n" % (v, sh.R, v, sh.BUF))
            n_mf = sum(l.startswith("v_mfma") for l in gen_pair(sh, True))
            print("dw narrow variant %d: %dx%d blocks/wave, ring %d x %d B, %d MFMAs + %d other per chunk pair"
                  % (v, sh.MT, sh.NT, sh.R, sh.BUF, n_mf, len(gen_pair(sh, True)) - n_mf))
        f.write("#define SN_DWN_VGPR_CLOBBERS " + ", ".join('"v%d"' % r for r in range(F0, 128)) + "\n")
        f.write("#define SN_DWN_AGPR_CLOBBERS " + ", ".join('"a%d"' % r for r in range(128)) + "\n")


if __name__ == "__main__":
    main()
