#!/usr/bin/env python3
"""Generator of the hand-scheduled inner loop of the fp32 256x256 weight-gradient kernel (sinnerf_amd/csrc/sn_dw_f32.hip):
dW[m, n] += sum over the 16 points of a staged chunk of G[p, m] X[p, n] on v_mfma_f32_32x32x2_f32.

The compiler-scheduled loop of sn_dw.hip reads every pair of fragments right in front of its first use behind an
s_waitcnt lgkmcnt(0) -- one exposed LDS round trip per 16 MFMAs -- and neither scheduling fences nor a rolled loop fixed that
(builtin MFMAs; measured, notes in sn_dw.hip).  Here one CHUNK (16 points: 8 k-step pairs x 16 MFMAs per wave) is ONE asm
statement laid out by hand:

  * accumulators: the whole AGPR file, block (a, b) of the wave's 4 x 4 blocks of 32 x 32 at a[16 (4a + b) : +15] (C = D in
    AGPRs, never moved); they live across the statements of a task, so the kernel file is checked by tools/check_agpr.py
    (no compiler-allocated AGPR, no scratch);
  * fragments: two sets of 4 + 4 VGPRs (v[F0 : F0+15]); the reads of pair s+1 (4 x ds_read2st64_b32: blocks a and a+2 are 256 B
    apart) are issued behind the first MFMAs of pair s, the counted wait sits in front of pair s+1's first MFMA;
  * one LDS-DMA piece of the chunk three ahead per pair (m0 + global_load_lds_dwordx4), the bias column sums (4 v_add, in ONE gap) too;
  * chunk entry: s_waitcnt vmcnt(16) (this chunk has landed, two younger ones in flight), s_barrier, the reads of pair 0.

Operands of the chunk statement (see the kernel): bs0..3 (+v) column sums; la0 la1 lb0 lb1 (v) LDS read addresses of this
chunk's slot (A / A + 128 B / B / B + 128 B, lane half and column folded in); oa0..3 ob0..3 (v) per-lane global byte offsets
of the DMA pieces; ga gb (s, 64 bit) global bases of the chunk being staged; md (s) its LDS destination + wave * 1024.

usage: gen_dw_f32.py out.inc
"""
import sys

F0 = 200                       # v[200:215]: fragment sets
A_BYTES = 16 * 256 * 4         # A tile of a chunk (16 points x 256 features fp32); the B tile follows it
MT = NT = 4


def fa(st, a): return F0 + 8 * st + a          # A fragment of block a, set st
def fb(st, b): return F0 + 8 * st + 4 + b
def acc(a, b): return 16 * (4 * a + b)


def frag_reads(st, s):
    """4 x ds_read2st64_b32 of k-step pair s into set st.  offset units = 64 dwords = 256 B: a row (point) is 1024 B = 4 units,
    pair s starts at point 2s (+ lane half, folded into the address) -> unit 8s; blocks a, a+2 are 256 B = 1 unit apart."""
    o = 8 * s
    return ["ds_read2st64_b32 v[%d:%d], %%[la0] offset0:%d offset1:%d" % (fa(st, 0), fa(st, 0) + 1, o, o + 1) if False else None,
            ]


def gen():
    out = []
    e = out.append
    # blocks (0, 2) come back as a register PAIR from one ds_read2: place A fragments so that pairs are adjacent registers:
    # register order inside a set: [a0, a2, a1, a3, b0, b2, b1, b3]
    ra = {0: 0, 2: 1, 1: 2, 3: 3}
    def FA(st, a): return F0 + 8 * st + ra[a]
    def FB(st, b): return F0 + 8 * st + 4 + ra[b]
    def reads(st, s):
        o = 8 * s
        return ["ds_read2st64_b32 v[%d:%d], %%[la0] offset0:%d offset1:%d" % (FA(st, 0), FA(st, 0) + 1, o, o + 1),
                "ds_read2st64_b32 v[%d:%d], %%[la1] offset0:%d offset1:%d" % (FA(st, 1), FA(st, 1) + 1, o, o + 1),
                "ds_read2st64_b32 v[%d:%d], %%[lb0] offset0:%d offset1:%d" % (FB(st, 0), FB(st, 0) + 1, o, o + 1),
                "ds_read2st64_b32 v[%d:%d], %%[lb1] offset0:%d offset1:%d" % (FB(st, 1), FB(st, 1) + 1, o, o + 1)]
    # chunk entry
    e("s_waitcnt vmcnt(16)")
    e("s_barrier")
    for r in reads(0, 0):
        e(r)
    e("s_waitcnt lgkmcnt(0)")
    pieces = [("ga", "oa%d" % i, i * 4096) for i in range(4)] + [("gb", "ob%d" % i, A_BYTES + i * 4096) for i in range(4)]
    for s in range(8):
        st = s & 1
        fill = {}
        if s + 1 < 8:
            rs = reads(st ^ 1, s + 1)
            for j in range(4):
                fill.setdefault(j, []).append(rs[j])
        # the four bias column sums in ONE gap (round 6): next to the f32-input MFMA a gap with n VALU instructions costs 9.6 + 4 n cycles
        # (tools/ubench/f32_gap_cost.hip) -- four gaps of one cost 54 cycles per 16 MFMAs, one gap of four 26
        for a in range(4):
            fill.setdefault(4, []).append("v_add_f32 %%[bs%d], %%[bs%d], v%d" % (a, a, FA(st, a)))
        g, o, lds = pieces[s]
        fill.setdefault(9, []).append("s_add_u32 m0, %%[md], %d" % lds)
        fill.setdefault(10, []).append("global_load_lds_dwordx4 %%[%s], %%[%s]" % (o, g))       # one MFMA between m0 and its use
        j = 0
        for a in range(4):
            for b in range(4):
                e("v_mfma_f32_32x32x2_f32 a[%d:%d], v%d, v%d, a[%d:%d]" % (acc(a, b), acc(a, b) + 15, FA(st, a), FB(st, b), acc(a, b), acc(a, b) + 15))
                for f in fill.get(j, []):
                    e(f)
                j += 1
        if s + 1 < 8:
            e("s_waitcnt lgkmcnt(0)")
    return out


def main():
    body = gen()
    n_mfma = sum(1 for l in body if l.startswith("v_mfma"))
    with open(sys.argv[1], "w") as f:
        f.write("// GENERATED by tools/gen_dw_f32.py -- do not edit.  %d MFMAs, %d other instructions per chunk.\n" % (n_mfma, len(body) - n_mfma))
        f.write("#define SN_DWF32_CHUNK_ASM \\\n")
        for l in body:
            f.write('  "%s\\n\\t" \\\n' % l)
        f.write('  ""\n')
        f.write("#define SN_DWF32_ZERO_ASM \\\n")
        for i in range(256):
            f.write('  "v_accvgpr_write_b32 a%d, 0\\n\\t" \\\n' % i)
        f.write('  ""\n')
        f.write("#define SN_DWF32_FRAG_CLOBBERS " + ", ".join('"v%d"' % r for r in range(F0, F0 + 16)) + "\n")
        f.write("#define SN_DWF32_AGPR_CLOBBERS " + ", ".join('"a%d"' % r for r in range(256)) + "\n")
    print("dw f32 chunk: %d MFMAs, %d other" % (n_mfma, len(body) - n_mfma))


if __name__ == "__main__":
    main()
