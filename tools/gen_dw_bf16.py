#!/usr/bin/env python3
"""Generator of the hand-scheduled inner loop of the bf16-state 256x256 weight-gradient kernel (sinnerf_amd/csrc/sn_dw_bf16.hip):
dW[m, n] += sum over the 16 points of a staged chunk of G[p, m] X[p, n] on v_mfma_f32_32x32x16_bf16 -- one MFMA k-step per
chunk and 32x32 block, operands gathered from the row-major bf16 tiles by the hardware transpose read.

The compiler-scheduled loop of sn_dw.hip (MODE 2) spends ~2300 cycles per chunk for 512 cycles of MFMAs (16 ds_read_b64_tr_b16 +
their waits, 4 DMA pieces, 16 dot products, a sync point every second chunk): 4.4 TB/s where the same box copies at 5.2-6.2.
Here one statement handles a PAIR of chunks (c, c+1), laid out by hand:

  entry    s_waitcnt vmcnt(16)   chunks c, c+1 have landed (the four younger ones, 4 DMA instructions per thread each, in flight)
           s_barrier             ... for every wave; every wave has left the slots of chunks c-2, c-1
           8 address adds + 16 transpose reads of chunk c -> fragment set 0, s_waitcnt lgkmcnt(0)
  half 0   16 MFMAs on set 0;  behind MFMA j: the reads of chunk c+1 -> set 1 (two per MFMA, all issued by MFMA 7), one bias
           dot product, the four DMA pieces of chunk c+6 (into the slot of chunk c-2)
           s_waitcnt lgkmcnt(0)
  half 1   16 MFMAs on set 1;  bias dot products, the four DMA pieces of chunk c+7 (slot of chunk c-1)

An odd last chunk of a K-range gets a statement of its own (SN_DWBF16_TAIL_ASM: no staging).  Nothing but the accumulators (the whole AGPR file: block (a, b) of the wave's 4 x 4 blocks at a[16 (4a + b) : +15]) and the
four bias sums lives across statements.  Fragment sets: v[F0 : F0+63] (set s: A blocks at F0 + 32 s + 4 a, B blocks 16 further),
address temporaries v[T0 : T0+7].

Operands of the statement (see the kernel): bs0..3 (+v) column sums; ta0..3 tb0..3 (v) this lane's transpose-read byte offsets
inside a slot (B offsets include the A tile); one (v) = 0x3f803f80; oa0 oa1 ob0 ob1 (v) per-thread global byte offsets of the DMA
pieces; sl0 sl1 (s) LDS byte offsets of the slots of chunks c, c+1; ga0 gb0 / ga1 gb1 (s, 64 bit) global bases of the chunks
staged in half 0 / 1; md0 md1 (s) their LDS destinations + wave * 1024.

usage: gen_dw_bf16.py out.inc
"""
import sys

F0 = 176                       # v[176:239]: two fragment sets of 8 x 4 registers
T0 = 240                       # v[240:247]: read addresses of the chunk being gathered
A_BYTES = 16 * 256 * 2         # A tile of a chunk (16 points x 256 features bf16); the B tile follows it
ROW4 = 4 * 256 * 2             # second transpose read of a block: 4 point rows further


def fa(st, a): return F0 + 32 * st + 4 * a
def fb(st, b): return F0 + 32 * st + 16 + 4 * b
def acc(a, b): return 16 * (4 * a + b)


def addr_adds(sl):
    return ["v_add_u32 v%d, %%[%s], %%[ta%d]" % (T0 + i, sl, i) for i in range(4)] + \
           ["v_add_u32 v%d, %%[%s], %%[tb%d]" % (T0 + 4 + i, sl, i) for i in range(4)]


def reads(st):
    out = []
    for i in range(4):
        out.append("ds_read_b64_tr_b16 v[%d:%d], v%d" % (fa(st, i), fa(st, i) + 1, T0 + i))
        out.append("ds_read_b64_tr_b16 v[%d:%d], v%d offset:%d" % (fa(st, i) + 2, fa(st, i) + 3, T0 + i, ROW4))
    for i in range(4):
        out.append("ds_read_b64_tr_b16 v[%d:%d], v%d" % (fb(st, i), fb(st, i) + 1, T0 + 4 + i))
        out.append("ds_read_b64_tr_b16 v[%d:%d], v%d offset:%d" % (fb(st, i) + 2, fb(st, i) + 3, T0 + 4 + i, ROW4))
    return out


def half(st, next_reads, g):
    """16 MFMAs on set st; fill[j] = what is issued behind MFMA j."""
    fill = {}
    if next_reads:
        rs = reads(st ^ 1)
        for j in range(8):
            fill.setdefault(j, []).extend(rs[2 * j:2 * j + 2])
    # bias column sums: the four dwords of every A block (8 points of this lane's feature), fp32 accumulation
    dots = ["v_dot2c_f32_bf16 %%[bs%d], v%d, %%[one]" % (a, fa(st, a) + w) for a in range(4) for w in range(4)]
    for j in range(16):
        fill.setdefault(j, []).append(dots[j])
    # DMA pieces of the chunk staged in this half: A it = 0, 1 then B it = 0, 1 (m0 one MFMA ahead of its use)
    pieces = [] if g is None else [("ga%d" % g, "oa0", 0), ("ga%d" % g, "oa1", 4096), ("gb%d" % g, "ob0", A_BYTES), ("gb%d" % g, "ob1", A_BYTES + 4096)]
    for k, (base, off, lds) in enumerate(pieces):
        fill.setdefault(8 + 2 * k, []).append("s_add_u32 m0, %%[md%d], %d" % (g, lds))
        fill.setdefault(9 + 2 * k, []).append("global_load_lds_dwordx4 %%[%s], %%[%s] nt" % (off, base))
    out = []
    j = 0
    for a in range(4):
        for b in range(4):
            out.append("v_mfma_f32_32x32x16_bf16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" % (
                acc(a, b), acc(a, b) + 15, fa(st, a), fa(st, a) + 3, fb(st, b), fb(st, b) + 3, acc(a, b), acc(a, b) + 15))
            out.extend(fill.get(j, []))
            j += 1
    return out


def gen():
    out = ["s_waitcnt vmcnt(16)", "s_barrier"]
    out += addr_adds("sl0") + reads(0)
    out += ["s_waitcnt lgkmcnt(0)"]
    out += addr_adds("sl1")                      # (behind the wait: the address registers are reused)
    out += half(0, True, 0)
    out += ["s_waitcnt lgkmcnt(0)"]
    out += half(1, False, 1)
    return out


def gen_tail():
    """the odd last chunk of a K-range: everything has landed, nothing is staged"""
    return ["s_waitcnt vmcnt(0)", "s_barrier"] + addr_adds("sl0") + reads(0) + ["s_waitcnt lgkmcnt(0)"] + half(0, False, None)


def main():
    body = gen()
    n_mfma = sum(1 for l in body if l.startswith("v_mfma"))
    with open(sys.argv[1], "w") as f:
        f.write("// GENERATED by tools/gen_dw_bf16.py -- do not edit.  %d MFMAs, %d other instructions per chunk pair.\n" % (n_mfma, len(body) - n_mfma))
        f.write("#define SN_DWBF16_PAIR_ASM \\\n")
        for l in body:
            f.write('  "%s\\n\\t" \\\n' % l)
        f.write('  ""\n')
        f.write("#define SN_DWBF16_TAIL_ASM \\\n")
        for l in gen_tail():
            f.write('  "%s\\n\\t" \\\n' % l)
        f.write('  ""\n')
        f.write("#define SN_DWBF16_ZERO_ASM \\\n")
        for i in range(256):
            f.write('  "v_accvgpr_write_b32 a%d, 0\\n\\t" \\\n' % i)
        f.write('  ""\n')
        f.write("#define SN_DWBF16_VGPR_CLOBBERS " + ", ".join('"v%d"' % r for r in list(range(F0, F0 + 64)) + list(range(T0, T0 + 8))) + "\n")
        f.write("#define SN_DWBF16_AGPR_CLOBBERS " + ", ".join('"a%d"' % r for r in range(256)) + "\n")
    print("dw bf16 chunk pair: %d MFMAs, %d other" % (n_mfma, len(body) - n_mfma))


if __name__ == "__main__":
    main()
