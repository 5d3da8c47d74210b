#!/usr/bin/env python3
"""bodies of tools/ubench/store_issue.hip: one loop iteration = 32 back-to-back bf16 MFMAs (a 'slab') with row stores dealt into the gaps.
   BODY_<name> macros; registers: a[0:63] accumulators, v[8:11] / v[12:15] operands, v[16:19] store data, v20 byte offset"""
import sys
def body(n_st, kind="store", rot=False, solo=False):
    out = []
    for k in range(4):
        out += ["s_cmp_eq_u32 %%[wave], %d" % k, "s_cselect_b32 s%d, -1, 0" % (40 + k)]
    every = 32 // n_st if n_st else 0
    for j in range(32):
        out.append("v_mfma_f32_32x32x16_bf16 a[%d:%d], v[8:11], v[12:15], a[%d:%d]" % (16 * (j % 4), 16 * (j % 4) + 15, 16 * (j % 4), 16 * (j % 4) + 15))
        if rot:                                   # a store slot in EVERY gap, live for wave (j % 4) only: the waves' stores never coincide
            out.append("s_mov_b32 exec_lo, s%d" % (40 + j % 4)); out.append("s_mov_b32 exec_hi, s%d" % (40 + j % 4))
            out.append("v_add_u32 v20, %[inc], v20")
            out.append("global_store_dwordx4 v20, v[16:19], %[base] nt")
            out.append("s_mov_b64 exec, -1")
        elif n_st and j % every == every - 1:
            if solo:
                out.append("s_mov_b32 exec_lo, s40"); out.append("s_mov_b32 exec_hi, s40")
            out.append("v_add_u32 v20, %[inc], v20")
            if kind == "store":
                out.append("global_store_dwordx4 v20, v[16:19], %[base] nt")
            else:
                out.append("global_load_dwordx4 v[24:27], v20, %[base] nt")
            if solo:
                out.append("s_mov_b64 exec, -1")
    out.append("v_and_b32 v20, %[wrap], v20")
    if kind == "load":
        out.append("s_waitcnt vmcnt(0)")
    return out
with open(sys.argv[1], "w") as f:
    for name, b in [("NONE", body(0)), ("ST8", body(8)), ("ST4", body(4)), ("ST16", body(16)), ("ROT", body(8, rot=True)),
                    ("SOLO", body(8, solo=True)), ("LD8", body(8, kind="load"))]:
        f.write("#define BODY_%s \\\n" % name)
        for l in b:
            f.write('  "%s\\n\\t" \\\n' % l)
        f.write('  ""\n')
