#!/usr/bin/env python3
"""Build check for sn_mlp_fwd.hip / sn_mlp_fwd_bf16.hip / sn_mlp_bwd_bf16.hip (run by sinnerf_amd/csrc/Makefile on the hipcc -S output).

These kernels manage the AGPR file by hand and emit their MFMAs as inline asm, so a few things the compiler normally
guarantees are checked on the generated code instead:
  1. the compiler allocated no AGPR itself (every AGPR reference sits inside ASMSTART/ASMEND) and spilled nothing;
  2. no VALU instruction writes a register an MFMA reads within the next 2 wait states (VALU write -> MFMA read hazard);
  3. no VALU / LDS / memory instruction reads an MFMA result before two further MFMAs (or 12 other instructions) have issued
     (MFMA write -> VALU read hazard: 11 wait states for the 8-pass bf16 MFMA, 18 for the 16-pass fp32 one);
  4. the destination of a global load issued from inline asm is not referenced before an s_waitcnt vmcnt(N) that COVERS it:
     vmcnt retires in issue order, so vmcnt(N) completes a load only if at least N vector-memory instructions were issued
     behind it (counted on the generated code -- a compiler that moves one nontemporal store across a counted wait fails
     the build instead of producing stale masks);
  5. a vector-memory instruction inside inline asm does not use an SGPR address within 5 wait states of the SALU instruction
     that wrote it (hipcc pads this hazard for its own instructions only);
  6. a v_permlane32_swap_b32 does not read a register a VALU instruction wrote within the previous 2 wait states (hipcc inserts
     s_nop 1 for its own code).
  7. an LDS-DMA load (global_load_lds_*) does not issue in the wait state right behind the SALU write of m0 (its LDS address): the
     kernels issue their DMA pieces from inline asm, some with the m0 write and the load in separate statements.
usage: check_agpr.py file.s"""
import re, sys

def vregs(tok):
    out = set()
    for m in re.finditer(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]', tok):
        if m.group(1): out.add(('v', int(m.group(1))))
        else: out |= {('v', r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r'\ba(\d+)\b|\ba\[(\w+)(?::(\w+))?\]', tok):
        if m.group(1): out.add(('a', int(m.group(1))))
        else:
            lo = int(m.group(2), 0); hi = int(m.group(3), 0) if m.group(3) else lo
            out |= {('a', r) for r in range(lo, hi + 1)}
    return out

kern = None; ina = False; ins = []; bad_agpr = []; spills = 0; pend_ld = {}; bad_async = []; VMEM = ('global_', 'buffer_', 'flat_', 'scratch_')
sgpr_written = {}; clock = 0; bad_sgpr = []; m0_written = -100; bad_m0 = []

def sregs(tok):
    out = set()
    for m in re.finditer(r'\bs(\d+)\b|\bs\[(\d+):(\d+)\]', tok):
        if m.group(1): out.add(int(m.group(1)))
        else: out |= set(range(int(m.group(2)), int(m.group(3)) + 1))
    return out

for ln, l in enumerate(open(sys.argv[1]), 1):
    m = re.match(r'^(_Z\S*(?:mlp_(?:fwd_bf16|fwd_bf16x3|fwd_bf16_v3|fwd_bf16_t|fwd_f32|bwd_chain_bf16|bwd_chain_bf16x3|bwd_chain_bf16_t|bwd_chain_f32)_kernel|dw_f32_asm_kernel|dw_bf16_asm_kernel|dw_narrow_bf16_asm_kernel)\S*):', l)
    if m: kern = m.group(1); continue
    if kern is None: continue
    if re.match(r'^\s*s_endpgm', l): kern = None; continue
    if 'ASMSTART' in l: ina = True; continue
    if 'ASMEND' in l: ina = False; continue
    t = l.strip().split(';')[0].strip()
    if not t or t[0] == '.' or t.endswith(':'): continue
    if 'scratch_' in t: spills += 1
    m = re.match(r'^([a-z_0-9]+)\s*(.*)$', t)
    op, args = m.groups()
    parts = [a.strip() for a in args.split(',')] if args else []
    if not ina and any(k == 'a' for k, _ in vregs(args)): bad_agpr.append((ln, t))
    is_valu = op.startswith('v_') and not op.startswith('v_mfma')
    reads_regs = is_valu or op.startswith(('ds_write', 'ds_read', 'global_store', 'global_load', 'flat_', 'scratch_', 'buffer_'))
    if op.startswith(('v_', 'ds_read', 'global_load_dword', 'scratch_load')) and not op.startswith(('v_cmp', 'v_readfirstlane')):
        dst = vregs(parts[0]) if parts else set(); src = set().union(*[vregs(p) for p in parts[1:]]) if len(parts) > 1 else set()
        if op.startswith(('v_fmac', 'v_pk_fma', 'v_mfma')): src |= dst if not op.startswith('v_mfma') else set()
    else:
        dst = set(); src = set().union(*[vregs(p) for p in parts]) if parts else set()
    ins.append((ln, op, dst, src, is_valu, t, reads_regs))
    # 5. SALU write -> VMEM address inside inline asm
    if ina and op.startswith(('global_', 'buffer_', 'flat_')) and parts:
        for r in sregs(' '.join(parts[1:])):
            if r in sgpr_written and clock - sgpr_written[r] < 6: bad_sgpr.append((ln, t))
    if op.startswith('s_') and not op.startswith(('s_nop', 's_waitcnt', 's_barrier', 's_cmp', 's_cbranch', 's_branch', 's_endpgm')) and parts:
        for r in sregs(parts[0]): sgpr_written[r] = clock
        if parts[0] == 'm0': m0_written = clock
    # 7. SALU write of m0 -> LDS-DMA load
    if op.startswith('global_load_lds') and clock - m0_written < 2: bad_m0.append((ln, t))
    clock += (int(t.split()[1], 0) + 1) if op == 's_nop' else 8 if op.startswith('v_mfma') else 1
    # 4. destination registers of a global load issued from inline asm (the wait is hand-placed) are not touched by anything
    #    until an s_waitcnt vmcnt has issued
    if op == 's_waitcnt' and 'vmcnt' in args:
        n = int(re.search(r'vmcnt\((\d+)\)', args).group(1))
        pend_ld = {r: c for r, c in pend_ld.items() if c < n}          # c ops issued behind the load: retired iff c >= n
    elif pend_ld and (vregs(args) & set(pend_ld)): bad_async.append((ln, t))
    if op.startswith(VMEM):
        for r in pend_ld: pend_ld[r] += 1
    if ina and op.startswith('global_load_dword') and 'lds' not in op:
        for r in vregs(parts[0]): pend_ld[r] = 0

haz1 = []; haz2 = []
for i, (ln, op, dst, src, is_valu, t, _rr) in enumerate(ins):
    if op.startswith('v_permlane32_swap'):
        ws = 0; j = i - 1
        while j >= 0 and ws < 2:
            if ins[j][1] == 's_nop': ws += int(ins[j][5].split()[1], 0) + 1
            else:
                if ins[j][4] and (ins[j][2] & (src | dst)): haz1.append((ln, ins[j][5], t))
                ws += 8 if ins[j][1].startswith('v_mfma') else 1
            j -= 1
        continue
    if not op.startswith('v_mfma'): continue
    # leading s_nop N inside the same asm statement shows up as the previous instruction
    ws = 0; j = i - 1
    while j >= 0 and ws < 2:
        pop = ins[j][1]
        if pop == 's_nop': ws += int(ins[j][5].split()[1], 0) + 1
        else:
            if ins[j][4] and (ins[j][2] & src): haz1.append((ln, ins[j][5], t))
            ws += 8 if pop.startswith('v_mfma') else 1
        j -= 1
    nm = 0; other = 0; j = i + 1
    while j < len(ins) and nm < 2 and other < 12:
        if ins[j][1].startswith('v_mfma'): nm += 1
        else:
            if ins[j][6] and (ins[j][3] & dst): haz2.append((ln, t, ins[j][5]))
            other += (int(ins[j][5].split()[1], 0) + 1) if ins[j][1] == 's_nop' else 1
        j += 1

print(f"compiler-allocated AGPR references: {len(bad_agpr)}; scratch instructions: {spills}; "
      f"VALU->MFMA read hazards: {len(haz1)}; MFMA->VALU read hazards: {len(haz2)}; early uses of asm loads: {len(bad_async)}; SALU->VMEM address hazards in asm: {len(bad_sgpr)}; m0->LDS-DMA hazards: {len(bad_m0)}")
for b in bad_m0[:5]: print("  m0    line %d: %s" % b)
for b in bad_sgpr[:5]: print("  sgpr  line %d: %s" % b)
for b in bad_async[:5]: print("  async line %d: %s" % b)
for b in bad_agpr[:5]: print("  agpr  line %d: %s" % b)
for b in haz1[:5]: print("  haz1  line %d: %s  ->  %s" % b)
for b in haz2[:5]: print("  haz2  line %d: %s  ->  %s" % b)
sys.exit(1 if bad_agpr or spills or haz1 or haz2 or bad_async or bad_sgpr or bad_m0 else 0)
