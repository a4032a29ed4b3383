#!/usr/bin/env python3
"""Checks the two invariants the hand-placed dK/dV kernels rely on but hipcc cannot know (their asm statements own literal
registers): inside the kernel's ISA, OUTSIDE the ASMSTART / ASMEND brackets, (1) no v_accvgpr_* names an accumulator register
below `amin` (a[0:191] hold the accumulators and the K / V fragments), (2) no scratch access exists (a spill would go through
registers hipcc believes free), and it lists the VGPRs >= 64 hipcc uses there (legal between two statements, but not for a value
that lives across the HOT body).
usage: python tools/check_dkdv_isa.py <attention.s> <kernel name substring> [amin=192]"""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
amin = int(sys.argv[3]) if len(sys.argv) > 3 else 192
m = re.search(r"^(\S*%s\S*):\s*;\s*@" % re.escape(name), s, flags=re.M)
start = m.end()
body = s[start:s.index(".Lfunc_end", start)]
outside = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", body, flags=re.S)
bad = []
for ln in outside.splitlines():
    if "accvgpr" in ln:
        regs = [int(x) for x in re.findall(r"\ba(\d+)\b", ln)] + [int(b) for _, b in re.findall(r"a\[(\d+):(\d+)\]", ln)]
        if any(r < amin for r in regs):
            bad.append(ln.strip())
hi = sorted({int(x) for ln in outside.splitlines() for x in re.findall(r"\bv(\d+)\b", ln) if int(x) >= 64} |
            {int(b) for ln in outside.splitlines() for _, b in re.findall(r"v\[(\d+):(\d+)\]", ln) if int(b) >= 64})
scr = [ln.strip() for ln in body.splitlines() if re.search(r"\bscratch_|buffer_(load|store)", ln)]
meta = s[s.index(".amdhsa_kernel " + m.group(1)):]
meta = meta[:meta.index(".end_amdhsa_kernel")]
get = lambda k: re.search(k + r"\s+(\d+)", meta).group(1)
print(f"{m.group(1)[:60]}...: {body.count(chr(10))} lines; next_free_vgpr {get('.amdhsa_next_free_vgpr')}, accum_offset "
      f"{get('.amdhsa_accum_offset')}, private_segment {get('.amdhsa_private_segment_fixed_size')}")
print(f"v_accvgpr_* outside asm naming a< {amin}: {len(bad)} {bad[:4]}")
print(f"accvgpr outside asm (any): {len([l for l in outside.splitlines() if 'accvgpr' in l])}")
print(f"scratch / buffer accesses: {len(scr)} {scr[:3]}")
print(f"VGPRs >= 64 named outside asm: {len(hi)} {hi[:24]}")
sys.exit(1 if bad or scr or get('.amdhsa_private_segment_fixed_size') != '0' else 0)
