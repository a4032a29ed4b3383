#!/usr/bin/env python3
"""gfx950 transcendental forwarding hazard, checked in a kernel library's ISA: a VALU instruction that reads the result of a
v_exp_f32 / v_log_f32 / v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32 must not be the very next instruction (one
wait state).  hipcc inserts the s_nop itself for instructions it can see; it cannot for the text of an asm statement -- a
hand-placed stream, or a one-instruction helper that happens to be scheduled right behind the v_exp_f32 feeding it (how
fa_fwd128_kernel came to pack un-exponentiated scores in round 3).
usage: python tools/check_trans_hazard.py file.s [...]   (hipcc -S --cuda-device-only output); exit status 1 on a finding"""
import re
import sys

TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_(f32|f16|bf16)\S*\s+v(\d+)")
bad = 0
for path in sys.argv[1:]:
    kernel = "?"
    prev = None                                     # (line number, destination register) of a transcendental just issued
    for n, raw in enumerate(open(path, errors="ignore"), 1):
        t = raw.strip()
        m = re.match(r"^(\S+):\s*; @", t)
        if m:
            kernel = m.group(1)
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue                                # comments, directives, labels: not instructions
        if prev is not None and t.startswith("v_"):
            ops = t.split(None, 1)[1] if " " in t else ""
            srcs = ops.split(",")[1:]               # everything after the destination
            regs = set()
            for s in srcs:
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", s):
                    regs |= set(range(int(a), int(b) + 1))
                regs |= {int(x) for x in re.findall(r"\bv(\d+)\b", s)}
            if prev[1] in regs:
                print(f"{path}:{n}: {kernel[:60]}: `{t[:70]}` reads v{prev[1]} written by the transcendental on line {prev[0]}")
                bad += 1
        m = TRANS.match(t)
        prev = (n, int(m.group(3))) if m else None
print(f"{bad} transcendental -> VALU adjacencies")
sys.exit(1 if bad else 0)
