#!/usr/bin/env python3
"""Checks what fa_fwd128w_kernel / fa_fwd64w_kernel (attention forward, one wave per SIMD) rely on but hipcc cannot know: its asm statements own
LITERAL registers -- v[64:227] from RPO_FW_INIT on and the accumulator file from RPO_FW_INIT_ACC on -- which hipcc only sees in the
statements' clobber lists.  In the kernel's ISA, OUTSIDE the ASMSTART / ASMEND brackets:

 (1) no instruction behind RPO_FW_INIT names a VGPR in 64..227 (scores, V^T and P^T fragments, the scale, the bf16 ones live there
     from one statement to the next; the epilogue's own values need none of them either, so the rule covers the whole rest);
 (2) no instruction behind RPO_FW_INIT_ACC names an accumulator register (O^T, l, Q^T, the K fragments: a[0:239]; a spill of hipcc's
     into a[240:255] would show here as well and is refused: spills mean the prologue / epilogue grew too fat);
 (3) the kernel uses no scratch.

usage: python tools/check_fwd128w_isa.py <attention.s>
Importable: `check(isa_text, kernel_name)` -> report dict with `ok`, `problems` and the counts the test pins.
"""
import re
import sys

INIT_MARK = "v_mov_b32 v224, 0x3f803f80"          # first instruction of RPO_FW_INIT: the bf16 ones
ACC_MARK = "v_accvgpr_write_b32 a0, 0"            # first instruction of RPO_FW_INIT_ACC / RPO_DQ_INIT_ACC
LO, HI = 64, 227
# kernel -> (first instruction of its INIT statement, owned VGPR range): the forwards, and the head_dim-64 dQ kernel of the same make
KERNELS = {"fa_fwd128w_kernel": (INIT_MARK, 64, 227), "fa_fwd64w_kernel": (INIT_MARK, 64, 227),
           "fa_bwd_dq64w_kernel": ("v_mov_b32 v192, ", 64, 215)}
OPTIONAL = ("fa_fwd64w_kernel", "fa_bwd_dq64w_kernel")     # only in a `make ONEWAVE64=1` build (rpo_build_flags())


def kernel_body(isa, name="fa_fwd128w_kernel"):
    lines = isa.split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % name, l)]
    if not start:
        raise ValueError("no kernel %s in the ISA" % name)
    end = next(i for i in range(start[0], len(lines)) if "s_endpgm" in lines[i])
    tail = "\n".join(lines[end:end + 400])
    return lines[start[0]:end], tail


def _names_owned_vgpr(t, lo=LO, hi=HI):
    for m in re.finditer(r"\bv(\d+)\b", t):
        if lo <= int(m.group(1)) <= hi:
            return True
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", t):
        if int(m.group(2)) >= lo and int(m.group(1)) <= hi:
            return True
    return False


def check(isa, name="fa_fwd128w_kernel"):
    body, tail = kernel_body(isa, name)
    init_mark, lo, hi = KERNELS.get(name, (INIT_MARK, LO, HI))
    init_at = next((i for i, l in enumerate(body) if init_mark in l), None)
    acc_at = next((i for i, l in enumerate(body) if ACC_MARK in l), None)
    problems = []
    if init_at is None or acc_at is None or acc_at > init_at:
        return {"ok": False, "problems": ["RPO_FW_INIT_ACC / RPO_FW_INIT not found in this order"], "checked": 0, "statements": 0}
    inasm, checked, statements = False, 0, 0
    for i, l in enumerate(body):
        if "ASMSTART" in l:
            inasm = True
            statements += 1
            continue
        if "ASMEND" in l:
            inasm = False
            continue
        if inasm or i < acc_at:
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        checked += 1
        if re.search(r"\ba\d+\b|\ba\[\d+", t):
            problems.append("line %d names an accumulator register outside the statements: %s" % (i, t))
        if i > init_at and _names_owned_vgpr(t, lo, hi):
            problems.append("line %d names a VGPR of v[%d:%d] outside the statements: %s" % (i, lo, hi, t))
        if t.startswith("scratch_"):
            problems.append("line %d: scratch access: %s" % (i, t))
    m = re.search(r"ScratchSize:\s*(\d+)", tail)
    if not m or int(m.group(1)) != 0:
        problems.append("ScratchSize is not 0: %s" % (m.group(0) if m else "not found"))
    return {"ok": not problems, "problems": problems, "checked": checked, "statements": statements}


if __name__ == "__main__":
    isa_ = open(sys.argv[1]).read()
    rc = 0
    for kern in KERNELS:
        try:
            rep = check(isa_, kern)
        except ValueError:
            if kern in OPTIONAL:
                print("%s: not in this build (make ONEWAVE64=1)" % kern)
                continue
            raise
        print("%s: %d instructions of hipcc's behind RPO_FW_INIT_ACC checked, %d asm statements: %s"
              % (kern, rep["checked"], rep["statements"], "ok" if rep["ok"] else "FAILED"))
        for p in rep["problems"][:20]:
            print("  ", p)
        rc |= 0 if rep["ok"] else 1
    sys.exit(rc)
