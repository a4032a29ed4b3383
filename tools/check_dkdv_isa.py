#!/usr/bin/env python3
"""Checks the invariants the hand-placed dK/dV kernels rely on but hipcc cannot know (their asm statements own literal
registers).  Inside the kernel's ISA, OUTSIDE the ASMSTART / ASMEND brackets:

 (1) no v_accvgpr_* names an accumulator register below `amin` (a[0:191] hold the accumulators and the K / V fragments);
 (2) no scratch access exists (a spill would go through registers hipcc believes free);
 (3) the VGPRs a slice body PREFETCHES for the next one stay untouched on the way there.  Every slice body ends by reading the
     next slice's row fragments / row constants into literal VGPRs (v[128:175] at head_dim 64, v[96:175] at 128) and the HOT
     variants of the next iteration use them without loading.  hipcc only sees those registers in the statements' clobber lists,
     so nothing stops it from using them for its own values in the code BETWEEN two statements (loop increment, vmcnt wait,
     barrier, LDS-DMA staging of slice it + 4, the branch ladder).  This walks the kernel's control-flow graph: the blocks that
     are reachable from the end of a body AND reach the start of a HOT body -- without passing through another body or through
     the masked (tail) branch, which sets hot = false and may use any register -- must not name a protected register.
     Which registers are protected is read off the ISA itself: the VGPRs >= 64 a HOT body reads before it writes them (hipcc's
     own operands live below v64: every body clobbers v[64:255]); the caller states the range it expects and a mismatch fails.

usage: python tools/check_dkdv_isa.py <attention.s> <kernel name substring> [amin=192] [protected=LO:HI]
Importable: `check(isa_text, kernel_substring, amin=192, protected=(lo, hi))` -> report dict with `ok` and `problems`.
"""
import re
import sys

BODY_MFMAS = 48          # a slice body holds 64 MFMAs; the masked path's separate statements hold <= 16 each


def _regs(text, kind="v"):
    """Register numbers of one file (`v` or `a`) named in an instruction's operand text."""
    out = {int(x) for x in re.findall(r"\b%s(\d+)\b" % kind, text)}
    for a, b in re.findall(r"\b%s\[(\d+):(\d+)\]" % kind, text):
        out |= set(range(int(a), int(b) + 1))
    return out


def _split_dst_src(ins):
    """(destination operand text, source operand text) of one instruction; stores / DMA / compares-to-vcc have no VGPR
    destination."""
    ins = ins.split(";")[0].strip()
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return "", ""
    op, rest = parts
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", rest)]
    if op.startswith(("ds_write", "ds_store", "global_store", "buffer_store", "global_load_lds", "s_", "v_cmp", "global_atomic")):
        return "", rest
    return ops[0], ", ".join(ops[1:])


def _read_before_written(asm_lines, lo=64):
    """VGPRs >= lo an asm statement reads before it writes them: values it expects from outside."""
    written, need = set(), set()
    for ln in asm_lines:
        dst, src = _split_dst_src(ln)
        s = _regs(src)
        # an MFMA / FMA whose destination is also its accumulator input reads it as well (it is in `src` then)
        need |= {r for r in s if r >= lo and r not in written}
        written |= _regs(dst)
    return need


def parse_kernel(isa, name):
    m = re.search(r"^(\S*%s\S*):\s*;\s*@" % re.escape(name), isa, flags=re.M)
    if m is None:
        raise KeyError(f"no kernel matching {name!r}")
    body = isa[m.end():isa.index(".Lfunc_end", m.end())]
    meta = isa[isa.index(".amdhsa_kernel " + m.group(1)):]
    meta = meta[:meta.index(".end_amdhsa_kernel")]
    return m.group(1), body, meta


def build_nodes(body):
    """Nodes of the control-flow graph: basic blocks, cut again at every slice body (a body is a node of its own).
    node = dict(label, kind in {'code', 'body'}, ins [text], asm [[lines]], succ [node index])."""
    lines = body.splitlines()
    nodes, label_at = [], {}

    def new(label, kind="code"):
        nodes.append(dict(label=label, kind=kind, ins=[], asm=[], succ=[], fall=True))
        return nodes[-1]
    cur = new("entry")
    label_at["entry"] = 0
    i = 0
    while i < len(lines):
        ln = lines[i]
        ml = re.match(r"^(\.LBB\d+_\d+):", ln)
        if ml:
            cur = new(ml.group(1))
            label_at[ml.group(1)] = len(nodes) - 1
        elif ";;#ASMSTART" in ln:
            j = i + 1
            while ";;#ASMEND" not in lines[j]:
                j += 1
            stmt = [x.strip() for x in lines[i + 1:j] if x.strip()]
            if sum(1 for x in stmt if x.startswith("v_mfma")) >= BODY_MFMAS:
                b = new(cur["label"] + "+body", "body")
                b["asm"].append(stmt)
                cur = new(cur["label"] + "+after")
            else:
                cur["asm"].append(stmt)
            i = j
        else:
            t = ln.strip()
            if t and not t.startswith((";", ".")):
                cur["ins"].append(t)
                mb = re.match(r"s_(cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", t)
                if mb:
                    cur["succ"].append(mb.group(2))
                    if mb.group(1) == "branch":
                        cur["fall"] = False
                        cur = new(cur["label"] + "+dead")          # anything behind an unconditional branch up to the next label
                    else:
                        cur = new(cur["label"] + "+")              # a conditional branch ends a basic block too
                elif t.startswith("s_endpgm"):
                    cur["fall"] = False
                    cur = new(cur["label"] + "+dead")
        i += 1
    for k, n in enumerate(nodes):
        succ = [label_at[s] for s in n["succ"]]
        if n["fall"] and k + 1 < len(nodes):
            succ.append(k + 1)
        n["succ"] = sorted(set(succ))
    return nodes


def check(isa, name, amin=192, protected=None):
    kname, body, meta = parse_kernel(isa, name)
    get = lambda k: re.search(k + r"\s+(\d+)", meta).group(1)
    problems = []
    outside = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", body, flags=re.S)
    bad_acc = []
    for ln in outside.splitlines():
        if "accvgpr" in ln and any(r < amin for r in _regs(ln, "a")):
            bad_acc.append(ln.strip())
    if bad_acc:
        problems.append(f"v_accvgpr_* outside asm names a< {amin}: {bad_acc[:4]}")
    scr = [ln.strip() for ln in body.splitlines() if re.search(r"\bscratch_|buffer_(load|store)", ln)]
    if scr or get(".amdhsa_private_segment_fixed_size") != "0":
        problems.append(f"scratch: private segment {get('.amdhsa_private_segment_fixed_size')} bytes, {len(scr)} accesses {scr[:3]}")

    nodes = build_nodes(body)
    bodies = [k for k, n in enumerate(nodes) if n["kind"] == "body"]
    need = {k: _read_before_written(nodes[k]["asm"][0]) for k in bodies}
    hot = [k for k in bodies if need[k]]
    prot = set().union(*[need[k] for k in hot]) if hot else set()
    if protected is not None:
        want = set(range(protected[0], protected[1] + 1))
        if prot != want:
            problems.append(f"HOT bodies expect v{sorted(prot)[:1]}..v{sorted(prot)[-1:]} ({len(prot)} registers) from the previous "
                            f"body's prefetch, the caller states v[{protected[0]}:{protected[1]}]: the kernel and its checker disagree")
    if len(bodies) < 2 or not hot or len(hot) == len(bodies):
        problems.append(f"expected prefetching bodies of both kinds, found {len(bodies)} bodies, {len(hot)} of them HOT")
    # the masked (tail) branch: code blocks that hold MFMA statements which are not bodies
    masked = {k for k, n in enumerate(nodes) if n["kind"] == "code" and any(x.startswith("v_mfma") for st in n["asm"] for x in st)}
    stop = set(bodies) | masked
    pred = {k: [] for k in range(len(nodes))}
    for k, n in enumerate(nodes):
        for s_ in n["succ"]:
            pred[s_].append(k)

    def reach(starts, edges):
        seen, todo = set(), [s_ for s_ in starts if s_ not in stop]
        while todo:
            k = todo.pop()
            if k in seen:
                continue
            seen.add(k)
            todo += [x for x in edges(k) if x not in stop and x not in seen]
        return seen
    fwd = reach([s_ for b in bodies for s_ in nodes[b]["succ"]], lambda k: nodes[k]["succ"])
    bwd = reach([p for b in hot for p in pred[b]], lambda k: pred[k])
    between = sorted(fwd & bwd)
    touched = []
    for k in between:
        for t in nodes[k]["ins"]:
            r = _regs(t) & prot
            if r:
                touched.append((nodes[k]["label"], t, sorted(r)[:4]))
    if touched:
        problems.append(f"{len(touched)} instruction(s) between a prefetching body and a HOT body name the prefetched registers: "
                        + "; ".join(f"{lab}: {t}" for lab, t, _ in touched[:4]))
    hi_all = sorted({r for ln in outside.splitlines() for r in _regs(ln) if r >= 64})
    return dict(kernel=kname, ok=not problems, problems=problems, bodies=len(bodies), hot_bodies=len(hot),
                protected=(min(prot), max(prot)) if prot else None, masked_blocks=len(masked), blocks_between=len(between),
                instructions_between=sum(len(nodes[k]["ins"]) for k in between), vgprs_ge64_outside_asm=len(hi_all),
                next_free_vgpr=int(get(".amdhsa_next_free_vgpr")), accum_offset=int(get(".amdhsa_accum_offset")))


if __name__ == "__main__":
    prot = None
    for a in sys.argv[4:]:
        if a.startswith("protected="):
            lo, hi = a.split("=", 1)[1].split(":")
            prot = (int(lo), int(hi))
    rep = check(open(sys.argv[1]).read(), sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 192, prot)
    for k, v in rep.items():
        print(f"{k}: {v}")
    sys.exit(0 if rep["ok"] else 1)
