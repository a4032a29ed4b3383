#!/usr/bin/env python3
"""Build-time check of the COUNTED `s_waitcnt vmcnt(N)` waits of the hand-scheduled kernels (attention forward one-wave-per-SIMD,
the 256 x 256 similarity kernel, the dK/dV kernels): they assume an exact number and order of vector-memory instructions in flight,
which is a property of hipcc's code generation that hipcc itself is never told about.

vmcnt counts a wave's outstanding vector-memory instructions (loads, LDS-DMA and, on gfx9-family parts, stores), which retire in
order: `vmcnt(N)` returns when at most the N YOUNGEST are outstanding.  A group G that the code is about to read has therefore
landed iff AT LEAST N vector-memory instructions were issued behind G's last one on every path to the wait.  More than N is merely
conservative; FEWER -- hipcc hoisting the rotary-table loads ahead of the Q pieces, merging or dropping a load -- is a silent race.

Two checks, both run by `make` after attention.o / infonce.o (rankpo_amd/csrc/Makefile: a failure fails the build):

 (1) SEMANTIC, the one-wave-per-SIMD attention forwards (fa_fwd128w_kernel, fa_fwd64w_kernel): every path of the control-flow graph
     from the kernel's entry to a counted wait of the PROLOGUE is enumerated (the prologue is acyclic).  On each, the first QP
     LDS-DMA instructions are the wave's Q pieces (issued FIRST: attention_fwdw_kernel.inc); the instructions issued behind the
     last of them must number at least N, and no plain (register) load may sit between or in front of the Q pieces (the rotary
     rows are meant to fly UNDER the DMAs, not to delay them).
 (2) SIGNATURE, every kernel listed in SIGNED: the sequence of vector-memory instructions and vmcnt waits in program order, basic
     block by basic block (hipcc's own and the asm statements' alike), hashed and compared with tools/isa_signatures.json -- the
     code generation the GPU suite validated.  Another hipcc, or an edit that moves a load, changes the hash: re-run the GPU suite
     (`pytest -m gpu tests/test_gpu_attention.py tests/test_gpu_kernels.py`), then `python tools/check_vmcnt_isa.py --update <.s files>`.

usage: python tools/check_vmcnt_isa.py [--update] attention.s [infonce.s ...]
Importable: `prologue_report(isa, kernel, qp)`, `signature(isa, kernel)`.
"""
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SIG_FILE = os.path.join(HERE, "isa_signatures.json")

# kernel name substring -> Q pieces per wave (64 rows x row bytes / 1 KiB pieces / 4 waves ... = what the source calls QP_)
PROLOGUE = {"fa_fwd128w_kernel": 16, "fa_fwd64w_kernel": 8}
SIGNED = {"attention.s": ["fa_fwd128w_kernel", "fa_fwd64w_kernel", "fa_bwd_dkdv4_kernel", "fa_bwd_dkdv128_kernel", "fa_bwd_dq64w_kernel"],
          "infonce.s": ["sim_tile256_kernel"]}

OPTIONAL = ("fa_fwd64w_kernel", "fa_bwd_dq64w_kernel")     # only in a `make ONEWAVE64=1` build (rpo_build_flags())

_VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")


def kernels(isa, sub):
    """[(mangled name, body text)] of every kernel whose name contains `sub` (template instances included)."""
    out = []
    for m in re.finditer(r"^(\S*%s\S*):\s*;\s*@" % re.escape(sub), isa, flags=re.M):
        end = isa.index(".Lfunc_end", m.end())
        out.append((m.group(1), isa[m.end():end]))
    if not out:
        raise KeyError("no kernel matching %r" % sub)
    return out


def _tok(ins):
    """Token of one instruction: 'L' LDS-DMA, 'G' load to registers, 'S' store / atomic, ('W', n) a vmcnt wait, None otherwise."""
    t = ins.split(";")[0].strip()
    if not t:
        return None
    op = t.split()[0]
    if _VMEM.match(op):
        if "_lds" in op or re.search(r"\blds\b", t):
            return "L"
        return "G" if "_load" in op else "S"
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", t)
        if m:
            return ("W", int(m.group(1)))
    return None


def blocks(body):
    """Basic blocks in program order: [dict(label, toks [...], succ [labels], fall bool)]; asm statements are inlined as they are."""
    out = [dict(label="entry", toks=[], succ=[], fall=True)]
    for ln in body.splitlines():
        ml = re.match(r"^(\.LBB\d+_\d+):", ln)
        if ml:
            out.append(dict(label=ml.group(1), toks=[], succ=[], fall=True))
            continue
        t = ln.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        cur = out[-1]
        mb = re.match(r"s_(cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", t)
        if mb:
            cur["succ"].append(mb.group(2))
            if mb.group(1) == "branch":
                cur["fall"] = False
            out.append(dict(label=cur["label"] + "+", toks=[], succ=[], fall=True))
            continue
        if t.startswith("s_endpgm"):
            cur["fall"] = False
            out.append(dict(label=cur["label"] + "+end", toks=[], succ=[], fall=True))
            continue
        k = _tok(t)
        if k is not None:
            cur["toks"].append(k)
        if "v_mov_b32 v224, 0x3f803f80" in t:
            cur["toks"].append("INIT")
    return out


def signature(isa, sub):
    """sha256 over the per-block token streams of every instance of the kernel (block labels replaced by their ordinal)."""
    h = hashlib.sha256()
    n_w = n_v = 0
    for name, body in kernels(isa, sub):
        bl = blocks(body)
        index = {b["label"]: i for i, b in enumerate(bl)}
        for i, b in enumerate(bl):
            toks = [t if isinstance(t, str) else "W%d" % t[1] for t in b["toks"]]
            n_w += sum(1 for t in toks if t.startswith("W") and t != "W0")
            n_v += sum(1 for t in toks if t in ("L", "G", "S"))
            if toks or b["succ"]:
                h.update(("%d:%s>%s;" % (i, ",".join(toks), ",".join(str(index.get(s, -1)) for s in b["succ"]))).encode())
    return dict(sha256=h.hexdigest()[:32], counted_waits=n_w, vmem_instructions=n_v)


def prologue_report(isa, sub, qp):
    """Check (1) of the module docstring for one kernel.  Paths are not enumerated one by one (exec-mask branches make millions):
    the abstract state of a path -- (LDS-DMA instructions seen so far, capped at qp; a register access seen in front of the last Q
    piece; vector-memory instructions behind the last Q piece) -- is propagated through the prologue's blocks to a fixpoint, which
    visits every distinct state a path can be in at every wait.  The graph does not know that the kernel picks its wait by the
    number of pieces it staged (`staged0 == 6 * PW_`, `rcos`): paths with fewer pieces reach the counted waits in the graph and not
    at run time.  What is checked is therefore what the INTENDED paths look like and that nothing lies beyond them:
      (a) on no path does a register load / store sit in front of or between the Q pieces;
      (b) every counted wait vmcnt(N) is reached by a path with EXACTLY N vector-memory instructions behind the last Q piece
          (the path it was written for exists as written: nothing hoisted in front of the Q pieces, nothing merged away);
      (c) no path carries more instructions behind the Q pieces than the largest counted wait (nothing was added).
    """
    problems, states_seen = set(), 0
    behind_at = {}                              # N -> set of `behind` values of the states that reach a vmcnt(N)
    for name, body in kernels(isa, sub):
        bl = blocks(body)
        index = {b["label"]: i for i, b in enumerate(bl)}
        reach = {0: {(0, False, 0)}}
        todo = [0]
        while todo:
            i = todo.pop()
            b = bl[i]
            out = set()
            for (nl, bad, behind) in reach[i]:
                alive = True
                for t in b["toks"]:
                    if t == "INIT":            # the loop region begins: the prologue's waits all lie in front of it
                        alive = False
                        break
                    if isinstance(t, tuple):
                        if t[1] > 0:
                            if nl < qp:
                                problems.add("%s: a path reaches vmcnt(%d) with only %d of the %d Q pieces issued" % (name, t[1], nl, qp))
                            elif bad:
                                problems.add("%s: a register load / store sits in front of or between the %d Q pieces on a path to "
                                             "vmcnt(%d)" % (name, qp, t[1]))
                            else:
                                behind_at.setdefault(t[1], set()).add(behind)
                        continue
                    if nl < qp:
                        if t == "L":
                            nl += 1
                        else:
                            bad = True
                    else:
                        behind = min(255, behind + 1)
                if alive:
                    out.add((nl, bad, behind))
            if not out:
                continue
            succ = [index[x] for x in b["succ"] if x in index]
            if b["fall"] and i + 1 < len(bl):
                succ.append(i + 1)
            for x in succ:
                if x <= i:                     # a back edge in front of INIT: not a prologue any more
                    problems.add("%s: loop in front of the INIT statement (block %d -> %d): the prologue check assumes none" % (name, i, x))
                    continue
                before = reach.setdefault(x, set())
                if not out <= before:
                    before |= out
                    todo.append(x)
        states_seen += sum(len(v) for v in reach.values())
    if not behind_at:
        problems.add("%s: no counted wait found in the prologue" % sub)
    for n, seen in sorted(behind_at.items()):
        if n not in seen:
            problems.add("%s: no path reaches vmcnt(%d) with exactly %d vector-memory instructions behind the last Q piece (paths carry %s): "
                         "the wait no longer matches the code it was counted for" % (sub, n, n, sorted(seen)))
    if behind_at:
        top = max(behind_at)
        most = max(max(v) for v in behind_at.values())
        if most > top:
            problems.add("%s: a path carries %d vector-memory instructions behind the Q pieces, more than the largest counted wait (%d): "
                         "hipcc added loads to the prologue" % (sub, most, top))
    problems = sorted(problems)
    return dict(ok=not problems, problems=problems, paths=states_seen, counted_waits=sorted(behind_at))


def main(argv):
    update = "--update" in argv
    files = [a for a in argv if not a.startswith("--")]
    sigs = json.load(open(SIG_FILE)) if os.path.exists(SIG_FILE) else {}
    rc = 0
    for f in files:
        base = os.path.basename(f)
        isa = open(f).read()
        if base == "attention.s":
            for kern, qp in PROLOGUE.items():
                try:
                    rep = prologue_report(isa, kern, qp)
                except KeyError:
                    if kern in OPTIONAL:
                        print("%s: not in this build (make ONEWAVE64=1)" % kern)
                        continue
                    raise
                print("%s prologue: %d (block, state) pairs up to the loop, counted waits %s: %s" % (kern, rep["paths"], rep["counted_waits"],
                                                                                     "ok" if rep["ok"] else "FAILED"))
                for p in rep["problems"][:10]:
                    print("   ", p)
                rc |= 0 if rep["ok"] else 1
        for kern in SIGNED.get(base, []):
            try:
                s = signature(isa, kern)
            except KeyError:
                if kern in OPTIONAL:
                    continue
                raise
            want = sigs.get(kern)
            if update:
                sigs[kern] = s
                print("%s: signature %s recorded (%d counted waits, %d vector-memory instructions)" % (kern, s["sha256"], s["counted_waits"],
                                                                                                      s["vmem_instructions"]))
            elif want is None or want["sha256"] != s["sha256"]:
                print("%s: vector-memory / vmcnt sequence CHANGED (have %s, validated %s): hipcc generated other code than the GPU suite "
                      "validated -- the counted waits of this kernel must be re-verified (module docstring)"
                      % (kern, s["sha256"], want and want["sha256"]))
                rc |= 1
            else:
                print("%s: vector-memory / vmcnt sequence as validated (%s)" % (kern, s["sha256"]))
    if update:
        with open(SIG_FILE, "w") as fh:
            json.dump(sigs, fh, indent=1, sort_keys=True)
            fh.write("\n")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
