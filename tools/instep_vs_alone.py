#!/usr/bin/env python3
"""What a kernel loses INSIDE the training step against the same kernel launched back to back on its own (round 5; the attention
forward ran 12 % and the backward 7 % slower in the step than in tools/fa_lib_ab.py).  Reads rocprofv3 counter passes
(`--pmc GRBM_GUI_ACTIVE --kernel-trace` and `--pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace`) of
  (a) a bench.py run                      <step dir>/grbm, <step dir>/tcc
  (b) tools/pmc_workload.py (stand-alone)  <alone dir>/grbm, <alone dir>/tcc
and prints per kernel: dispatches, median duration, the clock it held (GRBM_GUI_ACTIVE / 8 XCDs / duration, MI355X_MICROARCH.md
'DVFS give-back'), the L2 hit rate, and for the step the same split by what ran right before the dispatch (a GEMM or not).
usage: python tools/instep_vs_alone.py <step dir> <alone dir> [kernel substring ...]"""
import csv, glob, os, sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].split("<")[0].split("::")[-1].strip()


def load(d):
    """dispatch id -> dict(kernel, start, end, counters...)"""
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            x = rows.setdefault(int(r["Dispatch_Id"]), {"kernel": short(r["Kernel_Name"]), "full": r["Kernel_Name"],
                                                        "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"])})
            x[r["Counter_Name"]] = x.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


def summarize(rows, want):
    per = defaultdict(list)
    prev = None
    for r in rows:
        r["after_gemm"] = prev is not None and ("Cijk" in prev["full"] or "gemm" in prev["full"].lower())
        prev = r
        if any(w in r["kernel"] for w in want):
            per[r["kernel"]].append(r)
    return per


def line(tag, rs):
    dur = [r["end"] - r["start"] for r in rs]
    out = f"{tag:34s} n {len(rs):4d}  median {med(dur) / 1e3:9.1f} us"
    if rs and "GRBM_GUI_ACTIVE" in rs[0]:
        clk = [r["GRBM_GUI_ACTIVE"] / 8.0 / (r["end"] - r["start"]) for r in rs]
        out += f"  clock {med(clk):5.3f} GHz"
    if rs and "TCC_HIT_sum" in rs[0]:
        h, m = sum(r["TCC_HIT_sum"] for r in rs), sum(r["TCC_MISS_sum"] for r in rs)
        out += f"  L2 hit {100 * h / max(1.0, h + m):5.1f} %"
    return out


def by_predecessor(d, want):
    """One run that holds both regimes (bench.py --attn-standalone): a kernel's dispatches grouped by the kernel that ran right
    before them -- in the step the attention forward follows the rotary kernel, in the stand-alone loop the previous call's
    dK/dV kernel."""
    for sub in ("grbm", "tcc"):
        if not os.path.isdir(os.path.join(d, sub)):
            continue
        print(f"== {sub} pass")
        rows = load(os.path.join(d, sub))
        groups = defaultdict(list)
        prev = None
        for r in rows:
            if any(w in r["kernel"] for w in want):
                groups[(r["kernel"], prev["kernel"] if prev else "-")].append(r)
            prev = r
        for (k, pk), rs in sorted(groups.items()):
            if len(rs) >= 3:
                print(line(f"{k} after {pk}"[:60].ljust(60), rs))


def main():
    if sys.argv[1] == "--by-predecessor":
        return by_predecessor(sys.argv[2], sys.argv[3:] or ["fa_fwd", "fa_bwd"])
    step, alone = sys.argv[1], sys.argv[2]
    want = sys.argv[3:] or ["fa_fwd", "fa_bwd", "swiglu", "add_rmsnorm"]
    for sub in ("grbm", "tcc"):
        print(f"== {sub} pass")
        s, a = summarize(load(os.path.join(step, sub)), want), summarize(load(os.path.join(alone, sub)), want)
        for k in sorted(set(s) | set(a)):
            if k in a:
                print(line(k + " [alone]", a[k]))
            if k in s:
                print(line(k + " [step]", s[k]))
                ag, ng = [r for r in s[k] if r["after_gemm"]], [r for r in s[k] if not r["after_gemm"]]
                if ag and ng:
                    print(line("    right after a GEMM", ag))
                    print(line("    after another kernel", ng))


if __name__ == "__main__":
    main()
