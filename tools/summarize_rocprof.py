#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short, committed summary (profiles/*.md):
top kernels by time with short names + every hand-written kernel of librankpo_hip.so."""
import csv
import re
import sys

OURS = ("pool_normalize", "sim_tile", "sim_skinny", "sim_rowwise", "ce_finalize", "first_finalize", "infonce_",
        "grouped_dots", "rankpo_", "adamw_kernel", "sumsq_kernel", "zero_fill", "rmsnorm", "swiglu", "rope_", "fa_",
        "topk_", "wgrad_", "transpose_kernel", "sim_small")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_") or name.startswith("Custom_Cijk_"):
        m = re.search(r"MT\d+x\d+x\d+", name)
        return ("hipBLASLt " + name.split("_BBS")[0] + " " + (m.group(0) if m else ""))[:70]
    name = re.sub(r"at::native::", "", name)
    if any(k in name for k in OURS):
        name = name.split("(")[0]                     # our kernels: the name (+ template arguments) says it all
    return name[:110]


def rows_from_db(path):
    """rocprofv3's default output is a rocpd SQLite file: the same per-kernel statistics, from its `kernels` view."""
    import sqlite3
    con = sqlite3.connect(path)
    cur = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc")
    rows = [dict(Name=n, Calls=str(c), TotalDurationNs=str(int(t)), AverageNs=str(a), MinNs=str(int(lo)), MaxNs=str(int(hi)))
            for n, c, t, a, lo, hi in cur]
    tot = sum(int(r["TotalDurationNs"]) for r in rows) or 1
    for r in rows:
        r["Percentage"] = str(100.0 * int(r["TotalDurationNs"]) / tot)
    return rows


def main(path, title):
    rows = rows_from_db(path) if path.endswith(".db") else list(csv.DictReader(open(path)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print(f"# {title}\n")
    print(f"source: `{path.split('gpurun_out/')[-1]}` (rocprofv3 --kernel-trace --stats), total GPU kernel time "
          f"{tot / 1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|")
    for r in rows[:22]:
        print(f"| {short(r['Name'])} | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    print("\n## hand-written kernels (librankpo_hip.so)\n")
    print("| kernel | calls | total ms | avg us | min us | max us |\n|---|---:|---:|---:|---:|---:|")
    for r in rows:
        if any(k in r["Name"] for k in OURS):
            print(f"| {short(r['Name'])} | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | "
                  f"{float(r['AverageNs']) / 1e3:.2f} | {int(r['MinNs']) / 1e3:.2f} | {int(r['MaxNs']) / 1e3:.2f} |")


def timed_region(trace_path, skip, count):
    """From `*_kernel_trace.csv`: launches [skip, skip + count) of each hand-written attention kernel in time order = the timed
    steps of bench.py (its pre-size and warm-up steps come first, its parity sample and sweep after), the window bench.py's own
    HIP-event average covers."""
    by = {}
    for r in csv.DictReader(open(trace_path)):
        n = r["Kernel_Name"]
        if "fa_" in n:
            by.setdefault(short(n), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    print(f"\n## attention kernels over the timed steps only (launches {skip}..{skip + count - 1} of each in time order, from the kernel trace)\n")
    print("| kernel | launches | avg us | min us | max us |\n|---|---:|---:|---:|---:|")
    tot = 0.0
    for n, v in sorted(by.items()):
        d = [x[1] for x in sorted(v)[skip:skip + count]]
        tot += sum(d) / len(d) if "bwd" in n else 0.0
        print(f"| {n} | {len(d)} | {sum(d) / len(d) / 1e3:.1f} | {min(d) / 1e3:.1f} | {max(d) / 1e3:.1f} |")
    print(f"\nbackward entry point (dQ + dK/dV kernels) = {tot / 1e3:.1f} us per call under the profiler")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 kernel stats")
    if len(sys.argv) > 5:
        timed_region(sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
