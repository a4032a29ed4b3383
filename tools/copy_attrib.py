#!/usr/bin/env python3
"""Where do the `__amd_rocclr_copyBuffer` dispatches of a profiled `bench.py` run fall: inside the training steps, or around them?
Reads rocprofv3's `*_kernel_trace.csv`, orders the dispatches by start time and cuts the timeline at every `adamw_kernel` launch
(exactly one per optimizer step: pre-size step, warm-up steps, timed steps); what follows the last one is the sweep / parity /
CPU-baseline legs (state-dict copies for the oracle, controls).
usage: python tools/copy_attrib.py <kernel_trace.csv> [kernel-name substring, default copyBuffer]"""
import csv
import sys

path = sys.argv[1]
what = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
cuts = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
print(f"{len(rows)} dispatches, {len(cuts)} optimizer steps; `{what}` per segment:\n")
print("| segment | dispatches | `%s` calls | total ms | share of segment kernel time | largest (us) |" % what)
print("|---|---:|---:|---:|---:|---|")
prev = 0
for j, i in enumerate(cuts + [len(rows)]):
    seg = rows[prev:i + 1] if i < len(rows) else rows[prev:]
    cp = [r for r in seg if what in r["Kernel_Name"]]
    tot = sum(dur(r) for r in cp) / 1e6
    allk = sum(dur(r) for r in seg) / 1e6 or 1.0
    big = sorted((dur(r) / 1e3 for r in cp), reverse=True)[:4]
    name = f"step {j} (up to its adamw_kernel)" if i < len(rows) else "after the last step (sweep, parity, CPU legs)"
    print(f"| {name} | {len(seg)} | {len(cp)} | {tot:.2f} | {100 * tot / allk:.2f} % | {', '.join(f'{b:.0f}' for b in big)} |")
    prev = i + 1
