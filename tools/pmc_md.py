#!/usr/bin/env python3
"""Markdown tables from the rocprofv3 counter passes of tools/pmc_run.sh:
  python tools/pmc_md.py traffic profiles/rNN_pmc_traffic.json         -> the HBM-traffic table of profiles/rNN_pmc_traffic.md
  python tools/pmc_md.py sim <dir with sim_sq/ and sim_grbm/> [kernel]  -> MFMA utilisation of sim_tile256_kernel per shape"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def traffic(path):
    r = json.load(open(path))
    print("| entry point (kernels at HEAD) | shape | algorithmic | fetched GB (per kernel) | written GB | traffic / algorithmic | L2 hit rate | un-profiled us |")
    print("|---|---|---:|---:|---:|---:|---:|---:|")
    for e, v in r.items():
        if e == "_meta":
            continue
        pk = v["per_kernel"]
        f = " + ".join(f"{pk[k]['fetch_bytes'] / 1e9:.2f}" for k in v["kernels"] if k in pk)
        w = " + ".join(f"{pk[k]['write_bytes'] / 1e9:.2f}" for k in v["kernels"] if k in pk)
        h = " / ".join(f"{100 * pk[k].get('tcc_hit_rate', 0):.0f} %" for k in v["kernels"] if k in pk)
        print(f"| {e} ({', '.join(v['kernels'])}) | {v['shape']} | {v['algo_bytes'] / 1e9:.3f} GB | {f} | {w} | "
              f"**{v['traffic_over_algorithmic']:.3f}** | {h} | {v['event_us_unprofiled']} |")


def load(d, kernel):
    """dispatch id -> {counter: value, 'dur_ns': ..., 'grid': ...} for dispatches of `kernel`, in launch order"""
    out = defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                x = out[int(r["Dispatch_Id"])]
                x[r["Counter_Name"]] = x.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                x["dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                x["grid"] = int(r["Grid_Size"])
    return [out[k] for k in sorted(out)]


def sim(d, kernel="sim_tile256_kernel"):
    shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("SHAPES", "16384x16384x2048,4096x4096x4096,16384x16384x4096").split(",")]
    per = int(os.environ.get("DISPATCHES_PER_SHAPE", "5"))           # 2 warm-up + REPS = 3 calls per shape
    sq, gr = load(os.path.join(d, "sim_sq"), kernel), load(os.path.join(d, "sim_grbm"), kernel)
    assert len(sq) == len(gr) == per * len(shapes), (len(sq), len(gr), per, shapes)
    mean = lambda rows, k: sum(r[k] for r in rows) / len(rows)
    cols = []
    for i, (Q, P, dd) in enumerate(shapes):
        a, b = sq[i * per:(i + 1) * per], gr[i * per:(i + 1) * per]
        dur = mean(b, "dur_ns") * 1e-9                                  # the GRBM pass perturbs the kernel least (two counters)
        flop = 2.0 * Q * P * dd
        gui = mean(b, "GRBM_GUI_ACTIVE")
        clock = gui / 8 / dur
        busy = mean(a, "SQ_VALU_MFMA_BUSY_CYCLES")
        dur_sq = mean(a, "dur_ns") * 1e-9
        util = busy / (1024 * (gui / 8) * (dur_sq / dur))              # active cycles of the SQ pass scaled by its own duration
        wc = mean(a, "SQ_WAVE_CYCLES")
        cols.append({"shape": f"Q=P={Q}, d={dd}", "dur_us": dur * 1e6, "dur_sq_us": dur_sq * 1e6, "flop": flop, "tf": flop / dur / 1e12,
                     "gui": gui, "clock": clock, "busy": busy, "mops": mean(a, "SQ_INSTS_VALU_MFMA_MOPS_BF16"), "util": util, "wc": wc,
                     "wait_any": mean(a, "SQ_WAIT_ANY") / wc, "wait_inst": mean(a, "SQ_WAIT_INST_ANY") / wc,
                     "active": mean(a, "SQ_ACTIVE_INST_ANY") / wc, "conf": mean(a, "SQ_LDS_BANK_CONFLICT")})
    hdr = " | ".join(c["shape"] for c in cols)
    print(f"| quantity (mean over {per} dispatches) | {hdr} |")
    print("|---|" + "---:|" * len(cols))
    row = lambda name, fn: print(f"| {name} | " + " | ".join(fn(c) for c in cols) + " |")
    row("kernel duration (GRBM pass / SQ pass)", lambda c: f"{c['dur_us']:.1f} / {c['dur_sq_us']:.1f} us")
    row("algorithmic FLOP (2 Q P d)", lambda c: f"{c['flop']:.4g}")
    row("achieved (kernel alone)", lambda c: f"**{c['tf']:.0f} TFLOP/s = {c['tf'] / 2500:.3f} of 2.5 PF**")
    row("GRBM_GUI_ACTIVE (sum over 8 XCDs)", lambda c: f"{c['gui'] / 1e6:.3f} M")
    row("sustained clock = GRBM_GUI_ACTIVE / 8 / duration", lambda c: f"{c['clock'] / 1e9:.2f} GHz")
    row("SQ_VALU_MFMA_BUSY_CYCLES", lambda c: f"{c['busy']:.4g}")
    row("SQ_INSTS_VALU_MFMA_MOPS_BF16", lambda c: f"{c['mops']:.4g}")
    row("**MFMA utilisation** = MFMA busy / (1024 SIMDs x active cycles)", lambda c: f"**{100 * c['util']:.1f} %**")
    row("achieved / (peak x sustained clock / 2.4 GHz)", lambda c: f"{c['tf'] / (2500 * c['clock'] / 2.4e9):.3f}")
    row("SQ_WAVE_CYCLES, of which WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY",
        lambda c: f"{c['wc']:.3g}: {100 * c['wait_any']:.0f} % / {100 * c['wait_inst']:.0f} % / {100 * c['active']:.0f} %")
    row("SQ_LDS_BANK_CONFLICT (cycles)", lambda c: f"{c['conf'] / 1e6:.2f} M")


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2])
    else:
        sim(sys.argv[2], *(sys.argv[3:4]))
