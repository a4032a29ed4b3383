#!/usr/bin/env python3
"""<dir>/{fetch,write,tcc} (rocprofv3 --pmc passes of tools/pmc_workload.py) + <dir>/algo.json -> profiles/rNN_pmc_traffic.json.
HBM bytes per dispatch = 2 x FETCH_SIZE x 1024 (gfx950 correction, MI355X_MICROARCH.md, HBM section) + WRITE_SIZE x 1024, median
dispatch of each kernel; an entry point's traffic is the sum over the kernels it launches.
usage: python tools/pmc_assemble.py <dir> <out.json> [git head]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].split("<")[0].split("::")[-1].strip()


def load(d, counters):
    """kernel -> counter -> list of per-dispatch values, dispatches in launch order"""
    acc = defaultdict(lambda: defaultdict(dict))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in counters:
                k = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                did = int(r["Dispatch_Id"])
                k[did] = k.get(did, 0.0) + float(r["Counter_Value"])
    return acc


def seq(acc, kernel, counter):
    d = acc.get(kernel, {}).get(counter, {})
    return [d[k] for k in sorted(d)]


def main():
    src, out = sys.argv[1], sys.argv[2]
    head = sys.argv[3] if len(sys.argv) > 3 else None
    algo = json.load(open(os.path.join(src, "algo.json")))
    fetch = load(os.path.join(src, "fetch"), {"FETCH_SIZE"})
    write = load(os.path.join(src, "write"), {"WRITE_SIZE"})
    tcc = load(os.path.join(src, "tcc"), {"TCC_HIT_sum", "TCC_MISS_sum"})
    # a kernel that serves several entries (attention at two head dims has distinct kernels; transpose etc. appear once) is
    # launched (1 + REPS) times per entry, in the order of algo.json: slice the dispatch sequence accordingly
    seen = defaultdict(int)
    res = {"_meta": {"head": head, "source": "tools/pmc_workload.py under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum "
                                             "(separate passes, --kernel-trace only); read bytes = 2 x FETCH_SIZE KiB (gfx950), median dispatch",
                     "round": 5}}
    reps = None
    for entry, a in algo.items():
        per, tot_f, tot_w = {}, 0.0, 0.0
        for k in a["kernels"]:
            f_all, w_all = seq(fetch, k, "FETCH_SIZE"), seq(write, k, "WRITE_SIZE")
            n_entries = sum(1 for b in algo.values() if k in b["kernels"])
            n = len(f_all) // n_entries if n_entries else 0
            reps = n
            i0 = seen[k] * n
            f, w = sorted(f_all[i0:i0 + n]), sorted(w_all[i0:i0 + n])
            h, m = seq(tcc, k, "TCC_HIT_sum")[i0:i0 + n], seq(tcc, k, "TCC_MISS_sum")[i0:i0 + n]
            seen[k] += 1
            if not f or not w:
                continue
            fb, wb = 2 * f[len(f) // 2] * 1024, w[len(w) // 2] * 1024
            per[k] = {"fetch_bytes": fb, "write_bytes": wb, "dispatches": n}
            if h and m and (sum(h) + sum(m)) > 0:
                per[k]["tcc_hit_rate"] = round(sum(h) / (sum(h) + sum(m)), 4)
            tot_f += fb
            tot_w += wb
        res[entry] = {"shape": a["shape"], "kernels": a["kernels"], "algo_bytes": a["algo_bytes"], "algo_flops": a["algo_flops"],
                      "fetch_bytes": tot_f, "write_bytes": tot_w,
                      "traffic_over_algorithmic": round((tot_f + tot_w) / a["algo_bytes"], 4) if per else None,
                      "event_us_unprofiled": a["event_us"], "per_kernel": per}
    json.dump(res, open(out, "w"), indent=1)
    for e, r in res.items():
        if e != "_meta":
            print(f"{e:28s} traffic / algorithmic = {r['traffic_over_algorithmic']}")


if __name__ == "__main__":
    main()
