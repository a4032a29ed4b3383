#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs, --kernel-trace only) -> HBM traffic per dispatch and kernel.
usage: python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>
FETCH_SIZE / WRITE_SIZE are KiB; gfx950 correction (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE x 1024."""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
        for (did, name), v in per.items():
            name = name.replace("(anonymous namespace)::", "").replace("void ", "")
            acc[name.split("(")[0].split("<")[0].split("::")[-1].strip()].append(v)
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f = sorted(fetch.get(k, [0.0])); w = sorted(write.get(k, [0.0]))
    fm, wm = f[len(f) // 2], w[len(w) // 2]                       # median dispatch
    out[k] = {"dispatches": len(f), "FETCH_SIZE_KiB": fm, "WRITE_SIZE_KiB": wm, "traffic_bytes": 2 * fm * 1024 + wm * 1024}
print(json.dumps(out, indent=1))
