#!/usr/bin/env python3
"""Interleaved A/B in ONE process (rounds x variants, medians): the query-tile work list of the forward / dQ kernels in its
two formats -- [n, 2] + grid.y = heads (round 1) vs the XCD-dealt [n, 3] list -- on the cfg-2 passage batch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rankpo_amd import ops
DEV = "cuda"; torch.manual_seed(0)
hd = int(os.environ.get("HD", "64"))                    # HD=128: cfg 5's shape (24 sequences), key blocks of 128
nh, nkv, N, L = 32, 8, (48 if hd == 64 else 24), 4096
KB = 256 if hd == 64 else 128
SC = 1.0 / hd ** 0.5
lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
lens = lens.tolist()
if os.environ.get("WITH_QUERIES", "0") == "1":           # the packed training batch: + 8 (hd 64) / 4 query rows of <= 1280 tokens + a filler row
    lens += torch.randint(640, 1281, (8 if hd == 64 else 4,)).tolist()
    lens += [(-sum(lens)) % 256 or 256]
T = sum(lens)
q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
kt = ops.attn_key_tile_table(lens, DEV, nkv, KB)
tabs = {"list2": ops.attn_tile_table(lens, DEV), "xcd3": ops.attn_tile_table(lens, DEV, nh, nkv)}
fl = sum(4 * nh * hd * n * (n + 1) / 2 for n in lens)
out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tabs["list2"], SC)
go = torch.randn_like(out)


def t(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {(a, b): [] for a in tabs for b in ("fwd", "bwd")}
for name, tb in tabs.items():
    for _ in range(3):
        ops.flash_attn_varlen_fwd(q, k, v, cu, tb, SC); ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tb, kt, SC, key_block=KB)
for rnd in range(int(os.environ.get("ROUNDS", "7"))):
    for name, tb in tabs.items():
        res[(name, "fwd")].append(t(lambda: ops.flash_attn_varlen_fwd(q, k, v, cu, tb, SC)))
        res[(name, "bwd")].append(t(lambda: ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tb, kt, SC, key_block=KB)))
# dK/dV schedule: heaviest-first list + ascending sweep (round 1) vs group-ordered list + downward slice-major sweep
kts = {"heavy/up": (ops.attn_key_tile_table(lens, DEV, nkv, KB, group_order=False), False),
       "group/down": (ops.attn_key_tile_table(lens, DEV, nkv, KB, group_order=True), True),
       "heavy/down": (ops.attn_key_tile_table(lens, DEV, nkv, KB, group_order=False), True),
       "group/up": (ops.attn_key_tile_table(lens, DEV, nkv, KB, group_order=True), False),
       **{f"group+heavy tail {f}/down": (ops.attn_key_tile_table(lens, DEV, nkv, KB, group_order=f), True)
          for f in (0.2, 0.35, 0.5, 0.65, 0.8)}}
tb3 = tabs["xcd3"]
r2 = {n: [] for n in kts}
for n, (ktab, down) in kts.items():
    for _ in range(2):
        ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tb3, ktab, SC, sweep_down=down, key_block=KB)
for rnd in range(int(os.environ.get("ROUNDS", "7"))):
    for n, (ktab, down) in kts.items():
        r2[n].append(t(lambda: ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tb3, ktab, SC, sweep_down=down, key_block=KB)))
for n, ts in r2.items():
    ts.sort()
    print(f"bwd (xcd3 q list) dK/dV schedule {n}: median {ts[len(ts)//2]:.3f} ms (min {ts[0]:.3f}) = {2.5 * fl / ts[len(ts)//2] / 1e9:.0f} TFLOP/s", flush=True)
ra = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tb3, kts["heavy/up"][0], SC, sweep_down=False, key_block=KB)
rb = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tb3, kts["group/down"][0], SC, sweep_down=True, key_block=KB)
print("dK/dV schedules: dq identical", torch.equal(ra[0], rb[0]), "; max |d dk|, |d dv| =",
      float((ra[1].float() - rb[1].float()).abs().max()), float((ra[2].float() - rb[2].float()).abs().max()),
      "; |dk|max", float(ra[1].float().abs().max()))
for (name, what), ts in res.items():
    ts.sort(); med = ts[len(ts) // 2]
    f = fl if what == "fwd" else 2.5 * fl
    print(f"{what} {name}: median {med:.3f} ms (min {ts[0]:.3f}) = {f / med / 1e9:.0f} TFLOP/s", flush=True)
a = ops.flash_attn_varlen_fwd(q, k, v, cu, tabs["list2"], SC); b = ops.flash_attn_varlen_fwd(q, k, v, cu, tabs["xcd3"], SC)
print("fwd outputs bit-identical:", torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
a = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tabs["list2"], kt, SC, key_block=KB)
b = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tabs["xcd3"], kt, SC, key_block=KB)
print("bwd outputs bit-identical:", all(torch.equal(x, y) for x, y in zip(a, b)))
