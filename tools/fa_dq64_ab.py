#!/usr/bin/env python3
"""Same-process A/B of the head_dim-64 dQ kernels inside rpo_flash_attn_bwd (cfg-2 passage batch of the tests: 48 sequences of
2048..4096 tokens, 32 / 8 heads): q_block 128 (fa_bwd_dq_kernel) against q_block 64 (fa_bwd_dq64w_kernel, one wave per SIMD).  The
dK/dV kernel is the same in both arms; prints the whole backward call and, from a rocprof-free difference, nothing else -- run it
under `rocprofv3 --kernel-trace --stats` for the per-kernel split.  usage: python tools/fa_dq64_ab.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rankpo_amd import _lib
if os.environ.get("LIB"):                              # another build of the library (tools/exp/...), e.g. a generator variant
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])
    _lib._lib = None
from rankpo_amd import ops
DEV = "cuda"; torch.manual_seed(0)
hd, nh, nkv, N, L = 64, 32, 8, int(os.environ.get("NSEQ", "48")), 4096
SC = 1.0 / hd ** 0.5
lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
lens = lens.tolist(); T = sum(lens)
q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
go = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
t128 = ops.attn_tile_table(lens, DEV, nh, nkv)
t64 = ops.attn_tile_table(lens, DEV, nh, nkv, block_m=64, heads_per_block=4)
kt = ops.attn_key_tile_table(lens, DEV, nkv, ops.ATTN_KEY_BLOCK)
out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, t128, SC)
g = {n: tuple(torch.empty_like(t) for t in (q, k, v)) for n in ("old", "new")}
arms = {"old": lambda: ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, t128, kt, SC, grads=g["old"]),
        "new": lambda: ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, t64, kt, SC, grads=g["new"], q_block=64)}
if os.environ.get("ONLY"):
    arms = {os.environ["ONLY"]: arms[os.environ["ONLY"]]}
    g["old"] = g["new"] = g[os.environ["ONLY"]]
def t(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for fn in arms.values():
    for _ in range(3):
        fn()
torch.cuda.synchronize()
res = {n: [] for n in arms}
for rnd in range(7):
    for n, fn in arms.items():
        res[n].append(t(fn))
fl = sum(10 * nh * hd * n * (n + 1) / 2 for n in lens)
for n in arms:
    ts = sorted(res[n]); m = ts[len(ts) // 2]
    print(f"bwd64 {n}: median {m:.3f} ms (min {ts[0]:.3f}) = {fl / m / 1e9:.0f} TFLOP/s = {fl / m / 1e9 / 2500:.3f} of peak (algorithmic, whole backward)", flush=True)
print("max |d dq|", (g["old"][0].float() - g["new"][0].float()).abs().max().item(), "dk equal", torch.equal(g["old"][1], g["new"][1]),
      "dv equal", torch.equal(g["old"][2], g["new"][2]))
