#!/usr/bin/env python3
"""A/B of several builds of librankpo_hip.so on the flash-attention forward / backward entry points (C ABI, cfg-2 passage
batch), interleaved rounds in ONE process: python tools/fa_lib_ab.py other1.so [other2.so ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rankpo_amd import _lib, ops
libs = {"in-tree": _lib.load()}
for path in sys.argv[1:]:
    l = C.CDLL(os.path.abspath(path))
    for name in ("rpo_flash_attn_fwd", "rpo_flash_attn_bwd"):
        getattr(l, name).restype, getattr(l, name).argtypes = _lib.SIGNATURES[name]
    libs[os.path.basename(path)] = l
DEV = "cuda"; torch.manual_seed(0)
hd = int(os.environ.get("HD", "64"))                    # HD=128: cfg 5's shape (24 sequences), key blocks of 128
nh, nkv, N, L = 32, 8, (48 if hd == 64 else 24), 4096
KB = 256 if hd == 64 else 128
SC = 1.0 / hd ** 0.5
lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
lens = lens.tolist(); T = sum(lens)
q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
tiles = ops.attn_tile_table(lens, DEV, nh, nkv); kt = ops.attn_key_tile_table(lens, DEV, nkv, KB)
fl = sum(4 * nh * hd * n * (n + 1) / 2 for n in lens)
st = torch.cuda.current_stream().cuda_stream
out = {n: torch.empty(T, nh, hd, device=DEV, dtype=torch.bfloat16) for n in libs}
lse = {n: torch.empty(nh, T, device=DEV) for n in libs}
go = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
delta = torch.empty(2, nh, T, device=DEV)
dq = {n: torch.empty_like(q) for n in libs}; dk = {n: torch.empty_like(k) for n in libs}; dv = {n: torch.empty_like(v) for n in libs}
def fwd(n):
    return libs[n].rpo_flash_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), v.stride(0), cu.data_ptr(),
                                      tiles.data_ptr(), tiles.shape[0], tiles.shape[1], T, nh, nkv, hd, SC, out[n].data_ptr(), nh * hd,
                                      lse[n].data_ptr(), 0, None, None, 0, 128, st)
def bwd(n):
    return libs[n].rpo_flash_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out[n].data_ptr(), go.data_ptr(), q.stride(0), k.stride(0),
                                      v.stride(0), out[n].stride(0), go.stride(0), cu.data_ptr(), tiles.data_ptr(), tiles.shape[0],
                                      tiles.shape[1], kt.data_ptr(), kt.shape[0], KB, 0, T, nh, nkv, hd, SC, lse[n].data_ptr(),
                                      delta.data_ptr(), dq[n].data_ptr(), dk[n].data_ptr(), dv[n].data_ptr(), q.stride(0), k.stride(0),
                                      v.stride(0), None, None, 0, 128, st)
def t(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        assert fn() == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for n in libs:
    for _ in range(2):
        fwd(n); bwd(n)
res = {(n, w): [] for n in libs for w in ("fwd", "bwd")}
for rnd in range(int(os.environ.get("ROUNDS", "7"))):
    for n in libs:
        res[(n, "fwd")].append(t(lambda: fwd(n)))
        res[(n, "bwd")].append(t(lambda: bwd(n)))
first = next(iter(libs))
for (n, w), ts in res.items():
    ts.sort(); m = ts[len(ts) // 2]
    f = fl if w == "fwd" else 2.5 * fl
    print(f"{w} {n}: median {m:.3f} ms (min {ts[0]:.3f}) = {f / m / 1e9:.0f} TFLOP/s", flush=True)
for n in libs:
    if n != first:
        print(f"{n}: out identical {torch.equal(out[n], out[first])}, dq/dk/dv identical "
              f"{torch.equal(dq[n], dq[first]) and torch.equal(dk[n], dk[first]) and torch.equal(dv[n], dv[first])}")
