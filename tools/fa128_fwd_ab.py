#!/usr/bin/env python3
"""Same-process A/B of builds of librankpo_hip.so on the attention FORWARD (C ABI; head_dim 128, cfg 5's shape: 24 sequences of
2048..4096 tokens, 32 / 8 heads; HD=64 in the environment: 48 such sequences at head_dim 64), each arm with the query-tile work list of
ITS block size (64x4 = the one-wave-per-SIMD kernel, q_block 64):
    python tools/fa128_fwd_ab.py name=path.so[:block_m] ...        (block_m 128 when omitted; the in-tree build is always arm 0)
Interleaved rounds, median per arm, outputs compared with arm 0 (every (WAVES, SUB) instantiation is bit-identical by design;
the RPO_F128_EXP timing ablations are not)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rankpo_amd import _lib, ops
arms = [("in-tree", _lib.load(), (int(os.environ.get("BM0", "128")), int(os.environ.get("HPB0", "1"))))]
for a in sys.argv[1:]:
    name, rest = a.split("=", 1)
    path, _, bm = rest.partition(":")
    bm, _, hpb = bm.partition("x")                       # name=path.so:block_m[xheads_per_block]
    l = C.CDLL(os.path.abspath(path))
    l.rpo_flash_attn_fwd.restype, l.rpo_flash_attn_fwd.argtypes = _lib.SIGNATURES["rpo_flash_attn_fwd"]
    arms.append((name, l, (int(bm or 128), int(hpb or 1))))
DEV = "cuda"; torch.manual_seed(0)
hd = int(os.environ.get("HD", "128"))                 # HD=64: the cfg-2 passage batch of the tests (48 sequences)
nh, nkv, N, L = 32, 8, int(os.environ.get("NSEQ", "24" if hd == 128 else "48")), 4096
SC = 1.0 / hd ** 0.5
lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
lens = lens.tolist(); T = sum(lens)
q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
tiles = {bm: ops.attn_tile_table(lens, DEV, nh, nkv, block_m=bm[0], heads_per_block=bm[1]) for bm in {a[2] for a in arms}}
fl = sum(4 * nh * hd * n * (n + 1) / 2 for n in lens)
st = torch.cuda.current_stream().cuda_stream
out = {n: torch.zeros(T, nh, hd, device=DEV, dtype=torch.bfloat16) for n, _, _ in arms}
lse = {n: torch.zeros(nh, T, device=DEV) for n, _, _ in arms}
def fwd(name, lib, bm):
    tl = tiles[bm]
    return lib.rpo_flash_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), v.stride(0), cu.data_ptr(),
                                  tl.data_ptr(), tl.shape[0], tl.shape[1], T, nh, nkv, hd, SC, out[name].data_ptr(), nh * hd,
                                  lse[name].data_ptr(), 0, None, None, 0, 64 if bm[0] == 64 else 128, st)
def t(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        assert fn() == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for a in arms:
    for _ in range(3):
        assert fwd(*a) == 0
torch.cuda.synchronize()
res = {a[0]: [] for a in arms}
for rnd in range(int(os.environ.get("ROUNDS", "7"))):
    for a in arms:
        res[a[0]].append(t(lambda: fwd(*a)))
base = arms[0][0]
for name, _, bm in arms:
    ts = sorted(res[name]); m = ts[len(ts) // 2]
    same = "" if name == base else (f"  out identical {torch.equal(out[name], out[base])}, lse identical {torch.equal(lse[name], lse[base])}"
                                    f", max |d out| {(out[name].float() - out[base].float()).abs().max().item():.3g}"
                                    f", max |d lse| {(lse[name] - lse[base]).abs().max().item():.3g}")
    print(f"fwd{hd} {name:14s} block_m {bm}: median {m:.3f} ms (min {ts[0]:.3f}) = {fl / m / 1e9:.0f} TFLOP/s = {fl / m / 1e9 / 2500:.3f} of peak{same}", flush=True)
