"""Diagnostic: where an iteration of fa_bwd_dq64w_kernel (head_dim-64 dQ kernel, one wave per SIMD) spends its cycles -- a library built
with -DRPO_FA_STAMP -DRPO_FA_STAMP_DQ (tools/exp/build_variant.sh dq_stamp -DRPO_FA_STAMP -DRPO_FA_STAMP_DQ).  s_memtime stamps between
the generated statements (each stamp drains the scalar queue: ratios).  usage: python tools/fa_stamp_dq64w.py lib.so"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rankpo_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
_lib._lib = None
from rankpo_amd import ops
lib = _lib.load()
lib.rpo_debug_fa_stamps.restype = C.c_int
lib.rpo_debug_fa_stamps.argtypes = [C.c_void_p, C.c_int]
DEV = "cuda"; torch.manual_seed(0)
hd, nh, nkv, N, L = 64, 32, 8, 48, 4096
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
out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, t128, hd ** -0.5)
g = tuple(torch.empty_like(t) for t in (q, k, v))
call = lambda: ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, t64, kt, hd ** -0.5, grads=g, q_block=64)
buf = (C.c_ulonglong * 64)()
for _ in range(2):
    call()
torch.cuda.synchronize()
lib.rpo_debug_fa_stamps(buf, 1)
call(); torch.cuda.synchronize()
lib.rpo_debug_fa_stamps(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(4, 16)
blocks = sum((n + 63) // 64 for n in lens) * (nh // 4)
names = ["wait for the rings (vmcnt)", "s_barrier", "hipcc staging (not in-stream)", "X1: chains A(t+1) + rest of C(t) + K^T / V reads", "mask",
         "X2: D(t) + B(t+1) + first part of C(t+1) + K reads + DMA"]
print(f"{blocks} blocks; cycles per FULL iteration (32-key tile) and wave")
for i in range(6):
    print("%-60s" % names[i] + "".join("%9.0f" % (a[w, i] / a[w, 7]) for w in range(4)))
print("%-60s" % "sum per iteration" + "".join("%9.0f" % (a[w, :6].sum() / a[w, 7]) for w in range(4)))
print("%-60s" % "prologue cycles per block" + "".join("%9.0f" % (a[w, 6] / blocks) for w in range(4)))
print("%-60s" % "epilogue cycles per block (stores landed)" + "".join("%9.0f" % (a[w, 8] / blocks) for w in range(4)))
print("%-60s" % "full iterations per block" + "".join("%9.1f" % (a[w, 7] / blocks) for w in range(4)))
