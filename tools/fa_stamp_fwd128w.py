"""Diagnostic: where an iteration of fa_fwd128w_kernel (head_dim-128 forward, one wave per SIMD) spends its cycles -- a library built
with -DRPO_FA_STAMP -DRPO_FA_STAMP_FWD (tools/exp/build_variant.sh fw_stamp -DRPO_FA_STAMP -DRPO_FA_STAMP_FWD).  s_memtime stamps between
the generated statements (each stamp drains the LDS queue: the numbers are for RATIOS).  usage: python tools/fa_stamp_fwd128w.py lib.so"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rankpo_amd import _lib, ops
lib = C.CDLL(os.path.abspath(sys.argv[1]))
lib.rpo_flash_attn_fwd.restype, lib.rpo_flash_attn_fwd.argtypes = _lib.SIGNATURES["rpo_flash_attn_fwd"]
lib.rpo_debug_fa_stamps.restype = C.c_int
lib.rpo_debug_fa_stamps.argtypes = [C.c_void_p, C.c_int]
DEV = "cuda"; torch.manual_seed(0)
hd = int(os.environ.get("HD", "128"))
nh, nkv, N, L = 32, 8, 24 if hd == 128 else 48, 4096
lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
lens = lens.tolist(); T = sum(lens)
q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
tl = ops.attn_tile_table(lens, DEV, nh, nkv, block_m=64, heads_per_block=4)
out = torch.zeros(T, nh, hd, device=DEV, dtype=torch.bfloat16); lse = torch.zeros(nh, T, device=DEV)
st = torch.cuda.current_stream().cuda_stream
call = lambda: lib.rpo_flash_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), v.stride(0), cu.data_ptr(),
                                      tl.data_ptr(), tl.shape[0], tl.shape[1], T, nh, nkv, hd, 1.0 / hd ** 0.5, out.data_ptr(), nh * hd,
                                      lse.data_ptr(), 0, None, None, 0, 64, st)
buf = (C.c_ulonglong * 64)()
for _ in range(2):
    assert call() == 0
torch.cuda.synchronize()
lib.rpo_debug_fa_stamps(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
lib.rpo_debug_fa_stamps(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(4, 16)
blocks = sum((n + 63) // 64 for n in lens) * (nh // 4)
names = ["wait for the ring (vmcnt)", "s_barrier", "hipcc staging (not in-stream)", "P1: chains(t+1) + exp(t) + V^T reads", "mask",
         "P2: products(t) + max(t+1) + K reads + DMA + check"]
print(f"forward with stamps: {e0.elapsed_time(e1):.2f} ms; {blocks} blocks; cycles per FULL iteration (32-key tile) and wave")
print("%-52s" % "segment" + "".join("%9s" % f"w{w}" for w in range(4)))
for i in range(6):
    print("%-52s" % names[i] + "".join("%9.0f" % (a[w, i] / a[w, 7]) for w in range(4)))
print("%-52s" % "sum per iteration" + "".join("%9.0f" % (a[w, :6].sum() / a[w, 7]) for w in range(4)))
for i, nm in ((12, "prologue: tile entry + sequence bounds (scalar)"), (13, "prologue: Q pieces + first staging issued"), (14, "prologue: O = 0, Q landed, fragments read"),
              (9, "prologue: all of the above + rope + Q -> AGPR"), (10, "prologue: INIT (ones, scale)"), (11, "prologue: first tiles landed + barrier"),
              (8, "epilogue: O / l out of AGPRs, stores landed")):
    print("%-52s" % nm + "".join("%9.0f" % (a[w, i] / blocks) for w in range(4)))
print("%-52s" % "prologue cycles per block, all of it" + "".join("%9.0f" % (a[w, 6] / blocks) for w in range(4)))
print("%-52s" % "full iterations per block" + "".join("%9.1f" % (a[w, 7] / blocks) for w in range(4)))
