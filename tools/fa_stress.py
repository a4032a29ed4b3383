"""Stress test of the hand-written attention forward and backward (random shapes, determinism, agreement with PyTorch's op).
usage: [HD=128] [ROPE=1] python tools/fa_stress.py [cases]   (HD: head_dim 64 / 128; ROPE=1: rotary folded into forward and backward)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rankpo_amd import ops
DEV = "cuda"
rs = np.random.RandomState(123)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0
HD = int(os.environ.get("HD", "64"))
KB = ops.ATTN_KEY_BLOCK if HD == 64 else ops.ATTN_KEY_BLOCK_HD128
SC = 1.0 / HD ** 0.5
for ci in range(cases):
    nkv = int(rs.choice([1, 2, 4, 8]))
    nh = nkv * int(rs.choice([1, 2, 4]))
    N = int(rs.randint(1, 24))
    hi = int(rs.choice([40, 300, 700, 1500, 2600]))
    lens = [int(x) for x in rs.randint(1, hi + 1, size=N)]
    T = sum(lens)
    torch.manual_seed(ci)
    q = torch.randn(T, nh, HD, device=DEV).to(torch.bfloat16)
    k = torch.randn(T, nkv, HD, device=DEV).to(torch.bfloat16)
    v = torch.randn(T, nkv, HD, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    tiles = ops.attn_tile_table(lens, DEV, nh, nkv) if ci % 2 else ops.attn_tile_table(lens, DEV); kt = ops.attn_key_tile_table(lens, DEV, nkv, KB)
    out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, SC)
    go = torch.randn_like(out)
    a = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, SC, key_block=KB)
    for _ in range(int(os.environ.get("STRESS_REPEATS", "1"))):
        b = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, SC, key_block=KB)
        assert all(torch.equal(x, y) for x, y in zip(a, b)), ("not deterministic", ci, lens)
    r = torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False, scale=SC)
    out2, lse2 = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, SC)
    assert torch.equal(out, out2) and torch.equal(lse, lse2), ("forward not deterministic", ci, lens)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all(), ("forward non-finite", ci, lens)
    ferr = (out.float() - r[0].float()).abs().max().item() / max(1.0, r[0].float().abs().max().item())
    worst = max(worst, ferr)
    assert ferr < 0.02, ("out", ferr, ci, nh, nkv, lens)
    d = torch.ops.aten._flash_attention_backward(go, q, k, v, r[0], r[1], cu, cu, max(lens), max(lens), 0.0, True, r[2], r[3], scale=SC)
    for name, x, y in zip(("dq", "dk", "dv"), a, d):
        assert torch.isfinite(x.float()).all(), (name, "non-finite", ci, lens)
        err = (x.float() - y.float()).abs().max().item() / max(1.0, y.float().abs().max().item())
        worst = max(worst, err)
        assert err < 0.03, (name, err, ci, nh, nkv, lens)
    if ci % 10 == 9:
        print(f"{ci + 1} cases ok, worst relative error {worst:.4f}", flush=True)
print("stress ok:", cases, "cases, worst relative error %.4f" % worst)
