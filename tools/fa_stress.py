"""Stress test of the hand-written attention backward (random shapes, determinism, agreement with PyTorch's op).
usage: python tools/fa_stress.py [cases]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rankpo_amd import ops
DEV = "cuda"
rs = np.random.RandomState(123)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0
for ci in range(cases):
    nkv = int(rs.choice([1, 2, 4, 8]))
    nh = nkv * int(rs.choice([1, 2, 4]))
    N = int(rs.randint(1, 24))
    hi = int(rs.choice([40, 300, 700, 1500, 2600]))
    lens = [int(x) for x in rs.randint(1, hi + 1, size=N)]
    T = sum(lens)
    torch.manual_seed(ci)
    q = torch.randn(T, nh, 64, device=DEV).to(torch.bfloat16)
    k = torch.randn(T, nkv, 64, device=DEV).to(torch.bfloat16)
    v = torch.randn(T, nkv, 64, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    tiles = ops.attn_tile_table(lens, DEV); kt = ops.attn_key_tile_table(lens, DEV, nkv)
    out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, 0.125)
    go = torch.randn_like(out)
    a = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, 0.125)
    for _ in range(int(os.environ.get("STRESS_REPEATS", "1"))):
        b = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, 0.125)
        assert all(torch.equal(x, y) for x, y in zip(a, b)), ("not deterministic", ci, lens)
    r = torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False)
    d = torch.ops.aten._flash_attention_backward(go, q, k, v, r[0], r[1], cu, cu, max(lens), max(lens), 0.0, True, r[2], r[3])
    for name, x, y in zip(("dq", "dk", "dv"), a, d):
        assert torch.isfinite(x.float()).all(), (name, "non-finite", ci, lens)
        err = (x.float() - y.float()).abs().max().item() / max(1.0, y.float().abs().max().item())
        worst = max(worst, err)
        assert err < 0.03, (name, err, ci, nh, nkv, lens)
    if ci % 10 == 9:
        print(f"{ci + 1} cases ok, worst relative error {worst:.4f}", flush=True)
print("stress ok:", cases, "cases, worst relative error %.4f" % worst)
