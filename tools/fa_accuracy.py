#!/usr/bin/env python3
"""Relative L2 error of out / dq / dk / dv against the f32 reference, head_dim 64 vs 128, hand-written kernels vs PyTorch's
flash-attention ops, on randn inputs with a logit spread `SIGMA` (q scaled).  python tools/fa_accuracy.py"""
import os, sys, math
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from rankpo_amd import ops
from test_gpu_attention import ref_attention
DEV = "cuda"
lens = [320, 211, 256, 300, 160, 129]
T = sum(lens)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
for sigma in (1.0, 3.0):
    for hd, nh, nkv in ((64, 8, 2), (128, 4, 2)):
        torch.manual_seed(hd)
        q = (torch.randn(T, nh, hd, device=DEV) * sigma).to(torch.bfloat16)
        k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
        v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
        go = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
        scale = hd ** -0.5
        qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
        ro, _ = ref_attention(qr, kr, vr, lens, scale)
        ro.backward(go.float())
        refs = dict(out=ro.detach(), dq=qr.grad, dk=kr.grad, dv=vr.grad)
        tiles = ops.attn_tile_table(lens, DEV, nh, nkv)
        kb = ops.ATTN_KEY_BLOCK if hd == 64 else ops.ATTN_KEY_BLOCK_HD128
        kt = ops.attn_key_tile_table(lens, DEV, nkv, block_n=kb)
        res = {}
        for name, ktab in (("HIP", kt), ("HIP fwd + PyTorch bwd", None)):
            a, b, c = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
            o = ops.flash_attn_varlen(a, b, c, cu, tiles, max(lens), scale, k_tiles=ktab, key_block=kb)
            o.backward(go)
            res[name] = dict(out=o.detach(), dq=a.grad, dk=b.grad, dv=c.grad)
        a, b, c = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        r = torch.ops.aten._flash_attention_forward(a, b, c, cu, cu, max(lens), max(lens), 0.0, True, False, scale=scale)
        r[0].backward(go)
        res["PyTorch fwd + bwd"] = dict(out=r[0].detach(), dq=a.grad, dk=b.grad, dv=c.grad)
        # eager bf16 control: softmax in f32, P rounded to bf16 (HF eager)
        a, b, c = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        outs, o0 = [], 0
        for n in lens:
            qq = a[o0:o0 + n].transpose(0, 1); kk = b[o0:o0 + n].transpose(0, 1).repeat_interleave(nh // nkv, 0)
            vv = c[o0:o0 + n].transpose(0, 1).repeat_interleave(nh // nkv, 0)
            s = (qq @ kk.transpose(1, 2)) * scale
            s = s.masked_fill(~torch.ones(n, n, dtype=torch.bool, device=DEV).tril(), float("-inf"))
            outs.append((torch.softmax(s.float(), -1).to(torch.bfloat16) @ vv).transpose(0, 1)); o0 += n
        o = torch.cat(outs); o.backward(go)
        res["eager bf16"] = dict(out=o.detach(), dq=a.grad, dk=b.grad, dv=c.grad)
        for name, d in res.items():
            print(f"sigma {sigma} hd {hd:3d} {name:24s} " + "  ".join(
                f"{kk} {((d[kk].float() - refs[kk]).norm() / refs[kk].norm()).item():.5f}" for kk in ("out", "dq", "dk", "dv")), flush=True)
