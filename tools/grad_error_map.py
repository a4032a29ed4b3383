#!/usr/bin/env python3
"""Per-parameter gradient error of the fast path and of the stock eager bf16 control against the float32 oracle (the gated
test's model and batch).  HD=64|128 python tools/grad_error_map.py [pytorch]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rankpo_amd
from rankpo_amd import encoder as PE, ops
from oracle import encoder_ref as E
import test_gpu_fastpath as t
if len(sys.argv) > 1 and sys.argv[1] == "pytorch":
    ops.attn_key_tile_table = lambda *a, **k: None
hd = int(os.environ.get("HD", 128))
cfg, enc, model = t._model(PE, rankpo_amd, seed=0, hd=hd)
batch, tot = t._batch()
w = {k: v.detach().to("cpu", torch.float32).requires_grad_(True) for k, v in enc.state_dict().items()}
loss = E.contrastive_step(w, cfg.to_dict(), batch, t.T_CONTRASTIVE)[0]
loss.backward()
gb = {k: {kk: vv.to(t.DEV) for kk, vv in v.items()} for k, v in batch.items()}
enc.zero_grad()
model(**gb).loss.backward()
fast = {n: p.grad.detach().float().cpu() for n, p in enc.named_parameters()}
wd = {k: v.detach().to(t.DEV, torch.bfloat16).requires_grad_(True) for k, v in enc.state_dict().items()}
lc = E.contrastive_step(wd, cfg.to_dict(), gb, t.T_CONTRASTIVE, dtype=torch.bfloat16)[0]
lc.backward()
print(f"hd {hd}: oracle loss {loss.item():.5f} fast/control losses differ by {abs(lc.item() - loss.item()):.4f} (control)")
for n in fast:
    if not (n.startswith("layers.0.") or n.startswith("layers.2.") or "embed" in n or n == "norm.weight"):
        continue
    gr = w[n].grad
    ef = ((fast[n] - gr).norm() / gr.norm()).item()
    ec = ((wd[n].grad.float().cpu() - gr).norm() / gr.norm()).item()
    print(f"  {n:44s} fast {ef:.4f}  control {ec:.4f}  ratio {ef / ec:.2f}")
