#!/usr/bin/env python3
"""Gradient accuracy of the head_dim-128 encoder path by backward-attention variant (the gated test's model, batch and rule):
   python tools/fa128_parity_ab.py [lib.so | pytorch]      (one variant per process; no argument = the in-tree library)"""
import importlib, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from rankpo_amd import _lib
which = sys.argv[1] if len(sys.argv) > 1 else "in-tree"
if which.endswith(".so"):
    _lib.LIB_PATH = os.path.abspath(which)
import rankpo_amd
from rankpo_amd import encoder as PE, ops
if which == "pytorch":
    ops.attn_key_tile_table = lambda *a, **k: None         # no key-block table -> PyTorch's flash-attention backward op
import test_gpu_fastpath as t
bench = importlib.import_module("bench")
hd = int(os.environ.get("HD", 128))
for seed in (0, 1, 2):
    cfg, enc, model = t._model(PE, rankpo_amd, seed=seed, hd=hd)
    batch, tot = t._batch()
    w = {k: v.detach().to("cpu", torch.float32).requires_grad_(True) for k, v in enc.state_dict().items()}
    ref = bench.oracle_step(w, cfg.to_dict(), batch, t.T_CONTRASTIVE)
    rep = bench.step_parity(model, cfg, t.T_CONTRASTIVE, batch, ref, t.DEV, torch.bfloat16)
    f, c, c2 = rep["fast_path"], rep["control_stock_eager"], rep.get("control_stock_flash", {})
    print(which, "hd", hd, "seed", seed, {k: (round(f[k], 5), round(c[k], 5), round(c2.get(k, 0.0), 5)) for k in f if k != "loss"},
          "(fast, eager control, flash control) pass", rep["pass"], rep["failed"], flush=True)
