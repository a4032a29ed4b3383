#!/usr/bin/env python3
"""Generate tests/golden/lowp.npz by RUNNING THE REFERENCE (yflyzhang/RankPO) in 16-bit storage in the build container:

  * RankPO in bf16 (rankpo_trainer.py:436-443 is a bf16 `matmul` whose [B, 2] scores are bf16; :545-566 the loss chain in bf16)
    and in fp16 -- the same 48 knob cases as tests/golden/rankpo.npz (tools/make_golden.py:gen_rankpo), same seeded inputs;
  * the contrastive forward / backward / eval in fp16 (modeling.py:281-322; the reference's BGE setup trains in fp16:
    configs/ds_zero1_config_bge.json:2-11, modeling.py:417, 453-454) at d = 64 / 384 / 2048 on the inputs of contrastive.npz.

Runs only where /root/reference exists (never on the GPU box).  The fixture is data only (outputs + the seed rule of the inputs).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_lowp.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import itertools
import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, make_ref_model, seeded, unit  # noqa: E402


def gen_contrastive_fp16(modeling, tmpdir, rec):
    import torch
    for d in (64, 384, 2048):
        seed = 1000 + d
        qn = unit(seeded(seed, 8, d))
        pn = unit(seeded(seed + 1, 48, d))
        pn[::6] = unit(pn[::6] + (2.0 / np.sqrt(d)) * qn)
        for mode in ("inbatch", "noinbatch", "eval"):
            m, _, _ = make_ref_model(modeling, tmpdir, "llama", temperature=0.02, use_inbatch_neg=(mode != "noinbatch"))
            m.embed = lambda x: x
            q = torch.tensor(qn, dtype=torch.float32).to(torch.float16).requires_grad_(True)
            p = torch.tensor(pn, dtype=torch.float32).to(torch.float16).requires_grad_(True)
            if mode == "eval":
                m.eval()
                o = m(query=q, passage=p)
                out = dict(scores=o.scores.detach().float().numpy())
            else:
                m.train()
                o = m(query=q, passage=p)
                o.loss.backward()
                out = dict(scores=o.scores.detach().float().numpy(), loss=np.float64(o.loss.item()),
                           dq=q.grad.float().numpy(), dp=p.grad.float().numpy())
            if d == 2048:
                R = seeded(77, d, 8)
                for k in ("dq", "dp"):
                    if k in out:
                        out[k + "_proj"] = out.pop(k).astype(np.float64) @ R
            for k, v in out.items():
                rec[f"contrastive_{mode}_d{d}_fp16_{k}"] = v


def gen_rankpo_lowp(rankpo_trainer, rec):
    import torch
    T = rankpo_trainer.RankPOTrainer
    B, d = 8, 64
    qn = unit(seeded(21, B, d))
    pn = unit(seeded(22, 2 * B, d))
    pn[0::2] = unit(pn[0::2] + 0.4 * qn)
    pn[5] = unit(pn[5] + 1.0 * qn[2])
    ref_c = 0.1 * seeded(23, B)
    ref_r = 0.1 * seeded(24, B)
    cases = {"bf16": [], "fp16": []}
    for tag, tdt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        combos = itertools.product(("sigmoid", "hinge"), (0.0, 0.1), (True, False), (0.0, 0.5), (0.0, 0.5), (1.0, 0.0))
        for ci, (lt, ls, rf, sw, gbr, rw) in enumerate(combos):
            if rw == 0.0 and sw == 0.0:
                continue
            acc = SimpleNamespace(device=torch.device("cpu"), gather_for_metrics=lambda x: x)
            ns = SimpleNamespace(beta=2.0, gamma_beta_ratio=gbr, temperature=0.1, sft_weight=sw, rankpo_weight=rw,
                                 label_smoothing=ls, loss_type=lt, reference_free=rf, accelerator=acc)
            ns.single_forward = lambda model, inputs: inputs
            ns.concatenated_forward = lambda model, batch, ns=ns: T.concatenated_forward(ns, model, batch)
            ns.rankpo_loss = lambda *a, ns=ns: T.rankpo_loss(ns, *a)
            with_ref = not rf

            class RefModel:
                pass
            if with_ref:
                ns.ref_model = RefModel()
                # the ref model's scores are a bf16 / fp16 matmul output too (rankpo_trainer.py:468-477)
                rs = torch.tensor(np.stack([ref_c, ref_r], 1), dtype=torch.float32).to(tdt)
                real_cf = ns.concatenated_forward
                ns.concatenated_forward = lambda model, batch, rs=rs, real_cf=real_cf: (
                    rs if isinstance(model, RefModel) else real_cf(model, batch))
            else:
                ns.ref_model = None
            q = torch.tensor(qn, dtype=torch.float32).to(tdt).requires_grad_(True)
            p = torch.tensor(pn, dtype=torch.float32).to(tdt).requires_grad_(True)
            loss, metrics = T.get_batch_loss_metrics(ns, None, {"query": q, "passage": p}, "train")
            loss.backward()
            with torch.no_grad():
                sc = T.concatenated_forward(ns, None, {"query": q, "passage": p})
            name = f"rankpo_{tag}_c{ci}"
            cases[tag].append(dict(name=name, loss_type=lt, label_smoothing=ls, reference_free=rf, sft_weight=sw,
                                   gamma_beta_ratio=gbr, rankpo_weight=rw, beta=2.0, temperature=0.1,
                                   metrics={k: float(v) for k, v in metrics.items()}, loss=float(loss.item()),
                                   loss_dtype=str(loss.dtype)))
            rec[name + "_scores"] = sc.float().numpy()
            rec[name + "_dq"] = q.grad.float().numpy()
            rec[name + "_dp"] = p.grad.float().numpy()
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    a = ap.parse_args()
    modeling, rankpo_trainer, _ = import_reference()
    import torch
    rec = {}
    with tempfile.TemporaryDirectory() as tmp:
        gen_contrastive_fp16(modeling, tmp, rec)
    cases = gen_rankpo_lowp(rankpo_trainer, rec)
    meta = dict(torch=torch.__version__, rankpo_cases=cases,
                inputs="contrastive: the seed rule of contrastive.npz; rankpo: q / p / ref_chosen / ref_rejected of rankpo.npz, "
                       "cast to the storage dtype (the ref scores too)",
                note="the reference's own arithmetic in 16-bit storage on the host CPU (torch's CPU kernels accumulate the "
                     "matmul in float32 and round once, like the GPU kernels it trains with)")
    np.savez_compressed(os.path.join(a.out, "lowp.npz"), meta=json.dumps(meta), **rec)
    print("wrote", os.path.join(a.out, "lowp.npz"), len(rec), "arrays")


if __name__ == "__main__":
    main()
