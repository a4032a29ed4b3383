#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (yflyzhang/RankPO) in the build container.

Runs only where `/root/reference` exists (never on the GPU box).  It imports the reference's
`src/modeling.py`, `src/rankpo_trainer.py` and `src/data_utils.py` unmodified, drives the hot-path
functions on seeded inputs and stores inputs (or the seed that regenerates them) + outputs as small
fixtures.  The fixtures are data only; no reference source text is stored.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import itertools
import json
import os
import random
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

REF_SRC = "/root/reference/src"


def seeded(seed, *shape):
    """Inputs are regenerated in tests from the seed: RandomState.randn is frozen across numpy versions."""
    return np.random.RandomState(seed).randn(*shape)


def unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def import_reference():
    import torch  # noqa
    import accelerate, transformers, datasets  # noqa: F401  (must precede the stubs below)
    from transformers import Trainer, TrainingArguments  # noqa: F401
    sys.path.insert(0, REF_SRC)
    # deepspeed / trl are not installed; the hot-path functions never touch them.
    if "deepspeed" not in sys.modules:
        sys.modules["deepspeed"] = types.ModuleType("deepspeed")
    if "trl" not in sys.modules:
        trl = types.ModuleType("trl")
        trlt = types.ModuleType("trl.trainer")
        trlu = types.ModuleType("trl.trainer.utils")
        trlu.disable_dropout_in_model = lambda m: None
        trlu.peft_module_casting_to_bf16 = lambda m: None
        trlu.trl_sanitze_kwargs_for_tagging = lambda **k: k
        sys.modules.update({"trl": trl, "trl.trainer": trlt, "trl.trainer.utils": trlu})
    import modeling, rankpo_trainer, data_utils
    # the stub has no __spec__; drop it again so accelerate's find_spec("deepspeed") probes keep working
    if getattr(sys.modules.get("deepspeed"), "__spec__", None) is None:
        sys.modules.pop("deepspeed", None)
    return modeling, rankpo_trainer, data_utils


def make_ref_model(modeling, tmpdir, arch="llama", **kw):
    """A tiny random-init HF model on disk so that the reference's own constructor runs unmodified."""
    import torch
    from transformers import LlamaConfig, LlamaModel, BertConfig, BertModel
    torch.manual_seed(0)
    if arch == "llama":
        cfg = LlamaConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                          num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=64,
                          rms_norm_eps=1e-5, rope_theta=10000.0, pad_token_id=0,
                          attention_bias=False, tie_word_embeddings=False)
        hf = LlamaModel(cfg)
    else:
        cfg = BertConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                         num_attention_heads=4, max_position_embeddings=64, pad_token_id=0,
                         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        hf = BertModel(cfg)
    path = os.path.join(tmpdir, arch)
    hf.save_pretrained(path)
    m = modeling.ModelForTraining(path, attn_implementation="eager", **kw)
    return m, hf, cfg


# ----------------------------------------------------------------------------------------------
def gen_contrastive(modeling, out, tmpdir):
    """G1 in-batch, G2 no-in-batch, G3 eval: modeling.py:281-322 with embed bypassed."""
    import torch
    cases = {}
    for d, dtype in [(64, "fp32"), (384, "fp32"), (2048, "fp32"), (64, "bf16"), (2048, "bf16")]:
        tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
        seed = 1000 + d
        qn = unit(seeded(seed, 8, d))
        pn = unit(seeded(seed + 1, 48, d))
        # make the positives correlated with their query so the loss is not ~ln P
        pn[::6] = unit(pn[::6] + (2.0 / np.sqrt(d)) * qn)
        for mode in ("inbatch", "noinbatch", "eval"):
            m, _, _ = make_ref_model(modeling, tmpdir, "llama", temperature=0.02,
                                     use_inbatch_neg=(mode != "noinbatch"))
            m.embed = lambda x: x
            q = torch.tensor(qn, dtype=torch.float32).to(tdt).requires_grad_(True)
            p = torch.tensor(pn, dtype=torch.float32).to(tdt).requires_grad_(True)
            if mode == "eval":
                m.eval()
                o = m(query=q, passage=p)
                rec = dict(scores=o.scores.detach().float().numpy())
            else:
                m.train()
                o = m(query=q, passage=p)
                o.loss.backward()
                rec = dict(scores=o.scores.detach().float().numpy(), loss=np.float64(o.loss.item()),
                           dq=q.grad.float().numpy(), dp=p.grad.float().numpy())
            if d == 2048:  # keep the fixture small: store random projections of the big grads
                R = seeded(77, d, 8)
                for k in ("dq", "dp"):
                    if k in rec:
                        rec[k + "_proj"] = rec.pop(k).astype(np.float64) @ R
            for k, v in rec.items():
                cases[f"{mode}_d{d}_{dtype}_{k}"] = v
    meta = dict(temperature=0.02, Q=8, P=48, seed_rule="q: RandomState(1000+d).randn(8,d); p: RandomState(1001+d).randn(48,d); "
                "both row-normalised; p[::6] = unit(p[::6] + (2/sqrt(d)) q); d=2048 grads stored as grad @ RandomState(77).randn(d,8)")
    np.savez_compressed(os.path.join(out, "contrastive.npz"), meta=json.dumps(meta), **cases)


def gen_pooling(modeling, out, tmpdir):
    """G4: ModelForTraining.embed pooling/normalize branch (modeling.py:219-238) on a stub encoder."""
    import torch
    N, L, d = 6, 8, 16
    h_np = seeded(5, N, L, d)
    masks = {
        "allones": np.ones((N, L), dtype=np.int64),
        "rightpad": np.array([[1] * k + [0] * (L - k) for k in (8, 5, 1, 3, 7, 2)], dtype=np.int64),
        "leftpad": np.array([[0] * (L - k) + [1] * k for k in (8, 5, 1, 3, 7, 2)], dtype=np.int64),
        "mixed": np.array([[1, 1, 0, 1, 1, 0, 0, 0], [0, 1, 1, 1, 1, 1, 1, 1], [1, 0, 0, 0, 0, 0, 0, 0],
                           [1, 1, 1, 1, 1, 1, 1, 0], [0, 0, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 0, 1, 1, 1]],
                          dtype=np.int64),
    }
    g_np = seeded(6, N, d)
    rec = dict(h=h_np, g=g_np)
    for arch in ("llama", "bert"):
        for normalize in (True, False):
            m, _, _ = make_ref_model(modeling, tmpdir, arch, temperature=0.02, normalize_embeddings=normalize)

            class Stub(torch.nn.Module):
                def forward(self, hidden=None, attention_mask=None, return_dict=True):
                    return SimpleNamespace(last_hidden_state=hidden)
            m.model = Stub()
            for name, mk in masks.items():
                h = torch.tensor(h_np, dtype=torch.float64, requires_grad=True)
                e = m.embed({"hidden": h, "attention_mask": torch.tensor(mk)})
                e.backward(torch.tensor(g_np, dtype=torch.float64))
                key = f"{arch}_{'norm' if normalize else 'raw'}_{name}"
                rec[key + "_embeds"] = e.detach().numpy()
                rec[key + "_dh"] = h.grad.numpy()
                rec["mask_" + name] = mk
    # zero-norm row (eps clamp path of F.normalize)
    m, _, _ = make_ref_model(modeling, tmpdir, "llama", temperature=0.02)
    m.model = Stub()
    hz = h_np.copy()
    hz[1, 4] = 0.0            # rightpad row 1 has length 5 -> pooled index 4
    hz[2, 0] = 1e-14          # tiny but non-zero row (length 1 -> index 0)
    h = torch.tensor(hz, dtype=torch.float64, requires_grad=True)
    e = m.embed({"hidden": h, "attention_mask": torch.tensor(masks["rightpad"])})
    e.backward(torch.tensor(g_np, dtype=torch.float64))
    rec["zeronorm_h"] = hz
    rec["zeronorm_embeds"] = e.detach().numpy()
    rec["zeronorm_dh"] = h.grad.numpy()
    np.savez_compressed(os.path.join(out, "pooling.npz"), **rec)


def _xdev_worker(rank, world, port, tmpdir, q_all, p_all, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, REF_SRC)
    import modeling
    m = modeling.ModelForTraining(os.path.join(tmpdir, "llama"), attn_implementation="eager",
                                  temperature=0.02, negatives_cross_device=True)
    m.embed = lambda x: x
    m.train()
    q = torch.tensor(q_all[rank], dtype=torch.float64, requires_grad=True)
    p = torch.tensor(p_all[rank], dtype=torch.float64, requires_grad=True)
    o = m(query=q, passage=p)
    o.loss.backward()
    ret[rank] = dict(loss=o.loss.item(), scores=o.scores.detach().numpy(), dq=q.grad.numpy(), dp=p.grad.numpy(),
                     q_reps=o.q_reps.detach().numpy(), p_reps=o.p_reps.detach().numpy())
    dist.destroy_process_group()


def gen_crossdevice(modeling, out, tmpdir):
    """G5: negatives_cross_device on W gloo processes (modeling.py:287-290, 331-377)."""
    import torch.multiprocessing as mp
    make_ref_model(modeling, tmpdir, "llama", temperature=0.02)  # writes tmpdir/llama
    rec = {}
    B, G, d = 4, 3, 32
    for world in (2, 4):
        q_all = [unit(seeded(300 + 10 * world + r, B, d)) for r in range(world)]
        p_all = [unit(seeded(400 + 10 * world + r, B * G, d)) for r in range(world)]
        for r in range(world):
            p_all[r][::G] = unit(p_all[r][::G] + 0.35 * q_all[r])
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_xdev_worker, args=(world, 29600 + world, tmpdir, q_all, p_all, ret), nprocs=world, join=True)
        for r in range(world):
            rec[f"w{world}_r{r}_q"] = q_all[r]
            rec[f"w{world}_r{r}_p"] = p_all[r]
            for k, v in ret[r].items():
                rec[f"w{world}_r{r}_{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(out, "crossdevice.npz"), meta=json.dumps(dict(temperature=0.02, B=B, G=G, d=d)), **rec)


def gen_rankpo(rankpo_trainer, out):
    """G6: concatenated_forward / rankpo_loss / get_batch_loss_metrics (rankpo_trainer.py:420-568)."""
    import torch
    T = rankpo_trainer.RankPOTrainer
    B, d = 8, 64
    qn = unit(seeded(21, B, d))
    pn = unit(seeded(22, 2 * B, d))
    pn[0::2] = unit(pn[0::2] + 0.4 * qn)      # chosen a bit closer
    pn[5] = unit(pn[5] + 1.0 * qn[2])          # one pair where rejected wins
    ref_c = 0.1 * seeded(23, B)
    ref_r = 0.1 * seeded(24, B)
    rec = dict(q=qn, p=pn, ref_chosen=ref_c, ref_rejected=ref_r)
    cases = []
    combos = itertools.product(("sigmoid", "hinge"), (0.0, 0.1), (True, False), (0.0, 0.5), (0.0, 0.5), (1.0, 0.0))
    for ci, (lt, ls, rf, sw, gbr, rw) in enumerate(combos):
        if rw == 0.0 and sw == 0.0:
            continue
        acc = SimpleNamespace(device=torch.device("cpu"), gather_for_metrics=lambda x: x)
        ns = SimpleNamespace(beta=2.0, gamma_beta_ratio=gbr, temperature=0.1, sft_weight=sw, rankpo_weight=rw,
                             label_smoothing=ls, loss_type=lt, reference_free=rf, accelerator=acc)
        ns.single_forward = lambda model, inputs: inputs
        ns.concatenated_forward = lambda model, batch: T.concatenated_forward(ns, model, batch)
        ns.rankpo_loss = lambda *a: T.rankpo_loss(ns, *a)
        with_ref = not rf

        class RefModel:
            pass
        if with_ref:
            ns.ref_model = RefModel()
            rs = torch.tensor(np.stack([ref_c, ref_r], 1), dtype=torch.float64)
            real_cf = ns.concatenated_forward
            ns.concatenated_forward = lambda model, batch: (rs if isinstance(model, RefModel) else real_cf(model, batch))
        else:
            ns.ref_model = None
        q = torch.tensor(qn, dtype=torch.float64, requires_grad=True)
        p = torch.tensor(pn, dtype=torch.float64, requires_grad=True)
        loss, metrics = T.get_batch_loss_metrics(ns, None, {"query": q, "passage": p}, "train")
        loss.backward()
        with torch.no_grad():
            sc = T.concatenated_forward(ns, None, {"query": q, "passage": p})
            losses = T.rankpo_loss(ns, sc[:, 0].clone(), sc[:, 1].clone(), rs[:, 0] if with_ref else 0, rs[:, 1] if with_ref else 0)
        name = f"c{ci}"
        cases.append(dict(name=name, loss_type=lt, label_smoothing=ls, reference_free=rf, sft_weight=sw,
                          gamma_beta_ratio=gbr, rankpo_weight=rw, beta=2.0, temperature=0.1, metrics=metrics,
                          loss=loss.item()))
        rec[name + "_scores"] = sc.numpy()
        rec[name + "_losses"] = losses.numpy()
        rec[name + "_dq"] = q.grad.numpy()
        rec[name + "_dp"] = p.grad.numpy()
    # analytic KAT: confirmed against the reference here
    ns = SimpleNamespace(beta=2.0, gamma_beta_ratio=0.0, temperature=0.1, label_smoothing=0.0, loss_type="sigmoid",
                         reference_free=True, accelerator=SimpleNamespace(device=torch.device("cpu")))
    c = torch.tensor([.8, .2, .5], dtype=torch.float64)
    r = torch.tensor([.3, .6, .5], dtype=torch.float64)
    rec["kat_sigmoid"] = T.rankpo_loss(ns, c.clone(), r.clone(), 0, 0).numpy()
    ns.loss_type = "hinge"
    rec["kat_hinge"] = T.rankpo_loss(ns, c.clone(), r.clone(), 0, 0).numpy()
    try:
        ns.loss_type = "bogus"
        T.rankpo_loss(ns, c.clone(), r.clone(), 0, 0)
        err = ""
    except ValueError as e:
        err = str(e)
    np.savez_compressed(os.path.join(out, "rankpo.npz"), meta=json.dumps(dict(cases=cases, bad_loss_type_error=err)), **rec)


def gen_rankpo_single_forward(rankpo_trainer, out):
    """single_forward: forced last-token pooling + forced normalize (rankpo_trainer.py:402-418)."""
    import torch
    T = rankpo_trainer.RankPOTrainer
    N, L, d = 5, 7, 12
    h_np = seeded(31, N, L, d)
    mk = np.array([[1] * k + [0] * (L - k) for k in (7, 3, 1, 5, 6)], dtype=np.int64)

    class Stub(torch.nn.Module):
        def forward(self, hidden=None, attention_mask=None, return_dict=True):
            return SimpleNamespace(last_hidden_state=hidden)
    h = torch.tensor(h_np, dtype=torch.float64)
    e = T.single_forward(SimpleNamespace(), Stub(), {"hidden": h, "attention_mask": torch.tensor(mk)})
    np.savez_compressed(os.path.join(out, "rankpo_single_forward.npz"), h=h_np, mask=mk, embeds=e.numpy())


def gen_collators(data_utils, out):
    """G7: the collators (data_utils.py:25-77, 181-214) incl. the docstring example (144-172)."""
    rec = {}
    col = data_utils.RankPODataCollatorWithPadding(pad_token_id=128004)
    examples = [
        {"query": {"input_ids": [1, 2, 3], "attention_mask": [1, 1, 1]},
         "chosen": {"input_ids": [4, 5], "attention_mask": [1, 1]},
         "rejected": {"input_ids": [6], "attention_mask": [1]}},
        {"query": {"input_ids": [7, 8], "attention_mask": [1, 1]},
         "chosen": {"input_ids": [9], "attention_mask": [1]},
         "rejected": {"input_ids": [10, 11, 12], "attention_mask": [1, 1, 1]}},
    ]
    o = col(examples)
    for a in ("query", "passage"):
        for b in ("input_ids", "attention_mask"):
            rec[f"rankpo_{a}_{b}"] = o[a][b].numpy()
    # contrastive collator: python `random` drives the sampling (data_utils.py:44,50)
    rs = np.random.RandomState(9)
    feats = []
    for i in range(4):
        npos, nneg = int(rs.randint(1, 4)), 7
        mk = lambda n: {"input_ids": [[int(x) for x in rs.randint(5, 100, size=int(rs.randint(1, 9)))] for _ in range(n)]}
        pos, neg = mk(npos), mk(nneg)
        for dct in (pos, neg):
            dct["attention_mask"] = [[1] * len(x) for x in dct["input_ids"]]
        qi = [int(x) for x in rs.randint(5, 100, size=int(rs.randint(1, 6)))]
        feats.append({"query": {"input_ids": qi, "attention_mask": [1] * len(qi)}, "positives": pos, "negatives": neg})
    random.seed(1234)
    o = data_utils.ContrastiveDataCollatorWithPadding(pad_token_id=3, num_negatives=5)(feats)
    for a in ("query", "passage"):
        for b in ("input_ids", "attention_mask"):
            rec[f"contrastive_{a}_{b}"] = o[a][b].numpy()
    np.savez_compressed(os.path.join(out, "collators.npz"),
                        meta=json.dumps(dict(rankpo_examples=examples, contrastive_features=feats,
                                             python_random_seed=1234, pad_token_id=3, num_negatives=5)), **rec)


def gen_end_to_end(modeling, out, tmpdir):
    """Tiny Llama / BERT through the reference's ModelForTraining (HF encoder + eager attention):
    pins encoder -> pool -> normalize -> scores -> loss end to end, incl. weights."""
    import torch
    rec = {}
    for arch in ("llama", "bert"):
        for inbatch in (True, False):
            m, hf, cfg = make_ref_model(modeling, tmpdir, arch, temperature=0.02, use_inbatch_neg=inbatch)
            m.train()
            rs = np.random.RandomState(55)
            B, G, Lq, Lp = 4, 3, 10, 16
            qlen = [10, 4, 7, 1]
            plen = [16, 3, 9, 12, 16, 5, 7, 8, 2, 11, 13, 6]
            qi = rs.randint(1, 128, size=(B, Lq)); pi = rs.randint(1, 128, size=(B * G, Lp))
            qm = np.array([[1] * k + [0] * (Lq - k) for k in qlen]); pm = np.array([[1] * k + [0] * (Lp - k) for k in plen])
            qi = qi * qm; pi = pi * pm   # pad id 0
            batch = {"query": {"input_ids": torch.tensor(qi), "attention_mask": torch.tensor(qm)},
                     "passage": {"input_ids": torch.tensor(pi), "attention_mask": torch.tensor(pm)}}
            o = m(**batch)
            o.loss.backward()
            key = f"{arch}_{'inbatch' if inbatch else 'noinbatch'}"
            rec[key + "_loss"] = np.float64(o.loss.item())
            rec[key + "_scores"] = o.scores.detach().numpy()
            rec[key + "_q_reps"] = o.q_reps.detach().numpy()
            rec[key + "_p_reps"] = o.p_reps.detach().numpy()
            gname = "embed_tokens.weight" if arch == "llama" else "embeddings.word_embeddings.weight"
            rec[key + "_grad_embed"] = dict(m.model.named_parameters())[gname].grad.numpy()
            if inbatch:
                for k, v in hf.state_dict().items():
                    rec[f"{arch}_w_{k}"] = v.numpy()
                rec[f"{arch}_config"] = json.dumps(cfg.to_dict(), default=str)
                rec.update({f"{arch}_q_ids": qi, f"{arch}_q_mask": qm, f"{arch}_p_ids": pi, f"{arch}_p_mask": pm})
    np.savez_compressed(os.path.join(out, "end_to_end.npz"), **rec)


def gen_metrics(out):
    """compute_metrics of the reference (utils.py:87-153; faiss itself is not installed and not needed for it)."""
    if "faiss" not in sys.modules:
        sys.modules["faiss"] = types.ModuleType("faiss")
    import utils as ref_utils
    rs = np.random.RandomState(3)
    nq, ncorpus, k = 12, 60, 20
    scores = np.sort(rs.rand(nq, k))[:, ::-1].copy()
    preds = np.stack([rs.permutation(ncorpus)[:k] for _ in range(nq)])
    labels = [sorted(set(int(x) for x in rs.choice(ncorpus, size=int(rs.randint(1, 6)), replace=False)) | {int(preds[i, int(rs.randint(0, k))])})
              for i in range(nq)]
    m = ref_utils.compute_metrics(preds, scores, labels, cutoffs=[1, 5, 10, 20])
    np.savez_compressed(os.path.join(out, "metrics.npz"), preds=preds, scores=scores,
                        meta=json.dumps(dict(labels=labels, cutoffs=[1, 5, 10, 20], metrics={k_: float(v) for k_, v in m.items()})))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    out = os.path.abspath(a.out)
    os.makedirs(out, exist_ok=True)
    modeling, rankpo_trainer, data_utils = import_reference()
    only = set(a.only.split(",")) if a.only else None
    with tempfile.TemporaryDirectory() as tmpdir:
        steps = dict(
            contrastive=lambda: gen_contrastive(modeling, out, tmpdir),
            pooling=lambda: gen_pooling(modeling, out, tmpdir),
            crossdevice=lambda: gen_crossdevice(modeling, out, tmpdir),
            rankpo=lambda: (gen_rankpo(rankpo_trainer, out), gen_rankpo_single_forward(rankpo_trainer, out)),
            collators=lambda: gen_collators(data_utils, out),
            end_to_end=lambda: gen_end_to_end(modeling, out, tmpdir),
            metrics=lambda: gen_metrics(out),
        )
        for name, fn in steps.items():
            if only and name not in only:
                continue
            fn()
            print("wrote", name)


if __name__ == "__main__":
    main()
