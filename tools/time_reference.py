#!/usr/bin/env python3
"""the REFERENCE ITSELF timed on the build container's CPU cores, next to the oracle port on the same inputs.

Runs only where `/root/reference` exists (never on the GPU box).  It imports the reference's `src/modeling.py` unmodified
(as tools/make_golden.py does) and times
  * scoring only -- `ModelForTraining.forward` with `embed` bypassed, i.e. exactly modeling.py:281-314 + autograd backward -- at
    the five shapes of BASELINE.md §2, against the oracle port's scoring (the tail of oracle/encoder_ref.py:contrastive_step);
  * the full cfg-1 step (BASELINE.json configs[0]: BGE-small architecture, random init, B = 8, K = 5, Lq = 128, Lp = 256, f32,
    attn_implementation "sdpa") -- `model(query=..., passage=...)["loss"].backward()` -- against the oracle port's
    `contrastive_step` + backward on the SAME weights and the SAME batch (losses compared).
and writes profiles/ref_cpu_container.json.  `bench.py` reads `port_over_reference` from it for its `cpu_baseline` block: the
GPU box times the PORT (the reference cannot travel), and this file says what the port's time is worth in reference time.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/time_reference.py [--out profiles/ref_cpu_container.json] [--threads N]
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import tempfile
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

T = 0.02


def cpu_model():
    for ln in open("/proc/cpuinfo"):
        if ln.startswith("model name"):
            return ln.split(":", 1)[1].strip()
    return "unknown"


def timeit(fn, warmup, iters):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts


def scoring_cases(modeling, tmpdir, make_ref_model):
    """modeling.py:281-314 with `embed` bypassed vs the port's scoring, fwd + bwd."""
    m, _, _ = make_ref_model(modeling, tmpdir, "llama", temperature=T, use_inbatch_neg=True)
    m.embed = lambda x: x
    m.train()
    out = []
    for name, Q, P, d, dt in (("cfg 1 shape", 8, 48, 384, torch.float32), ("cfg 2 shape", 8, 48, 2048, torch.float32),
                              ("cfg 2 shape", 8, 48, 2048, torch.bfloat16), ("cfg 3 shape (W = 8 gathered)", 64, 384, 2048, torch.bfloat16),
                              ("cfg 5 shape (W = 8 gathered)", 64, 384, 4096, torch.bfloat16)):
        g = torch.Generator().manual_seed(1234)
        q = torch.nn.functional.normalize(torch.randn(Q, d, generator=g), dim=-1).to(dt).requires_grad_(True)
        p = torch.nn.functional.normalize(torch.randn(P, d, generator=g), dim=-1).to(dt).requires_grad_(True)
        G = P // Q
        losses = {}

        def ref_step():
            q.grad = p.grad = None
            o = m(query=q, passage=p)
            o.loss.backward()
            losses["reference"] = float(o.loss)

        def port_step():            # oracle/encoder_ref.py:contrastive_step after its two embed() calls
            q.grad = p.grad = None
            s = q @ p.T / T
            t = torch.arange(Q) * G
            loss = (torch.logsumexp(s, -1) - s[torch.arange(Q), t]).mean()
            loss.backward()
            losses["port"] = float(loss)
        tr = timeit(ref_step, 20, 200)
        tp = timeit(port_step, 20, 200)
        mr, mp_ = statistics.median(tr), statistics.median(tp)
        out.append({"case": name, "Q": Q, "P": P, "d": d, "dtype": str(dt).split(".")[-1], "reference_us": round(1e6 * mr, 1),
                    "port_us": round(1e6 * mp_, 1), "port_over_reference": round(mp_ / mr, 3),
                    "reference_scored_pairs_per_s": round(Q * P / mr), "loss_reference": losses["reference"], "loss_port": losses["port"],
                    "method": "median of 200 fwd+bwd after 20 warm-up"})
        print(out[-1], flush=True)
    return out


def full_step_cfg1(modeling, tmpdir, repeats):
    """BASELINE.json configs[0]: the reference's own ModelForTraining on a random-init BGE-small checkpoint vs the oracle port."""
    from transformers import BertConfig, BertModel
    from oracle import encoder_ref as E
    torch.manual_seed(0)
    hcfg = BertConfig(vocab_size=30522, hidden_size=384, intermediate_size=1536, num_hidden_layers=12, num_attention_heads=12,
                      max_position_embeddings=512, pad_token_id=0, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    path = os.path.join(tmpdir, "bge_small_arch")
    BertModel(hcfg).save_pretrained(path)
    ref = modeling.ModelForTraining(path, attn_implementation="sdpa", temperature=T, use_inbatch_neg=True,
                                    negatives_cross_device=False, normalize_embeddings=True)
    ref.train()
    B, K, Lq, Lp = 8, 5, 128, 256
    g = torch.Generator().manual_seed(1234)

    def side(N, L):
        ids = torch.randint(1000, 30522 - 1000, (N, L), generator=g)
        lens = torch.randint(L // 2, L + 1, (N,), generator=g)
        lens[0] = L
        mk = (torch.arange(L)[None, :] < lens[:, None]).long()
        return {"input_ids": ids * mk, "attention_mask": mk}
    batch = {"query": side(B, Lq), "passage": side(B * (1 + K), Lp)}
    sd = {k: v for k, v in ref.model.state_dict().items() if not k.startswith("pooler.")}
    w = {k: v.detach().clone().float().requires_grad_(True) for k, v in sd.items()}
    cd = dict(hcfg.to_dict(), architectures=["BertModel"])
    losses = {}

    def ref_step():
        ref.zero_grad(set_to_none=True)
        o = ref(**batch)
        o["loss"].backward()
        losses["reference"] = float(o["loss"])

    def port_step():
        for t in w.values():
            t.grad = None
        loss = E.contrastive_step(w, cd, batch, T)[0]
        loss.backward()
        losses["port"] = float(loss)
    tr = timeit(ref_step, 1, repeats)
    tp = timeit(port_step, 1, repeats)
    mr, mp_ = statistics.median(tr), statistics.median(tp)
    res = {"case": "cfg 1 full step: BGE-small architecture (12 blocks, d 384, 12 heads, ff 1536, vocab 30522), random init, f32, "
                   "B = 8, K = 5, Lq = 128, Lp = 256, right-padded rows of random length, T = 0.02, in-batch negatives, dropout 0 "
                   "(so that the two losses are comparable), attn_implementation sdpa",
           "trained_pairs": B * (1 + K), "reference_step_s": [round(t, 3) for t in tr], "port_step_s": [round(t, 3) for t in tp],
           "reference_median_s": round(mr, 3), "port_median_s": round(mp_, 3), "port_over_reference": round(mp_ / mr, 3),
           "reference_pairs_per_s": round(B * (1 + K) / mr, 3), "port_pairs_per_s": round(B * (1 + K) / mp_, 3),
           "loss_reference": losses["reference"], "loss_port": losses["port"],
           "method": f"median of {repeats} fwd+bwd after 1 warm-up, same weights, same batch"}
    assert abs(losses["reference"] - losses["port"]) < 2e-4 * max(1.0, abs(losses["reference"])), losses
    print(res, flush=True)
    return res


def full_step_llama(modeling, tmpdir, repeats):
    """The architecture of the HEADLINE workload (BASELINE.json configs[1]: Llama-3.2-1B, random init, f32 on the CPU) on the sample
    `bench.py`'s cpu_baseline times on the GPU box (8 queries x 20 tokens + 24 passages x 42 tokens, right-padded): the reference's own
    ModelForTraining (transformers' LlamaModel, sdpa) against the oracle port on the same weights and batch."""
    from transformers import LlamaConfig, LlamaModel
    from oracle import encoder_ref as E
    from rankpo_amd import encoder as PE
    torch.manual_seed(0)
    mine = PE.llama_3_2_1b_config()
    hcfg = LlamaConfig(vocab_size=mine.vocab_size, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
                       num_attention_heads=32, num_key_value_heads=8, head_dim=64, rms_norm_eps=1e-5, rope_theta=500000.0,
                       rope_scaling=dict(PE.LLAMA3_ROPE), max_position_embeddings=131072, pad_token_id=mine.pad_token_id,
                       attention_bias=False, tie_word_embeddings=False)
    path = os.path.join(tmpdir, "llama_3_2_1b_arch")
    LlamaModel(hcfg).save_pretrained(path)
    ref = modeling.ModelForTraining(path, attn_implementation="sdpa", temperature=T, use_inbatch_neg=True,
                                    negatives_cross_device=False, normalize_embeddings=True)
    ref.train()
    g = torch.Generator().manual_seed(7)

    def side(N, L):
        lens = torch.randint(L // 2, L + 1, (N,), generator=g)
        lens[0] = L
        mk = (torch.arange(L)[None, :] < lens[:, None]).long()
        ids = torch.randint(1000, mine.vocab_size - 1000, (N, L), generator=g)
        return {"input_ids": ids * mk + mine.pad_token_id * (1 - mk), "attention_mask": mk}
    batch = {"query": side(8, 20), "passage": side(24, 42)}
    w = {k: v.detach().clone().float().requires_grad_(True) for k, v in ref.model.state_dict().items()}
    cd = mine.to_dict()
    losses = {}

    def ref_step():
        ref.zero_grad(set_to_none=True)
        o = ref(**batch)
        o["loss"].backward()
        losses["reference"] = float(o["loss"])

    def port_step():
        for t in w.values():
            t.grad = None
        loss = E.contrastive_step(w, cd, batch, T)[0]
        loss.backward()
        losses["port"] = float(loss)
    tr = timeit(ref_step, 1, repeats)
    tp = timeit(port_step, 1, repeats)
    mr, mp_ = statistics.median(tr), statistics.median(tp)
    res = {"case": "Llama-3.2-1B architecture (16 blocks, d 2048, 32 / 8 heads, ff 8192, llama3 rope scaling, vocab 128263), random init, f32, "
                   "8 queries x 20 tok + 24 passages x 42 tok (bench.py's cpu_baseline sample), T = 0.02, in-batch negatives, sdpa",
           "tokens": 8 * 20 + 24 * 42, "reference_step_s": [round(t, 3) for t in tr], "port_step_s": [round(t, 3) for t in tp],
           "reference_median_s": round(mr, 3), "port_median_s": round(mp_, 3), "port_over_reference": round(mp_ / mr, 3),
           "loss_reference": losses["reference"], "loss_port": losses["port"],
           "method": f"median of {repeats} fwd+bwd after 1 warm-up, same weights, same batch"}
    assert abs(losses["reference"] - losses["port"]) < 5e-4 * max(1.0, abs(losses["reference"])), losses
    print(res, flush=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "ref_cpu_container.json"))
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--repeats", type=int, default=5)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from make_golden import import_reference, make_ref_model
    modeling, _, _ = import_reference()
    with tempfile.TemporaryDirectory() as tmp:
        sc = scoring_cases(modeling, tmp, make_ref_model)
        full = full_step_cfg1(modeling, tmp, args.repeats)
        llama = full_step_llama(modeling, tmp, max(2, args.repeats - 2))
    import transformers
    out = {"what": "the reference (yflyzhang/RankPO, /root/reference/src/modeling.py, imported unmodified) and the oracle port timed on the "
                   "same inputs in the build container (SURVEY.md §8d(1)); tools/time_reference.py",
           "host": {"cpu_model": cpu_model(), "threads": args.threads, "torch": torch.__version__, "transformers": transformers.__version__},
           "scoring_only": sc, "full_step_cfg1": full, "full_step_llama_3_2_1b_sample": llama,
           "port_over_reference": full["port_over_reference"], "port_over_reference_llama": llama["port_over_reference"],
           "port_over_reference_note": "time of the oracle port / time of the reference on the cfg-1 full step (encoder-bound, like the "
                                       "metric); > 1 means the port is SLOWER than the reference, so a cpu_baseline timed with the port "
                                       "UNDERSTATES the reference by that factor"}
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
