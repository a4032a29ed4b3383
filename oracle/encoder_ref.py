"""CPU oracle for the encoder + full contrastive / RankPO step  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Plain eager torch (CPU, float32 or float64; the SAME functions also run on a GPU in bf16 as the "stock reduced-precision
path" control of the tolerance tests: tensors follow the device of the weights) restatement of what the reference gets from HF `AutoModel`
(transformers `LlamaModel` / `BertModel`, eager attention; call sites modeling.py:175-178, 219;
rankpo_trainer.py:402).  The encoder arithmetic lives in a third-party dependency that is not under
/root/reference (transformers==4.45.2 pinned by the reference's requirements.txt); parity is therefore anchored on
  * the installed transformers' LlamaModel / BertModel (tests/test_encoder_parity.py, CPU), and
  * tests/golden/end_to_end.npz: the reference's own ModelForTraining run on tiny HF models, weights included.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Functional style on a dict of weights with HF parameter names, deliberately not sharing code with
rankpo_amd/encoder.py (the product).
"""
from __future__ import annotations

import math

import torch


def _rope_inv_freq(head_dim, theta, scaling=None):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float64) / head_dim))
    if scaling and scaling.get("rope_type", scaling.get("type")) == "llama3":
        f, lo, hi, old = (scaling["factor"], scaling["low_freq_factor"], scaling["high_freq_factor"],
                          scaling["original_max_position_embeddings"])
        out = []
        for v in inv.tolist():
            wl = 2 * math.pi / v
            if wl < old / hi:
                out.append(v)
            elif wl > old / lo:
                out.append(v / f)
            else:
                s = (old / wl - lo) / (hi - lo)
                out.append((1 - s) * v / f + s * v)
        inv = torch.tensor(out, dtype=torch.float64)
    return inv


_LOW = (torch.bfloat16, torch.float16)


def _rms(x, w, eps):
    # HF LlamaRMSNorm: statistics in float32, result cast back, then the weight
    if x.dtype in _LOW:
        x32 = x.float()
        return w * (x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def _softmax(s):
    # HF eager attention: softmax(dtype=float32).to(query dtype)
    return torch.softmax(s.float(), dim=-1).to(s.dtype) if s.dtype in _LOW else torch.softmax(s, dim=-1)


def llama_forward(w: dict, cfg: dict, input_ids: torch.Tensor, attention_mask: torch.Tensor, dtype=torch.float32,
                  block_checkpoint=False):
    """last_hidden_state [N, L, d] of a Llama stack with causal + key-padding masking (HF eager semantics).
    block_checkpoint: every block under torch.utils.checkpoint (non-reentrant) -- the SAME arithmetic, recomputed in backward
    instead of stored: eager attention keeps a [N, heads, L, L] probability tensor per block (12.9 GB in float32 for six
    4096-token rows of the Llama-3.2-1B shape), which is what lets bench.py run all 16 blocks of the headline model through this
    code on the device (tests/test_encoder_parity.py: results bit-identical with and without)."""
    W = lambda k: w[k].to(dtype)
    d, nh = cfg["hidden_size"], cfg["num_attention_heads"]
    nkv = cfg.get("num_key_value_heads") or nh
    hd = cfg.get("head_dim") or d // nh
    eps = cfg["rms_norm_eps"]
    N, L = input_ids.shape
    x = W("embed_tokens.weight")[input_ids]
    dev = x.device           # follows the weights: the same arithmetic runs on a GPU as the reduced-precision control
    inv = _rope_inv_freq(hd, cfg.get("rope_theta", 10000.0), cfg.get("rope_scaling"))
    ang = torch.outer(torch.arange(L, dtype=torch.float64), inv).to(torch.float32)
    ang = torch.cat([ang, ang], -1)
    cos, sin = ang.cos().to(dev, dtype), ang.sin().to(dev, dtype)
    rot = lambda t: torch.cat([-t[..., hd // 2:], t[..., : hd // 2]], -1)
    neg = torch.finfo(dtype).min
    allow = torch.ones(L, L, dtype=torch.bool, device=dev).tril()[None, None] & attention_mask.bool()[:, None, None, :]
    bias = torch.zeros(N, 1, L, L, dtype=dtype, device=dev).masked_fill(~allow, neg)
    def block(x, i):
        p = f"layers.{i}."
        h = _rms(x, W(p + "input_layernorm.weight"), eps)
        q = (h @ W(p + "self_attn.q_proj.weight").T).view(N, L, nh, hd).transpose(1, 2)
        k = (h @ W(p + "self_attn.k_proj.weight").T).view(N, L, nkv, hd).transpose(1, 2)
        v = (h @ W(p + "self_attn.v_proj.weight").T).view(N, L, nkv, hd).transpose(1, 2)
        q = q * cos + rot(q) * sin
        k = k * cos + rot(k) * sin
        k = k.repeat_interleave(nh // nkv, dim=1)
        v = v.repeat_interleave(nh // nkv, dim=1)
        att = _softmax(q @ k.transpose(-1, -2) / math.sqrt(hd) + bias)
        o = (att @ v).transpose(1, 2).reshape(N, L, nh * hd)
        x = x + o @ W(p + "self_attn.o_proj.weight").T
        h = _rms(x, W(p + "post_attention_layernorm.weight"), eps)
        g = h @ W(p + "mlp.gate_proj.weight").T
        u = h @ W(p + "mlp.up_proj.weight").T
        return x + (torch.nn.functional.silu(g) * u) @ W(p + "mlp.down_proj.weight").T

    for i in range(cfg["num_hidden_layers"]):
        if block_checkpoint and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint
            x = checkpoint(block, x, i, use_reentrant=False)
        else:
            x = block(x, i)
    return _rms(x, W("norm.weight"), eps)


def _ln(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def bert_forward(w: dict, cfg: dict, input_ids: torch.Tensor, attention_mask: torch.Tensor, dtype=torch.float32):
    W = lambda k: w[k].to(dtype)
    d, nh, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    hd = d // nh
    N, L = input_ids.shape
    arch = (cfg.get("architectures") or ["BertModel"])[0]
    if "Roberta" in arch:     # HF create_position_ids_from_input_ids: non-pad tokens count from padding_idx + 1
        pad = cfg["pad_token_id"]
        keep = (input_ids != pad).long()
        pos = W("embeddings.position_embeddings.weight")[torch.cumsum(keep, 1) * keep + pad]
    else:
        pos = W("embeddings.position_embeddings.weight")[:L][None]
    x = W("embeddings.word_embeddings.weight")[input_ids] + W("embeddings.token_type_embeddings.weight")[0] + pos
    x = _ln(x, W("embeddings.LayerNorm.weight"), W("embeddings.LayerNorm.bias"), eps)
    bias = torch.zeros(N, 1, 1, L, dtype=dtype, device=x.device).masked_fill(~attention_mask.bool()[:, None, None, :],
                                                                             torch.finfo(dtype).min)
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        lin = lambda name, t: t @ W(p + name + ".weight").T + W(p + name + ".bias")
        sp = lambda t: t.view(N, L, nh, hd).transpose(1, 2)
        q, k, v = sp(lin("attention.self.query", x)), sp(lin("attention.self.key", x)), sp(lin("attention.self.value", x))
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd) + bias, dim=-1)
        o = (att @ v).transpose(1, 2).reshape(N, L, d)
        x = _ln(lin("attention.output.dense", o) + x, W(p + "attention.output.LayerNorm.weight"),
                W(p + "attention.output.LayerNorm.bias"), eps)
        h = torch.nn.functional.gelu(lin("intermediate.dense", x))
        x = _ln(lin("output.dense", h) + x, W(p + "output.LayerNorm.weight"), W(p + "output.LayerNorm.bias"), eps)
    return x


def encoder_forward(w, cfg, input_ids, attention_mask, dtype=torch.float32, block_checkpoint=False):
    arch = (cfg.get("architectures") or ["Llama"])[0]
    if "Llama" in arch:
        return llama_forward(w, cfg, input_ids, attention_mask, dtype, block_checkpoint=block_checkpoint)
    return bert_forward(w, cfg, input_ids, attention_mask, dtype)


def embed(w, cfg, inputs, normalize=True, dtype=torch.float32, force_last=False, block_checkpoint=False):
    """ModelForTraining.embed (modeling.py:206-238) / RankPOTrainer.single_forward (rankpo_trainer.py:392-418)."""
    h = encoder_forward(w, cfg, inputs["input_ids"], inputs["attention_mask"], dtype, block_checkpoint=block_checkpoint)
    m = inputs["attention_mask"]
    arch = (cfg.get("architectures") or ["Llama"])[0]
    if force_last or "Llama" in arch:
        idx = (m.argmin(-1) - 1) % m.shape[-1]
        e = h[torch.arange(m.shape[0], device=h.device), idx]
    else:
        e = h[:, 0]
    if normalize:
        e = e / e.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return e


def contrastive_step(w, cfg, batch, temperature, use_inbatch_neg=True, normalize=True, dtype=torch.float32,
                     block_checkpoint=False):
    """Full ModelForTraining.forward training branch (modeling.py:278-314) -> (loss, scores, q, p) with autograd."""
    q = embed(w, cfg, batch["query"], normalize, dtype, block_checkpoint=block_checkpoint)
    p = embed(w, cfg, batch["passage"], normalize, dtype, block_checkpoint=block_checkpoint)
    Q = q.shape[0]
    G = p.shape[0] // Q
    if use_inbatch_neg:
        s = q @ p.T / temperature
        t = torch.arange(Q, device=q.device) * G
    else:
        s = torch.einsum("bd,bgd->bg", q, p.view(Q, G, -1)) / temperature
        t = torch.zeros(Q, dtype=torch.long, device=q.device)
    loss = (torch.logsumexp(s, -1) - s[torch.arange(Q, device=q.device), t]).mean()
    return loss, s, q, p


def state_dict_to_f32(module_or_sd):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return {k: v.detach().to("cpu", torch.float32) for k, v in sd.items()}
