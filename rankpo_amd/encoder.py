"""From-scratch transformer encoders that run under PyTorch-ROCm (the north star keeps the encoder there; the
hand-written HIP starts at the pooled embedding).  They stand where the reference calls HF `AutoModel`
(modeling.py:175-178, 219; rankpo_trainer.py:402; run_rankpo.py:120): same call signature
`model(input_ids=..., attention_mask=..., return_dict=True).last_hidden_state`, same `config.architectures`,
same parameter names as HF `LlamaModel` / `BertModel`, so checkpoints in HF safetensors layout load and save.

MI355X notes
  * Llama + a right-padded MASK (what both reference collators produce, data_utils.py:64-70, 205-212): a causal model
    never lets a real token see the pad tokens behind it, so the padding mask is dropped and attention runs as
    pure causal flash attention; only pad positions (never pooled) differ from a masked run.  The decision is taken
    from the mask's content (`LlamaEncoder._mask`), never from a config flag: any other mask (left padding, holes)
    runs with HF's `causal & keep` mask.
  * `pooled_last_token`: tokens packed (no pad tokens), variable-length causal flash attention, the LAST block computes
    only K/V for all tokens and everything else for the pooled rows, final RMSNorm on the pooled rows only.
"""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn
from torch.utils.checkpoint import checkpoint

from . import ops as _ops


class EncoderConfig(SimpleNamespace):
    """Minimal stand-in for a HF PretrainedConfig (attribute access + to_dict)."""

    def to_dict(self):
        return dict(self.__dict__)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))


def llama_config(**kw) -> EncoderConfig:
    d = dict(architectures=["LlamaModel"], model_type="llama", vocab_size=128256, hidden_size=2048,
             intermediate_size=8192, num_hidden_layers=16, num_attention_heads=32, num_key_value_heads=8,
             head_dim=None, rms_norm_eps=1e-5, rope_theta=500000.0, rope_scaling=None,
             max_position_embeddings=131072, pad_token_id=None, attention_bias=False, mlp_bias=False,
             initializer_range=0.02, hidden_act="silu", padding_side="right", attention_dropout=0.0)
    d.update(kw)
    rp = d.pop("rope_parameters", None)
    if rp:
        # config.json as transformers >= 5 writes it: rope_theta and the scaling rule live in ONE `rope_parameters` dict
        # (4.45, the reference's pin, has `rope_theta` + `rope_scaling` at the top level; both are accepted).  Found by
        # tests/test_checkpoints.py: a 5.x-written checkpoint loaded with the default theta and gave wrong embeddings.
        if "rope_theta" in rp:
            d["rope_theta"] = float(rp["rope_theta"])
        if rp.get("rope_type", rp.get("type", "default")) != "default":
            d["rope_scaling"] = {k: v for k, v in rp.items() if k != "rope_theta"}
    if d["head_dim"] is None:
        d["head_dim"] = d["hidden_size"] // d["num_attention_heads"]
    return EncoderConfig(**d)


LLAMA3_ROPE = dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                   original_max_position_embeddings=8192)


def llama_3_2_1b_config(**kw):
    """Llama-3.2-1B architecture (BASELINE.json configs[1]); +7 special tokens as run_contrastive.py:132-142."""
    base = dict(vocab_size=128256 + 7, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
                num_attention_heads=32, num_key_value_heads=8, head_dim=64, rope_theta=500000.0,
                rope_scaling=dict(LLAMA3_ROPE), rms_norm_eps=1e-5, pad_token_id=128004)
    base.update(kw)
    return llama_config(**base)


def llama_3_8b_config(**kw):
    base = dict(vocab_size=128256 + 7, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                num_attention_heads=32, num_key_value_heads=8, head_dim=128, rope_theta=500000.0,
                rope_scaling=None, rms_norm_eps=1e-5, max_position_embeddings=8192, pad_token_id=128004)
    base.update(kw)
    return llama_config(**base)


def bert_config(**kw) -> EncoderConfig:
    d = dict(architectures=["BertModel"], model_type="bert", vocab_size=30522, hidden_size=384,
             intermediate_size=1536, num_hidden_layers=12, num_attention_heads=12, max_position_embeddings=512,
             type_vocab_size=2, layer_norm_eps=1e-12, pad_token_id=0, hidden_act="gelu", initializer_range=0.02,
             hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)     # HF BertConfig's defaults (the reference trains
    #          BGE / BGE-M3 with them: modeling.py:175-178 loads the checkpoint's config as it is, arguments.py has no override)
    d.update(kw)
    return EncoderConfig(**d)


def bge_small_config(**kw):
    """BAAI/bge-small-en architecture (BASELINE.json configs[0])."""
    return bert_config(**kw)


def xlm_roberta_config(**kw) -> EncoderConfig:
    """XLM-RoBERTa (BGE-M3's backbone, which the reference names next to BGE: README / modeling.py:231 CLS branch): the BERT
    block with RoBERTa's position ids (`BertEmbeddings.forward`), pad id 1, one token type, eps 1e-5."""
    d = dict(architectures=["XLMRobertaModel"], model_type="xlm-roberta", vocab_size=250002, hidden_size=1024,
             intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, max_position_embeddings=8194,
             type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1)
    d.update(kw)
    return bert_config(**d)


class EncoderOutput(dict):
    """`.last_hidden_state` + dict access, like transformers' ModelOutput."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


# ----------------------------------------------------------------------------------------------------
# Llama
# ----------------------------------------------------------------------------------------------------
def _rope_inv_freq(cfg) -> torch.Tensor:
    dim = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim))
    rs = getattr(cfg, "rope_scaling", None)
    kind = rs.get("rope_type", rs.get("type")) if rs else None
    if kind not in (None, "default", "llama3"):
        raise ValueError(f"rope scaling {kind!r} is not implemented (default and llama3 are: every Llama checkpoint the reference names)")
    if kind == "llama3":
        factor, lo, hi = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"]
        old = rs["original_max_position_embeddings"]
        wavelen = 2 * math.pi / inv
        low_wl, high_wl = old / lo, old / hi
        scaled = torch.where(wavelen > low_wl, inv / factor, inv)
        smooth = (old / wavelen - lo) / (hi - lo)
        mid = (1 - smooth) * inv / factor + smooth * inv
        is_mid = (wavelen <= low_wl) & (wavelen >= high_wl)
        inv = torch.where(is_mid, mid, scaled)
    return inv.to(torch.float32)


class RMSNorm(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x):
        return F.rms_norm(x, (x.shape[-1],), self.weight, self.eps)


def _add_norm(x, delta, norm: "RMSNorm"):
    """(x + delta, norm(x + delta)); one fused HIP pass on HIP tensors, plain PyTorch otherwise."""
    if _ops.fused_norm_ok(x):
        return _ops.add_rmsnorm(x, delta, norm.weight, norm.eps)
    if delta is not None:
        x = x + delta
    return x, norm(x)


def _rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


class RopeTables:
    """cos / sin of the rotary angles, f32 [P, head_dim/2] (P = positions or packed tokens)."""

    def __init__(self, freqs: torch.Tensor):
        self.cos32, self.sin32 = freqs.cos().contiguous(), freqs.sin().contiguous()

    def select(self, idx):
        out = RopeTables.__new__(RopeTables)
        out.cos32, out.sin32 = self.cos32.index_select(0, idx).contiguous(), self.sin32.index_select(0, idx).contiguous()
        return out

    def full(self, dtype, packed: bool):
        # HF LlamaRotaryEmbedding: emb = cat(freqs, freqs); cos(emb).to(dtype)
        cos = torch.cat((self.cos32, self.cos32), -1).to(dtype)
        sin = torch.cat((self.sin32, self.sin32), -1).to(dtype)
        return (cos[:, None, :], sin[:, None, :]) if packed else (cos[None, None], sin[None, None])


FOLD_ROPE = 2        # packed training path: rotary + attention as ONE autograd node; 2 = q rotated by the attention forward block
#                      that loads it + inverse rotary in the dQ / dK epilogues, 1 = the epilogues only, 0 = two nodes with
#                      separate rpo_rope passes both ways (the A/B arms of `bench.py --fold-rope`)
CKPT_SINGLE_INPUT = True  # a CHECKPOINTED block receives the residual stream as ONE tensor: x + delta is formed before the checkpoint
#                           boundary (one elementwise pass per checkpointed block) instead of inside the block's first add + RMSNorm, so
#                           that the checkpoint keeps one [tokens, d] tensor per block, not two: 50 GiB at cfg 5 (32 blocks x 206 848
#                           tokens x 4096 x 2 bytes), which the plan spends on un-checkpointed blocks.  (HF forms the sum in bf16 and
#                           normalises the rounded sum: this is the reference's own arithmetic; the fused kernel normalises the
#                           unrounded f32 sum.)  False: rounds 2-4 (the A/B arm of `bench.py --ckpt-inputs 2`)
FWD128_ONE_WAVE = True   # head_dim 128 with 4 (8, ..) q heads per kv head: the forward walks its own list (64 queries x 4 heads per entry)
#                          with the one-wave-per-SIMD kernel (ops.FWD_ONE_WAVE_HEAD_DIMS names the head dims); False: the 128-query kernel
#                          of rounds 3-4 (the A/B arm of `bench.py --fwd128 classic`)


def _checkpoint_contexts():
    """torch.utils.checkpoint's `context_fn`: (context of the forward, context of the recomputation) -- the latter tells the ops
    that what a block computes LAST is wasted there (`ops.recomputing`)."""
    import contextlib
    return contextlib.nullcontext(), _ops.recomputing()


class VarlenCtx:
    """cu_seqlens (int32, device), the host copy of the lengths and the longest length of a packed batch; the attention kernels'
    work lists: `tiles` (128 queries per entry: forward and dQ), `k_tiles` (dK/dV), `fwd_tiles` (the forward's own list where it has
    one -- head_dim 128: 64 queries x 4 q heads per entry, `ops.attn_fwd_tile_table` -- else None)."""

    def __init__(self, cu, lens, max_len, tiles=None, k_tiles=None, fwd_tiles=None):
        self.cu, self.lens, self.max_len, self.tiles, self.k_tiles, self.fwd_tiles = cu, lens, max_len, tiles, k_tiles, fwd_tiles


def _varlen_causal_attention(q, k, v, ctx: VarlenCtx):
    """q [T, nh, hd], k/v [T, nkv, hd] packed; causal attention inside each sequence."""
    if q.is_cuda and q.dtype == torch.bfloat16 and q.shape[-1] in (64, 128) and ctx.tiles is not None:
        # hand-written HIP kernels, head_dim 64 and 128: forward, and backward when the key-block table was built (grad mode);
        # without it PyTorch's flash-attention backward runs on the saved (out, padded lse)
        return _ops.flash_attn_varlen(q, k, v, ctx.cu, ctx.tiles, ctx.max_len, 1.0 / math.sqrt(q.shape[-1]),
                                      k_tiles=ctx.k_tiles,
                                      key_block=_ops.ATTN_KEY_BLOCK if q.shape[-1] == 64 else _ops.ATTN_KEY_BLOCK_HD128,
                                      fwd_tiles=ctx.fwd_tiles)
    if q.is_cuda and q.dtype in (torch.bfloat16, torch.float16):
        return torch.ops.aten._flash_attention_forward(q, k, v, ctx.cu, ctx.cu, ctx.max_len, ctx.max_len, 0.0, True,
                                                       False)[0]
    outs, o0 = [], 0                      # f32 or CPU (tests, config 1): one SDPA call per sequence
    for n in ctx.lens:
        a = F.scaled_dot_product_attention(q[o0:o0 + n].transpose(0, 1)[None], k[o0:o0 + n].transpose(0, 1)[None],
                                           v[o0:o0 + n].transpose(0, 1)[None], is_causal=True,
                                           enable_gqa=k.shape[1] != q.shape[1])
        outs.append(a[0].transpose(0, 1))
        o0 += n
    return torch.cat(outs, 0)


def _varlen_last_query_attention(q, k, v, ctx: VarlenCtx):
    """q [N, nh, hd]: ONE query per sequence (its last token); k/v [T, nkv, hd] packed.  The last token attends to
    every key of its own sequence, so no causal mask is needed."""
    N = q.shape[0]
    if q.is_cuda and q.dtype in (torch.bfloat16, torch.float16):
        cu_q = torch.arange(N + 1, device=q.device, dtype=torch.int32)
        return torch.ops.aten._flash_attention_forward(q, k, v, cu_q, ctx.cu, 1, ctx.max_len, 0.0, False, False)[0]
    outs, o0 = [], 0
    for i, n in enumerate(ctx.lens):
        a = F.scaled_dot_product_attention(q[i:i + 1].transpose(0, 1)[None], k[o0:o0 + n].transpose(0, 1)[None],
                                           v[o0:o0 + n].transpose(0, 1)[None], enable_gqa=k.shape[1] != q.shape[1])
        outs.append(a[0].transpose(0, 1))
        o0 += n
    return torch.cat(outs, 0)



def tag_fuse_group(*weights):
    """Marks weights whose row-concatenation is used as ONE GEMM operand.  A flat-buffer optimizer
    (distributed.FlatGradAllReducer / train_step.FlatAdamW) lays a tagged group out contiguously and in this order, which
    turns `fused_weight` into a zero-copy view."""
    key = object()
    for i, w in enumerate(weights):
        w._rpo_fuse_group = (key, i)


class _FusedWeight(torch.autograd.Function):
    """Row-concatenation of 2-D weights [n_i, k] -> [sum n_i, k].  When the weights already sit back to back in ONE
    storage (they do once FlatAdamW has moved the parameters into its flat buffer) the result is a view of that storage:
    no copy (torch.cat's copy kernel moved 80 MB per block at ~0.4 TB/s: 12 ms per cfg-2 step).  Backward hands every
    weight its row slice of the gradient."""

    @staticmethod
    def forward(ctx, *ws):
        ctx.rows = [w.shape[0] for w in ws]
        w0 = ws[0]
        adjacent = all(w.is_contiguous() and w.dim() == 2 and w.shape[1] == w0.shape[1] and w.dtype == w0.dtype
                       for w in ws)
        if adjacent:
            es = w0.element_size()
            for a, b in zip(ws[:-1], ws[1:]):
                if (a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
                        or a.data_ptr() + a.numel() * es != b.data_ptr()):
                    adjacent = False
                    break
        if adjacent:
            return w0.detach().as_strided((sum(ctx.rows), w0.shape[1]), (w0.shape[1], 1), w0.storage_offset())
        return torch.cat([w.detach() for w in ws], 0)

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.rows, 0))


def fused_weight(ws):
    return _FusedWeight.apply(*ws)


class LlamaAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.nh, self.nkv, self.hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        b = bool(cfg.attention_bias)
        self.q_proj = nn.Linear(cfg.hidden_size, self.nh * self.hd, bias=b)
        self.k_proj = nn.Linear(cfg.hidden_size, self.nkv * self.hd, bias=b)
        self.v_proj = nn.Linear(cfg.hidden_size, self.nkv * self.hd, bias=b)
        self.o_proj = nn.Linear(self.nh * self.hd, cfg.hidden_size, bias=b)
        tag_fuse_group(self.q_proj.weight, self.k_proj.weight, self.v_proj.weight)

    def _fused(self, x, mods):
        """ONE projection GEMM for several Linear modules that read the same input (q|k|v, k|v): better GEMM shapes
        than the narrow k / v projections alone and ONE input-gradient GEMM instead of a GEMM + add per branch.  The
        weights stay separate parameters (HF names); see `fused_weight`."""
        w = fused_weight([m.weight for m in mods])
        b = torch.cat([m.bias for m in mods], 0) if mods[0].bias is not None else None
        return _ops.linear(x, w, b)

    def forward(self, x, rope, attn_mask):
        N, L, _ = x.shape
        nq, nk = self.nh * self.hd, self.nkv * self.hd
        qkv = self._fused(x, (self.q_proj, self.k_proj, self.v_proj))            # [N, L, nq + 2 nk]
        fused = _ops.fused_encoder_ops_ok(x, self.hd)
        if (isinstance(attn_mask, VarlenCtx) and fused and attn_mask.k_tiles is not None and self.hd in (64, 128)
                and x.dtype == torch.bfloat16 and FOLD_ROPE):
            # training on packed tokens: rotary + attention as one autograd node -- the rotary pass runs in place on the
            # projection output, attention reads q / k / v as its column blocks, and the backward writes ONE d(q|k|v)
            # buffer (no split / cat copies) with the inverse rotation already applied in the dQ / dK epilogues
            o = _ops.rope_flash_attn_varlen_qkv(qkv, rope.cos32, rope.sin32, self.nh, self.nkv, attn_mask.cu,
                                                attn_mask.tiles, attn_mask.k_tiles, 1.0 / math.sqrt(self.hd), head_dim=self.hd,
                                                fold_forward=FOLD_ROPE >= 2, fwd_tiles=attn_mask.fwd_tiles)
            return _ops.linear(o.reshape(1, L, self.nh * self.hd), self.o_proj.weight, self.o_proj.bias)
        if (isinstance(attn_mask, VarlenCtx) and fused and attn_mask.tiles is not None and self.hd in (64, 128)
                and x.dtype == torch.bfloat16 and FOLD_ROPE and not torch.is_grad_enabled()):
            # no-grad callers (ModelForInference.encode, the RankPO ref_model under inference_mode, eval-mode scoring): the same
            # rotary fold and forward kernel, without a key-block table, an autograd node or row statistics
            o = _ops.rope_flash_attn_varlen_qkv_fwd(qkv, rope.cos32, rope.sin32, self.nh, self.nkv, attn_mask.cu, attn_mask.tiles,
                                                    1.0 / math.sqrt(self.hd), head_dim=self.hd, fwd_tiles=attn_mask.fwd_tiles)
            return F.linear(o.reshape(1, L, self.nh * self.hd), self.o_proj.weight, self.o_proj.bias)
        if fused:       # one in-place HIP pass over the q and k heads instead of neg / cat / 2 mul / add per tensor
            qkv = _ops.rope_(qkv, rope.cos32, rope.sin32, self.nh + self.nkv, self.hd, grad_inplace=True)
        q, k, v = qkv.split([nq, nk, nk], dim=-1)
        if (isinstance(attn_mask, VarlenCtx) and fused and attn_mask.k_tiles is not None and self.hd in (64, 128)
                and x.dtype == torch.bfloat16):
            # A/B arm (FOLD_ROPE = False): the rotary pass and the attention as two autograd nodes
            o = _ops.flash_attn_varlen_qkv(qkv.view(L, -1), self.nh, self.nkv, attn_mask.cu, attn_mask.tiles,
                                           attn_mask.k_tiles, 1.0 / math.sqrt(self.hd), head_dim=self.hd,
                                           fwd_tiles=attn_mask.fwd_tiles)
            return _ops.linear(o.reshape(1, L, self.nh * self.hd), self.o_proj.weight, self.o_proj.bias)
        if isinstance(attn_mask, VarlenCtx):
            # packed tokens [1, T, d]: variable-length causal flash attention, no pad tokens anywhere
            q, k, v = q.view(L, self.nh, self.hd), k.view(L, self.nkv, self.hd), v.view(L, self.nkv, self.hd)
            if not fused:
                cos, sin = rope.full(x.dtype, packed=True)
                q = q * cos + _rotate_half(q) * sin
                k = k * cos + _rotate_half(k) * sin
            o = _varlen_causal_attention(q, k, v, attn_mask)
            return _ops.linear(o.reshape(1, L, self.nh * self.hd), self.o_proj.weight, self.o_proj.bias)
        q = q.view(N, L, self.nh, self.hd).transpose(1, 2)
        k = k.view(N, L, self.nkv, self.hd).transpose(1, 2)
        v = v.view(N, L, self.nkv, self.hd).transpose(1, 2)
        if not fused:
            cos, sin = rope.full(x.dtype, packed=False)
            q = q * cos + _rotate_half(q) * sin
            k = k * cos + _rotate_half(k) * sin
        if attn_mask is None:
            o = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=self.nkv != self.nh)
        else:
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, enable_gqa=self.nkv != self.nh)
        return self.o_proj(o.transpose(1, 2).reshape(N, L, self.nh * self.hd))


class LlamaMLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        b = bool(getattr(cfg, "mlp_bias", False))
        self.gate_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=b)
        self.up_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=b)
        tag_fuse_group(self.gate_proj.weight, self.up_proj.weight)
        self.down_proj = nn.Linear(cfg.intermediate_size, cfg.hidden_size, bias=b)

    def forward(self, x, last_in_block: bool = False):
        """last_in_block: the caller's function ends with this MLP (see ops.swiglu_down)."""
        if self.down_proj.bias is None and self.gate_proj.bias is None and _ops.fused_encoder_ops_ok(x):
            # ONE gate|up projection GEMM; fused HIP silu*mul on its two halves, product not kept alive
            gu = _ops.linear(x, fused_weight([self.gate_proj.weight, self.up_proj.weight]))
            return _ops.swiglu_down(gu, self.down_proj.weight, last_in_block)
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class LlamaLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self_attn = LlamaAttention(cfg)
        self.mlp = LlamaMLP(cfg)
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)

    def forward(self, x, delta, rope, attn_mask):
        """Residual stream as a pair: the block receives the stream `x` and the not-yet-added output `delta` of the
        previous block (None for the first), and returns (stream, its own MLP output): every `x + delta` is fused
        into the RMSNorm that follows it."""
        x, h = _add_norm(x, delta, self.input_layernorm)
        x, h = _add_norm(x, self.self_attn(h, rope, attn_mask), self.post_attention_layernorm)
        return x, self.mlp(h, last_in_block=True)       # the block's last computation: `_run_layers` checkpoints ONE block per call

    def forward_last_rows(self, x, delta, rope, ctx, last_idx):
        """The LAST block of a last-token-pooled encoder: only the pooled rows are consumed downstream, so K and V
        are computed for every token but Q, the attention output, o_proj and the whole MLP only for the N pooled
        rows (saves ~1/num_layers of the GEMM and attention work, forward and backward).  x: packed [1, T, d];
        returns [N, d]."""
        att = self.self_attn
        x, h = _add_norm(x, delta, self.input_layernorm)
        T = h.shape[1]
        nk = att.nkv * att.hd
        kv = att._fused(h, (att.k_proj, att.v_proj))                         # [1, T, 2 nk]
        q = att.q_proj(h[0].index_select(0, last_idx))                       # [N, nh*hd]
        rope_last = rope.select(last_idx)
        if _ops.fused_encoder_ops_ok(x, att.hd):
            q = _ops.rope_(q, rope_last.cos32, rope_last.sin32, att.nh, att.hd)
            kv = _ops.rope_(kv, rope.cos32, rope.sin32, att.nkv, att.hd, grad_inplace=True)   # rotates the k heads only
            q = q.view(-1, att.nh, att.hd)
            if ctx.tiles is not None and _ops.last_query_attn_ok(q, kv, att.nh, att.nkv, att.hd):
                # hand-written one-query-per-sequence attention on the fused k|v buffer (ctx.tiles: hand_attention is on)
                o = _ops.last_query_attn(q, kv, ctx.cu, att.nkv, att.hd, 1.0 / math.sqrt(att.hd))
                xl = x[0].index_select(0, last_idx) + att.o_proj(o.reshape(-1, att.nh * att.hd))
                # last_in_block: only the residual add follows, and an add saves nothing for backward
                return xl + self.mlp(self.post_attention_layernorm(xl), last_in_block=True)
            k, v = kv.split([nk, nk], dim=-1)
            k = k.view(T, att.nkv, att.hd)
        else:
            k, v = kv.split([nk, nk], dim=-1)
            q, k = q.view(-1, att.nh, att.hd), k.view(T, att.nkv, att.hd)
            cq, sq = rope_last.full(x.dtype, packed=True)
            ck, sk = rope.full(x.dtype, packed=True)
            q = q * cq + _rotate_half(q) * sq
            k = k * ck + _rotate_half(k) * sk
        o = _varlen_last_query_attention(q, k, v.view(T, att.nkv, att.hd), ctx)
        xl = x[0].index_select(0, last_idx) + att.o_proj(o.reshape(-1, att.nh * att.hd))
        return xl + self.mlp(self.post_attention_layernorm(xl), last_in_block=True)


def _resized_embedding(old: nn.Embedding, n: int, std: float) -> nn.Embedding:
    if n == old.num_embeddings:
        return old
    new = nn.Embedding(n, old.embedding_dim, padding_idx=old.padding_idx, device=old.weight.device, dtype=old.weight.dtype)
    k = min(n, old.num_embeddings)
    with torch.no_grad():
        nn.init.normal_(new.weight, 0.0, std)
        if new.padding_idx is not None and new.padding_idx < n:
            new.weight[new.padding_idx].zero_()
        new.weight[:k] = old.weight[:k]
    new.weight.requires_grad_(old.weight.requires_grad)
    return new


class LlamaEncoder(nn.Module):
    """Decoder-only Llama stack without lm_head (== HF `LlamaModel`)."""

    def __init__(self, config: EncoderConfig):
        super().__init__()
        if float(getattr(config, "attention_dropout", 0.0) or 0.0) != 0.0:
            # HF LlamaModel's only dropout; every Llama checkpoint the reference names ships 0.0.  Not implemented in the
            # attention kernels: refuse loudly rather than train without a regularisation the config asks for.
            raise ValueError(f"attention_dropout={config.attention_dropout} is not supported by LlamaEncoder (only 0.0)")
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList(LlamaLayer(config) for _ in range(config.num_hidden_layers))
        self.norm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        # The rotary frequencies are NOT a module buffer: `module.to(torch.bfloat16)` casts floating-point buffers too, and a
        # frequency rounded to bf16 (8 significant bits) turns position 4096 by up to 2^-9 x 4096 = 8 rad the wrong way -- the
        # pooled embeddings of 4096-token rows then sit 0.8 % (cosine) off the float32 reference arithmetic and the weight gradients 43 %
        # (tests/test_gpu_fastpath.py::test_real_width_and_length_parity; short rows hide it: the error grows with the position).
        # HF computes inv_freq in float32 whatever the model dtype; so does this: a float32 tensor per device, never cast.
        self._inv_freq = {"cpu": _rope_inv_freq(config)}
        self.gradient_checkpointing = False
        self.checkpoint_layers = None      # None = all layers when gradient_checkpointing, "auto" = rankpo_amd.memory's plan, else the first k
        self.memory_plan = None            # the last EncoderPlan "auto" resolved to (tokens it was made for, blocks, modelled peak)
        self.memory_plan_hints = {}        # world= / partitioned= for the plan, set by whoever owns the optimizer (train_step)
        self.pack_fill = True              # packed path: round the token count up to a multiple of 256 with a filler sequence
        self.hand_attention = True         # False: PyTorch's own flash-attention ops both ways (the "stock flash" control of
        #                                    bench.step_parity; a Python attribute, not an environment switch)
        self.apply(self._init)

    def _init(self, m):
        std = self.config.initializer_range
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0.0, std)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, 0.0, std)

    def gradient_checkpointing_enable(self, layers="auto", **_):
        """The reference's `--gradient_checkpointing` (scripts/train/run_contrastive.sh:39; HF checkpoints every block).
        `layers`: "auto" (the default, what the HF Trainer's bare call gets): as FEW blocks as the HBM plan allows
        (rankpo_amd.memory.plan_for_encoder: measured free HBM, world size, the padded token count of the batches that arrive --
        288 GB lets Llama-3.2-1B keep every block at the BASELINE batch); None or "all": every block (HF's meaning); an int k: the
        first k blocks."""
        self.gradient_checkpointing = True
        self.checkpoint_layers = None if layers == "all" else layers
        self.memory_plan = None

    def _checkpointed_blocks(self, tokens: int) -> int:
        """How many of the first blocks run under torch.utils.checkpoint for a step of `tokens` padded tokens.  "auto" plans once
        per LARGEST token count seen (a longer batch re-plans: only ever towards more checkpointing, so the plan holds for every
        batch so far) and hands ops the verdict on the transposed d(gate|up) buffer."""
        if self.checkpoint_layers is None:
            return len(self.layers)
        if self.checkpoint_layers != "auto":
            return int(self.checkpoint_layers)
        if self.memory_plan is None or tokens > self.memory_plan.tokens:
            from . import memory
            plan = memory.plan_for_encoder(self, tokens, block_inputs=1 if CKPT_SINGLE_INPUT else 2, **self.memory_plan_hints)
            if self.memory_plan is not None:
                plan.checkpoint_blocks = max(plan.checkpoint_blocks, self.memory_plan.checkpoint_blocks)
            self.memory_plan = plan
            if plan.transposed_dgu_limit:
                _ops.SWIGLU_DGU_T_MAX_BYTES = max(_ops.SWIGLU_DGU_T_DEFAULT_MAX_BYTES, plan.transposed_dgu_limit)
        return self.memory_plan.checkpoint_blocks

    def gradient_checkpointing_disable(self):
        self.gradient_checkpointing = False

    def get_input_embeddings(self):
        return self.embed_tokens

    def resize_token_embeddings(self, n):
        """HF `PreTrainedModel.resize_token_embeddings` (run_contrastive.py:132-142: seven special tokens are added to the
        tokenizer, then `model.model.resize_token_embeddings(len(tokenizer))`): the first min(old, n) rows are kept bit for bit,
        new rows are drawn N(0, initializer_range) as transformers 4.45's `_init_weights` does, `config.vocab_size` follows.
        The table is a NEW parameter: build the optimizer (train_step.FlatAdamW) after the resize, as the reference does."""
        self.embed_tokens = _resized_embedding(self.embed_tokens, n, self.config.initializer_range)
        self.config.vocab_size = self.embed_tokens.num_embeddings
        return self.embed_tokens

    @property
    def inv_freq(self) -> torch.Tensor:
        """float32 rotary frequencies (host copy); see __init__ for why this is not a buffer."""
        return self._inv_freq["cpu"]

    def _rope(self, pos):
        # same op order as HF LlamaRotaryEmbedding: f32 outer product of positions and inverse frequencies
        key = str(pos.device)
        inv = self._inv_freq.get(key)
        if inv is None:
            inv = self._inv_freq[key] = self._inv_freq["cpu"].to(pos.device)
        return RopeTables(torch.outer(pos.to(torch.float32), inv))

    def _mask(self, attention_mask, L, dtype, right_padded=None):
        """None (pure causal attention, module docstring) when the MASK ITSELF is right-padded -- every row is ones followed
        by zeros -- else the boolean [N, 1, L, L] `causal & key-is-kept` mask HF builds from any attention_mask
        (modeling.py:219).  Decided from the mask, never from the config: a left-padded or holed batch through a
        default-config encoder must not attend to its pad tokens.  One host sync (this is the general padded path; the
        packed training path has its own) -- unless the caller already knows the answer: `right_padded` = what the packed
        path's own check of this very mask found (`last_right_padded`), so a batch that fell back from it is not
        synchronised on twice."""
        if attention_mask is None:
            return None
        m = attention_mask
        if right_padded is None:
            right_padded = bool((m[:, 1:].ne(0) <= m[:, :-1].ne(0)).all())
        if right_padded:
            return None
        keep = m.ne(0)[:, None, None, :]
        causal = torch.ones(L, L, dtype=torch.bool, device=m.device).tril()[None, None]
        # a pad token in front of the first real one has no key it may look at: it keeps itself (its row is never pooled
        # nor read as a key), so that no softmax row is empty whatever the attention backend does with those
        return (keep & causal) | torch.eye(L, dtype=torch.bool, device=m.device)[None, None]

    def hidden_states(self, input_ids, attention_mask=None, right_padded=None):
        """Output of the last block, BEFORE the final RMSNorm."""
        x, delta = self._stack(input_ids, attention_mask, right_padded)
        return x if delta is None else x + delta

    def _stack(self, input_ids, attention_mask, right_padded=None):
        x = self.embed_tokens(input_ids)
        N, L, _ = x.shape
        rope = self._rope(torch.arange(L, device=x.device))
        mask = self._mask(attention_mask, L, x.dtype, right_padded)
        return self._run_layers(x, rope, mask, tokens=N * L)

    def forward(self, input_ids=None, attention_mask=None, return_dict=True, right_padded=None, **_):
        x, delta = self._stack(input_ids, attention_mask, right_padded)
        h = _add_norm(x, delta, self.norm)[1]
        return EncoderOutput(last_hidden_state=h) if return_dict else (h,)

    def _run_layers(self, x, rope, ctx, upto=None, tokens=None):
        ck = self.gradient_checkpointing and self.training and torch.is_grad_enabled()
        nck = self._checkpointed_blocks(x.shape[0] * x.shape[1] if tokens is None else tokens) if ck else 0
        delta = None
        for i, layer in enumerate(self.layers if upto is None else self.layers[:upto]):
            if ck and i < nck:
                if CKPT_SINGLE_INPUT and delta is not None:
                    x, delta = x + delta, None
                x, delta = checkpoint(layer, x, delta, rope, ctx, use_reentrant=False, context_fn=_checkpoint_contexts)
            else:
                x, delta = layer(x, delta, rope, ctx)
        return x, delta

    def pooled_last_token(self, input_ids, attention_mask):
        """== forward(...).last_hidden_state[n, last real token] for RIGHT-padded 0/1 masks, computed without ever
        touching a pad token: tokens are packed to [T, d], attention is variable-length causal flash attention, and
        the final RMSNorm runs on the N pooled rows only.  Returns None when the mask is not a right-padded 0/1
        mask with at least one token per row (the caller then takes the general padded path).
        One host sync per call (sequence lengths), none per layer."""
        out = self.pooled_last_token_multi([(input_ids, attention_mask)])
        return None if out is None else out[0]

    def pooled_last_token_multi(self, batches):
        """`pooled_last_token` for several (input_ids, attention_mask) batches of different padded widths in ONE packed pass
        (e.g. the query and the passage batch of a training step: sequences are independent, so every pooled row is what the
        separate calls give, but the small batch no longer runs as its own set of under-filled GEMMs and launches).
        Returns a list of [N_i, d] tensors, or None if any batch is not right-padded 0/1 with at least one token per row."""
        dev = self.embed_tokens.weight.device
        # HOST batches for a model on the GPU (ModelForInference.encode hands over the tokenizer's CPU tensors): the checks, the
        # lengths and the packing are done on the host and ONLY the real tokens are uploaded -- no device sync at all, so the
        # batches of an encode() loop queue up behind each other on the stream
        on_host = dev.type == "cuda" and all(m.device.type == "cpu" and i.device.type == "cpu" for i, m in batches)
        stats = []
        for _, m in batches:
            lens_d = m.sum(-1)
            rp = (m[:, 1:].ne(0) <= m[:, :-1].ne(0)).all()                               # what `_mask` asks of a mask
            ok = (m[:, 1:] <= m[:, :-1]).all() & (lens_d > 0).all() & ((m == 0) | (m == 1)).all()
            stats.append(torch.cat([lens_d.to(torch.int64), ok.to(torch.int64)[None], rp.to(torch.int64)[None]]))
        info = torch.cat(stats).tolist()                                                   # the one sync (none for host batches)
        lens, o, oks = [], 0, []
        self.last_right_padded = []          # per batch; a caller that falls back to the padded path hands it to forward()
        for _, m in batches:
            n = m.shape[0]
            oks.append(bool(info[o + n]))
            self.last_right_padded.append(bool(info[o + n + 1]))
            lens += info[o:o + n]
            o += n + 2
        if not all(oks):
            return None
        ids_parts, pos_parts = [], []
        for ids, m in batches:
            L = m.shape[1]
            flat = torch.nonzero(m.reshape(-1).to(torch.bool), as_tuple=False).squeeze(1)  # size known: no surprise
            ids_parts.append(ids.reshape(-1)[flat])
            pos_parts.append(flat % L)
        # hipBLASLt's 256-row tiles: a packed token count that is not a multiple of 256 costs ~1.2 % of every GEMM (38.3 vs
        # 37.9 ms per block at 138 k tokens).  A filler sequence of < 256 pad tokens rounds it up; its pooled row is dropped
        # and, having no gradient, it contributes exact zeros to every weight gradient.
        n_fill = 0
        if (dev.type == "cuda" and self.embed_tokens.weight.dtype == torch.bfloat16 and sum(lens) >= 4096
                and self.pack_fill):
            n_fill = (-sum(lens)) % 256
        if n_fill:
            pad_id = self.config.pad_token_id if self.config.pad_token_id is not None else 0
            ids_parts.append(torch.full((n_fill,), pad_id, dtype=ids_parts[0].dtype, device=ids_parts[0].device))
            pos_parts.append(torch.arange(n_fill, dtype=pos_parts[0].dtype, device=pos_parts[0].device))
            lens = lens + [n_fill]
        ids = ids_parts[0] if len(ids_parts) == 1 else torch.cat(ids_parts)
        pos = pos_parts[0] if len(pos_parts) == 1 else torch.cat(pos_parts)
        if on_host:                                      # one upload: packed ids | positions (int32, pinned, asynchronous)
            both = torch.stack([ids.to(torch.int32), pos.to(torch.int32)]).pin_memory().to(dev, non_blocking=True)
            ids, pos = both[0], both[1]
        x = self.embed_tokens(ids)[None]                                                   # [1, T, d]
        rope = self._rope(pos)                                                             # per-token angles [T, hd/2]
        N = len(lens)
        cu = torch.zeros(N + 1, dtype=torch.int32, device=x.device)
        cu[1:] = torch.tensor(lens, dtype=torch.int64).cumsum(0).to(torch.int32).to(x.device, non_blocking=True)
        tiles = k_tiles = fwd_tiles = None
        if (x.is_cuda and self.config.head_dim in (64, 128) and x.dtype == torch.bfloat16
                and self.hand_attention):                                                        # hand-written flash attention
            tiles = _ops.attn_tile_table(lens, x.device, self.config.num_attention_heads, self.config.num_key_value_heads)
            if FWD128_ONE_WAVE:
                fwd_tiles = _ops.attn_fwd_tile_table(lens, x.device, self.config.num_attention_heads,
                                                     self.config.num_key_value_heads, self.config.head_dim)
            if torch.is_grad_enabled():
                k_tiles = _ops.attn_key_tile_table(
                    lens, x.device, self.config.num_key_value_heads,
                    _ops.ATTN_KEY_BLOCK if self.config.head_dim == 64 else _ops.ATTN_KEY_BLOCK_HD128)
        ctx = VarlenCtx(cu, lens, max(lens), tiles, k_tiles, fwd_tiles)
        last_idx = (cu[1:] - 1).to(torch.int64)
        # the plan's token count is the PADDED one (every row at its batch's full width): what the collator can hand over at worst
        tok_pad = sum(m.shape[0] * m.shape[1] for _, m in batches)
        x, delta = self._run_layers(x, rope, ctx, upto=len(self.layers) - 1, tokens=tok_pad)
        li = len(self.layers) - 1
        ck = self.gradient_checkpointing and self.training and torch.is_grad_enabled()
        nck = self._checkpointed_blocks(tok_pad) if ck else 0
        if ck and li < nck:
            if CKPT_SINGLE_INPUT and delta is not None:
                x, delta = x + delta, None
            last = checkpoint(self.layers[li].forward_last_rows, x, delta, rope, ctx, last_idx, use_reentrant=False,
                              context_fn=_checkpoint_contexts)
        else:
            last = self.layers[li].forward_last_rows(x, delta, rope, ctx, last_idx)         # [N, d]
        pooled = self.norm(last)
        if n_fill:
            pooled = pooled[:-1]
        return list(pooled.split([m.shape[0] for _, m in batches], 0))


# ----------------------------------------------------------------------------------------------------
# BERT (BGE / config 1)
# ----------------------------------------------------------------------------------------------------
class BertEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size, padding_idx=cfg.pad_token_id)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.dropout = nn.Dropout(cfg.hidden_dropout_prob)
        # RoBERTa family (XLM-R / BGE-M3): position ids count the NON-PAD tokens and start at padding_idx + 1; pad tokens
        # sit on padding_idx (HF create_position_ids_from_input_ids).  BERT: plain arange.
        self.roberta_positions = "Roberta" in cfg.architectures[0]
        self.pad_id = cfg.pad_token_id

    def forward(self, input_ids, token_type_ids=None):
        L = input_ids.shape[1]
        if self.roberta_positions:
            keep = input_ids.ne(self.pad_id).to(torch.int64)
            pos_emb = self.position_embeddings(torch.cumsum(keep, dim=1) * keep + self.pad_id)
        else:
            pos_emb = self.position_embeddings(torch.arange(L, device=input_ids.device))[None]
        tt = self.token_type_embeddings.weight[0] if token_type_ids is None else self.token_type_embeddings(token_type_ids)
        return self.dropout(self.LayerNorm(self.word_embeddings(input_ids) + tt + pos_emb))


class _BertSelf(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.nh = cfg.num_attention_heads
        self.query = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.key = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.value = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.dropout = nn.Dropout(cfg.attention_probs_dropout_prob)      # on the attention probabilities (HF BertSelfAttention)


class _BertSelfOutput(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.dropout = nn.Dropout(cfg.hidden_dropout_prob)


class BertAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self = _BertSelf(cfg)
        self.output = _BertSelfOutput(cfg)

    def forward(self, x, mask):
        N, L, D = x.shape
        nh = self.self.nh
        q = self.self.query(x).view(N, L, nh, D // nh).transpose(1, 2)
        k = self.self.key(x).view(N, L, nh, D // nh).transpose(1, 2)
        v = self.self.value(x).view(N, L, nh, D // nh).transpose(1, 2)
        # attention-probability dropout inside the fused attention op (HF's sdpa path: dropout_p = p if training else 0)
        pdrop = self.self.dropout.p if self.training else 0.0
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=pdrop).transpose(1, 2).reshape(N, L, D)
        return self.output.LayerNorm(self.output.dropout(self.output.dense(o)) + x)


class _Dense(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.dense = nn.Linear(i, o)


class _BertOutput(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.intermediate_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.dropout = nn.Dropout(cfg.hidden_dropout_prob)


class BertLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = BertAttention(cfg)
        self.intermediate = _Dense(cfg.hidden_size, cfg.intermediate_size)
        self.output = _BertOutput(cfg)

    def forward(self, x, mask):
        x = self.attention(x, mask)
        h = F.gelu(self.intermediate.dense(x))
        return self.output.LayerNorm(self.output.dropout(self.output.dense(h)) + x)


class _BertStack(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList(BertLayer(cfg) for _ in range(cfg.num_hidden_layers))


class BertEncoder(nn.Module):
    """== HF `BertModel` without the pooler (the reference only reads last_hidden_state[:, 0], modeling.py:232)."""

    def __init__(self, config: EncoderConfig):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = _BertStack(config)
        self.gradient_checkpointing = False
        self.apply(self._init)

    def _init(self, m):
        std = self.config.initializer_range
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0.0, std)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, 0.0, std)
            if m.padding_idx is not None:
                with torch.no_grad():
                    m.weight[m.padding_idx].zero_()

    def gradient_checkpointing_enable(self, **_):
        self.gradient_checkpointing = True

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def resize_token_embeddings(self, n):
        """See LlamaEncoder.resize_token_embeddings (the reference resizes its XLM-R / BERT encoders the same way)."""
        self.embeddings.word_embeddings = _resized_embedding(self.embeddings.word_embeddings, n, self.config.initializer_range)
        self.config.vocab_size = self.embeddings.word_embeddings.num_embeddings
        return self.embeddings.word_embeddings

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, return_dict=True, **_):
        x = self.embeddings(input_ids, token_type_ids)
        mask = None
        if attention_mask is not None:
            mask = attention_mask.to(torch.bool)[:, None, None, :]
        ck = self.gradient_checkpointing and self.training and torch.is_grad_enabled()
        for layer in self.encoder.layer:
            x = checkpoint(layer, x, mask, use_reentrant=False) if ck else layer(x, mask)
        return EncoderOutput(last_hidden_state=x) if return_dict else (x,)


def disable_dropout_in_model(model: nn.Module) -> None:
    """trl.trainer.utils.disable_dropout_in_model, which the reference calls when `disable_dropout` is set
    (rankpo_trainer.py:209-213, arguments.py:778-779): every nn.Dropout of the model gets p = 0."""
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0


# ----------------------------------------------------------------------------------------------------
# construction / HF-layout checkpoints
# ----------------------------------------------------------------------------------------------------
def build_encoder(config: EncoderConfig) -> nn.Module:
    arch = config.architectures[0]
    if "Llama" in arch:
        return LlamaEncoder(config)
    if "Bert" in arch or "Roberta" in arch:       # XLMRoberta* / Roberta*: the BERT block + RoBERTa position ids
        return BertEncoder(config)
    raise ValueError(f"unsupported architecture {arch!r} (Llama*, Bert* and (XLM)Roberta* encoders are implemented)")


SAFE_WEIGHTS, SAFE_INDEX = "model.safetensors", "model.safetensors.index.json"
BIN_WEIGHTS, BIN_INDEX = "pytorch_model.bin", "pytorch_model.bin.index.json"
# keys of *For<Task> heads and HF buffers the bare encoder has no use for (AutoModel.from_pretrained drops them too)
_IGNORED_KEYS = ("pooler.", "lm_head.", "cls.", "classifier.", "score.", "qa_outputs.")
_IGNORED_PARTS = ("position_ids", "token_type_ids", "inv_freq")


def _checkpoint_files(path: str):
    """The weight files of an HF checkpoint directory, in shard order, as (file name, keys it must hold or None, format).
    `AutoModel.from_pretrained` (modeling.py:175-178) reads, in this order of preference: one `model.safetensors`; a sharded
    `model.safetensors.index.json` (how `save_pretrained` writes anything above `max_shard_size`, 5 GB by default in
    transformers 4.45: Llama-3-8B ships as 4 shards); the same two forms as `pytorch_model.bin`."""
    for single, index, fmt in ((SAFE_WEIGHTS, SAFE_INDEX, "safetensors"), (BIN_WEIGHTS, BIN_INDEX, "bin")):
        if os.path.exists(os.path.join(path, single)):
            return [(single, None, fmt)]
        if os.path.exists(os.path.join(path, index)):
            with open(os.path.join(path, index)) as f:
                wm = json.load(f)["weight_map"]
            shards = {}
            for key, fn in wm.items():
                shards.setdefault(fn, []).append(key)
            absent = [fn for fn in shards if not os.path.exists(os.path.join(path, fn))]
            if absent:
                raise FileNotFoundError(f"checkpoint {path}: {index} names shard files that do not exist: {sorted(absent)}")
            return [(fn, shards[fn], fmt) for fn in sorted(shards)]
    raise FileNotFoundError(f"no {SAFE_WEIGHTS}, {SAFE_INDEX}, {BIN_WEIGHTS} or {BIN_INDEX} under {path}")


def _encoder_key(key: str, arch: str) -> Optional[str]:
    """Checkpoint key -> name in the bare encoder's state dict (None: a head / buffer the encoder does not have).  A checkpoint
    written from LlamaForCausalLM (how meta-llama ships) carries `model.` in front of every encoder key and `lm_head.weight`
    (absent when tied: Llama-3.2-1B); *ForMaskedLM / *ForSequenceClassification carry `bert.` / `roberta.`."""
    if key.startswith(_IGNORED_KEYS) or any(part in key for part in _IGNORED_PARTS):
        return None
    if "Llama" in arch and key.startswith("model."):
        return key[len("model."):]
    for prefix in ("bert.", "roberta."):
        if key.startswith(prefix):
            key = key[len(prefix):]
            return None if key.startswith(_IGNORED_KEYS) else key
    return key


def load_encoder(path: str, torch_dtype=None) -> nn.Module:
    """Load `config.json` + the weights of an HF-layout checkpoint directory: `model.safetensors`, or
    `model.safetensors.index.json` + its shards, or the `pytorch_model.bin` forms of the two (what
    `AutoModel.from_pretrained` accepts, modeling.py:175-178).  The encoder is allocated ONCE (in `torch_dtype` if given, as
    HF's `torch_dtype=` does; float32 otherwise, HF's default) and filled tensor by tensor, shard by shard -- safetensors
    files are memory-mapped and one tensor is materialised at a time, so a Llama-3-8B load peaks at the model's own size plus
    one tensor, not twice the checkpoint."""
    with open(os.path.join(path, "config.json")) as f:
        raw = json.load(f)
    default_arch = {"llama": "LlamaModel", "xlm-roberta": "XLMRobertaModel", "roberta": "RobertaModel"}
    arch = (raw.get("architectures") or [default_arch.get(raw.get("model_type"), "BertModel")])[0]
    raw["architectures"] = [arch]
    cfg = llama_config(**raw) if "Llama" in arch else bert_config(**raw)
    files = _checkpoint_files(path)
    with torch.device("meta"):
        enc = build_encoder(cfg)
    if torch_dtype is not None:
        enc = enc.to(torch_dtype)                              # on the meta device: no bytes move
    enc = enc.to_empty(device="cpu")
    target = enc.state_dict()                                  # name -> the parameter's / buffer's own storage
    loaded, unexpected = set(), []

    def put(key, get):
        name = _encoder_key(key, arch)
        if name is None:
            return
        if name not in target:
            unexpected.append(key)
            return
        t = get()
        if tuple(t.shape) != tuple(target[name].shape):
            raise RuntimeError(f"checkpoint {path}: size mismatch for {key}: checkpoint {tuple(t.shape)}, "
                               f"encoder {tuple(target[name].shape)} (config.json: vocab_size={getattr(cfg, 'vocab_size', None)})")
        with torch.no_grad():
            target[name].copy_(t)                              # casts to the encoder's dtype on the way
        loaded.add(name)

    for fn, expect, fmt in files:
        full = os.path.join(path, fn)
        if fmt == "safetensors":
            from safetensors import safe_open
            with safe_open(full, framework="pt", device="cpu") as f:
                keys = list(f.keys())
                for k in keys:
                    put(k, lambda k=k: f.get_tensor(k))
        else:
            try:
                sd = torch.load(full, map_location="cpu", weights_only=True, mmap=True)
            except RuntimeError:            # a legacy (non-zip) torch.save file cannot be memory-mapped; AutoModel.from_pretrained reads it
                sd = torch.load(full, map_location="cpu", weights_only=True)
            keys = list(sd)
            for k in keys:
                put(k, lambda k=k: sd[k])
            del sd
        if expect is not None and set(expect) - set(keys):
            raise RuntimeError(f"checkpoint {path}: the index places {sorted(set(expect) - set(keys))[:4]} ... in {fn}, which does not hold them")
    missing = [k for k in target if k not in loaded]
    if missing or unexpected:
        raise RuntimeError(f"checkpoint {path} does not match the encoder: missing={missing} unexpected={unexpected}")
    if "Llama" in arch:
        enc._inv_freq = {"cpu": _rope_inv_freq(cfg)}         # (built under the meta device above)
    return enc


def _parse_size(size) -> int:
    """'5GB' / '200KB' / '10MiB' / int bytes, as transformers' `max_shard_size` (decimal units for KB / MB / GB)."""
    if isinstance(size, (int, float)):
        return int(size)
    s = str(size).strip().upper()
    for unit, mult in (("KIB", 2 ** 10), ("MIB", 2 ** 20), ("GIB", 2 ** 30), ("KB", 10 ** 3), ("MB", 10 ** 6), ("GB", 10 ** 9)):
        if s.endswith(unit):
            return int(float(s[:-len(unit)]) * mult)
    return int(s)


def save_encoder(enc: nn.Module, path: str, max_shard_size="5GB"):
    """Write the INNER encoder in HF layout (what the reference's save_model does: contrastive_trainer.py:964-1027, which calls
    `self.model.model.save_pretrained(output_dir, safe_serialization=True)`): `config.json` (with the CURRENT vocab_size, i.e.
    after `resize_token_embeddings`, run_contrastive.py:132-142) + `model.safetensors`, or -- above `max_shard_size`, 5 GB as in
    transformers 4.45 -- `model-0000i-of-0000N.safetensors` + `model.safetensors.index.json` (greedy split in state-dict order,
    HF's rule).  Tensors go to the host one shard at a time: parameters may be views of ONE flat optimizer buffer
    (train_step.FlatAdamW), so each gets a private contiguous copy, and a Llama-3-8B save holds 5 GB of host copies, not 16."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    limit = _parse_size(max_shard_size)
    sd = enc.state_dict()
    shards, cur, cur_bytes = [], [], 0
    for k, v in sd.items():
        nb = v.numel() * v.element_size()
        if cur and cur_bytes + nb > limit:
            shards.append(cur)
            cur, cur_bytes = [], 0
        cur.append(k)
        cur_bytes += nb
    if cur:
        shards.append(cur)
    import re
    for fn in os.listdir(path):                                 # a previous save of another shard count must not shine through:
        # ONLY the names save_pretrained itself writes (a user's own `model_notes.safetensors` in the directory stays)
        if fn in (SAFE_INDEX, SAFE_WEIGHTS) or re.fullmatch(r"model-\d{5}-of-\d{5}\.safetensors", fn):
            os.remove(os.path.join(path, fn))
    names = [SAFE_WEIGHTS] if len(shards) == 1 else [f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors" for i in range(len(shards))]
    weight_map, total = {}, 0
    for fn, keys in zip(names, shards):
        part = {k: sd[k].detach().to("cpu", copy=True).contiguous() for k in keys}
        save_file(part, os.path.join(path, fn), metadata={"format": "pt"})
        for k in keys:
            weight_map[k] = fn
            total += part[k].numel() * part[k].element_size()
        del part
    if len(shards) > 1:
        with open(os.path.join(path, SAFE_INDEX), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2, sort_keys=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        cfgd = {k: v for k, v in enc.config.to_dict().items()}
        p0 = next(iter(enc.parameters()), None)
        if p0 is not None:
            cfgd["torch_dtype"] = str(p0.dtype).replace("torch.", "")
        json.dump(cfgd, f, indent=2, default=str)
