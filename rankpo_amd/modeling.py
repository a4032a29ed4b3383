"""Drop-in for the reference's `src/modeling.py` surface: `ModelOutput`, `ModelForTraining`
(`embed / compute_similarity / forward / distributed_gather`, modeling.py:116-404) and `ModelForInference.encode`
(modeling.py:411-554), with the scoring path executed by librankpo_hip.so (see rankpo_amd/ops.py).

Same constructor keywords, attribute names, `forward(query=..., passage=...)` keywords, return container and
error behaviour as the reference.  Additions (keyword-only, default off): `encoder=` / `config=` to build from an
in-memory encoder or an architecture config instead of a checkpoint directory (no network here), `torch_dtype=`.
"""
from __future__ import annotations

import logging
import os
from typing import Dict, List, Optional, Union

import numpy as np
import torch
import torch.distributed as dist
from torch import Tensor, nn

from . import ops
from .distributed import EmbeddingGather, FusedQPGather, all_gather_with_local_grad
from .encoder import build_encoder, load_encoder

logger = logging.getLogger(__name__)


class ModelOutput(dict):
    """`q_reps, p_reps, loss, scores` (modeling.py:17-22).  Like transformers' ModelOutput it is a dict with
    attribute access whose keys are the non-None fields; the trainer reads `outputs["loss"]`."""

    _fields = ("q_reps", "p_reps", "loss", "scores")

    def __init__(self, q_reps: Optional[Tensor] = None, p_reps: Optional[Tensor] = None,
                 loss: Optional[Tensor] = None, scores: Optional[Tensor] = None):
        super().__init__()
        for k, v in (("q_reps", q_reps), ("p_reps", p_reps), ("loss", loss), ("scores", scores)):
            object.__setattr__(self, k, v)
            if v is not None:
                self[k] = v

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


def _load_or_build(model_name_or_path, encoder, config, torch_dtype):
    if encoder is not None:
        return encoder
    if config is not None:
        enc = build_encoder(config)
        return enc.to(torch_dtype) if torch_dtype is not None else enc
    if model_name_or_path is None or not os.path.isdir(str(model_name_or_path)):
        raise ValueError(
            f"model_name_or_path={model_name_or_path!r} is not a local directory; hub downloads are not available "
            "here.  Pass a directory in HF layout (config.json + model.safetensors), or encoder= / config=.")
    return load_encoder(model_name_or_path, torch_dtype)


class _TwoGathers:
    def __init__(self, gq, gp):
        self.gq, self.gp = gq, gp

    def wait(self):
        return self.gq.wait(), self.gp.wait()


class ModelForTraining(nn.Module):
    """modeling.py:116-404."""

    def __init__(
        self,
        model_name_or_path: str = None,
        *,
        attn_implementation: str = None,
        use_cache: bool = True,
        cache_dir: str = None,
        token: str = None,
        trust_remote_code: bool = False,
        normalize_embeddings: bool = True,
        use_inbatch_neg: bool = True,
        negatives_cross_device: bool = False,
        temperature: float = 1.0,
        encoder: nn.Module = None,
        config=None,
        torch_dtype=None,
        unpad: bool = True,
    ):
        super().__init__()
        self.unpad = unpad
        # attn_implementation / use_cache / cache_dir / token / trust_remote_code are accepted for signature
        # compatibility; attention always runs through torch SDPA (flash on ROCm) and there is no KV cache.
        self.model = _load_or_build(model_name_or_path, encoder, config, torch_dtype)
        self.criterion = nn.CrossEntropyLoss(reduction="mean")     # kept for attribute parity (modeling.py:179)

        self.normalize_embeddings = normalize_embeddings
        self.temperature = temperature
        self.use_inbatch_neg = use_inbatch_neg
        self.config = self.model.config

        if not normalize_embeddings:                                # modeling.py:186-188
            self.temperature = 1.0
            logger.info("reset temperature = 1.0 due to using inner product to compute similarity")
        if normalize_embeddings:                                    # modeling.py:189-191
            if self.temperature > 0.5:
                raise ValueError("Temperature should be smaller than 1.0 when use cosine similarity (i.e., "
                                 "normalize_embeddings=True). Recommend to set it 0.01-0.1")

        self.negatives_cross_device = negatives_cross_device
        if self.negatives_cross_device:                             # modeling.py:194-201
            if not dist.is_initialized():
                raise ValueError("Distributed training has not been initialized for representation all gather.")
            self.process_rank = dist.get_rank()
            self.world_size = dist.get_world_size()

    def gradient_checkpointing_enable(self, **kwargs):
        self.model.gradient_checkpointing_enable(**kwargs)

    @property
    def pooling_mode(self) -> str:
        return "last" if "Llama" in self.config.architectures[0] else "cls"   # modeling.py:224 / 231

    def embed(self, inputs):
        """modeling.py:206-238: encoder -> last-token / CLS pooling -> (normalize) -> contiguous [N, d]."""
        if inputs is None:
            return None
        if self._one_pass_configured():
            # right-padded batches (what both collators produce): run the encoder on the real tokens only and
            # normalise the N pooled rows; identical to the padded path on every pooled row.
            pooled = self.model.pooled_last_token(inputs["input_ids"], inputs["attention_mask"])
            if pooled is not None:
                return ops.pool_normalize(pooled[:, None, :], None, "cls", self.normalize_embeddings)
            # fell back: the packed path's check already synchronised on this mask once; its verdict spares the padded path's own
            outputs = self.model(**inputs, return_dict=True, right_padded=self.model.last_right_padded[0])
        else:
            outputs = self.model(**inputs, return_dict=True)
        last_hidden_state = outputs.last_hidden_state
        attention_mask = inputs["attention_mask"]
        return ops.pool_normalize(last_hidden_state, attention_mask, self.pooling_mode, self.normalize_embeddings)

    def _one_pass_configured(self) -> bool:
        """Whether this model runs both towers as ONE packed encoder pass.  A property of the CONFIGURATION (constructor
        flag, pooling mode, encoder class), never of a batch: it decides the sequence of collectives under
        `negatives_cross_device`, which has to be the same on every rank whatever rows each rank happens to hold."""
        return self.unpad and self.pooling_mode == "last" and hasattr(self.model, "pooled_last_token_multi")

    def _embed_both(self, query, passage):
        """The normalised pooled rows of both batches as ONE [B + B G, d] block `q ‖ p` through ONE packed encoder pass, or
        None when the batches do not qualify for the unpadded path (sequences are independent, so the rows are the ones the
        two separate `embed` calls produce)."""
        if query is not None and passage is not None and self._one_pass_configured():
            pooled = self.model.pooled_last_token_multi([(query["input_ids"], query["attention_mask"]),
                                                        (passage["input_ids"], passage["attention_mask"])])
            if pooled is not None:
                return ops.pool_normalize(torch.cat(pooled, 0)[:, None, :], None, "cls", self.normalize_embeddings)
        return None

    def embed_pair(self, query, passage):
        """embed(query), embed(passage) -- through ONE packed encoder pass when both batches qualify for the unpadded path."""
        both = self._embed_both(query, passage)
        if both is not None:
            nq = query["input_ids"].shape[0]
            return both[:nq], both[nq:]             # row blocks of a contiguous matrix: contiguous themselves
        return self.embed(query), self.embed(passage)

    def compute_similarity(self, q_reps, p_reps):
        """modeling.py:240-252: `q @ p.transpose(-2, -1)`; 2-D inputs use the MFMA similarity kernel."""
        if q_reps.dim() == 2 and p_reps.dim() == 2 and q_reps.is_cuda and not (
                torch.is_grad_enabled() and (q_reps.requires_grad or p_reps.requires_grad)):
            return ops.similarity(q_reps, p_reps)
        return torch.matmul(q_reps, p_reps.transpose(-2, -1))

    def forward(self, query: Dict[str, Tensor] = None, passage: Dict[str, Tensor] = None):
        """modeling.py:254-328.  Keyword names `query` / `passage` are the collator's keys."""
        gather = None
        if self.training and self.negatives_cross_device and self.use_inbatch_neg and passage is not None:
            # modeling.py:287-290, MI355X-first: ONE all-gather of this rank's `q ‖ p` block instead of two collectives
            # (xGMI is point-to-point: 2 x (W - 1) messages of 32 / 192 KiB become W - 1 of 224 KiB), issued on RCCL's
            # stream as soon as the pooled rows exist and waited for just before the scoring kernel.
            # The SEQUENCE of collectives is decided by the configuration alone (`_one_pass_configured`), never by what this
            # rank's batch looks like: a rank whose batch falls off the packed path (an empty row, a left-padded or holed mask)
            # still contributes ONE [B + B G, d] block to the ONE all-gather its peers issue.
            if self._one_pass_configured():
                both = self._embed_both(query, passage)
                nq = query["input_ids"].shape[0]
                if both is not None:
                    q_reps, p_reps = both[:nq], both[nq:]
                else:                       # this rank's batch only: general padded path, same collective
                    q_reps, p_reps = self.embed(query), self.embed(passage)
                    both = torch.cat([q_reps.detach(), p_reps.detach()], 0)
                gather = FusedQPGather(both, nq)
            else:
                # towers one after the other (padded / CLS encoders): the passage gather is in flight during the query tower
                p_reps = self.embed(passage)
                gp = EmbeddingGather(p_reps)
                q_reps = self.embed(query)
                gq = EmbeddingGather(q_reps)
                gather = _TwoGathers(gq, gp)
        else:
            q_reps, p_reps = self.embed_pair(query, passage)

        if self.training:
            q_all = p_all = None
            q_row0 = p_row0 = 0
            if gather is not None:
                # rank-major order; only this rank's rows receive gradients (modeling.py:374-377)
                q_all, p_all = gather.wait()
                q_row0 = self.process_rank * q_reps.shape[0]
                p_row0 = self.process_rank * p_reps.shape[0]
            loss, scores = ops.infonce_loss(q_reps, p_reps, self.temperature, self.use_inbatch_neg,
                                            q_all=q_all, p_all=p_all, q_row0=q_row0, p_row0=p_row0)
            if q_all is not None:       # the reference returns the gathered representations (modeling.py:289-290, 326-327)
                q_reps, p_reps = q_all, p_all
        else:
            scores = self.compute_similarity(q_reps, p_reps)                  # modeling.py:321
            loss = None
        return ModelOutput(loss=loss, scores=scores, q_reps=q_reps, p_reps=p_reps)

    def distributed_gather(self, tensor: Optional[torch.Tensor], use_method: int = 1):
        """modeling.py:331-404: rank-major gather along dim 0; gradient flows to this rank's slice only.
        The three `use_method`s of the reference are mathematically identical; one implementation serves all."""
        if tensor.ndim == 0:
            tensor = tensor.clone()[None]
        if not tensor.is_contiguous():
            tensor = tensor.contiguous()
        if use_method not in (1, 2, 3):
            return None
        return all_gather_with_local_grad(tensor)


class ModelForInference(nn.Module):
    """modeling.py:411-554."""

    def __init__(
        self,
        model_name_or_path: str = None,
        attn_implementation: str = None,
        normalize_embeddings: bool = True,
        use_fp16: bool = False,
        use_bf16: bool = False,
        device: int = 0,
        *,
        encoder: nn.Module = None,
        config=None,
        tokenizer=None,
    ) -> None:
        super().__init__()
        if use_bf16 and use_fp16:                                     # modeling.py:443-444
            raise ValueError("Cannot use fp16 and bf16 in the same time!")
        if torch.cuda.is_available():
            self.device = torch.device(device)
        else:
            self.device = torch.device("cpu")
            use_fp16 = False
        torch_dtype = torch.float32
        if use_fp16:
            torch_dtype = torch.float16
        elif use_bf16:
            torch_dtype = torch.bfloat16
        self.model = _load_or_build(model_name_or_path, encoder, config, torch_dtype)
        if encoder is not None and (use_fp16 or use_bf16):           # an in-memory encoder follows the flag like a checkpoint does
            self.model = self.model.to(torch_dtype)
        if tokenizer is None:
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(model_name_or_path)
        self.tokenizer = tokenizer
        self.normalize_embeddings = normalize_embeddings
        self.config = self.model.config
        if not getattr(self.tokenizer, "pad_token", None):            # modeling.py:467-468
            raise ValueError("pad_token is not specified!")
        self.model = self.model.to(self.device)

    @torch.inference_mode()
    def encode(
        self,
        sentences: Union[List[str], str],
        batch_size: int = 256,
        max_length: int = 512,
        convert_to_numpy: bool = True,
        description: str = "Encoding",
        bucket_by_length: bool = False,
    ) -> Union[np.ndarray, torch.Tensor]:
        """modeling.py:473-554.  Differences that do not change results: no per-batch `empty_cache()` (a device
        sync per batch, modeling.py:543-544); numpy conversion happens once at the end; batch i + 1 is tokenised while the GPU
        runs batch i (program order: nothing here waits for the device); right-padded Llama batches are packed on the HOST from
        the tokenizer's own CPU tensors and only the real tokens are uploaded, so that no batch waits for the one before it
        (`LlamaEncoder.pooled_last_token_multi`: no device sync).
        `bucket_by_length` (an addition, default off = the reference's batching): batches are formed over the sentences sorted by
        text length, longest first, and the rows are put back in input order at the end -- padded encoders (BERT / XLM-R, CLS
        pooling) then pad every batch to ITS longest row instead of the corpus' (the packed Llama path computes no pad token
        either way).  Same rows up to the round-off of other GEMM shapes."""
        self.model.eval()
        input_was_string = False
        if isinstance(sentences, str):
            sentences = [sentences]
            input_was_string = True
        if not isinstance(sentences[0], str):
            raise ValueError("Input items should be text.")
        mode = "last" if "Llama" in self.config.architectures[0] else "cls"
        all_embeddings = []
        order = None
        if bucket_by_length and len(sentences) > batch_size:
            order = sorted(range(len(sentences)), key=lambda i: -len(sentences[i]))
            sentences = [sentences[i] for i in order]
        starts = list(range(0, len(sentences), batch_size))

        def tokenise(i):
            return self.tokenizer(sentences[i:i + batch_size], padding=True, truncation=True, max_length=max_length,
                                  return_tensors="pt")

        # Batch n + 1 is tokenised AFTER batch n's kernels have been queued and WHILE the GPU runs them: nothing in this loop waits
        # for the device (host-side packing, asynchronous upload), so plain program order gives the overlap.  (A worker thread that
        # tokenised during the launches measured worse: the tokenizer's thread pool took the cores from the launching thread and the
        # GPU idled -- 1280-token queries: 205 sentences/s against 500 pre-tokenised, profiles/r06f_bench_encode.json.)
        inputs = tokenise(starts[0]) if starts else None
        for n, i in enumerate(starts):
            pooled = None
            packed_tried = mode == "last" and hasattr(self.model, "pooled_last_token")
            if packed_tried:
                # right-padded batches (the tokenizer's default): packed tokens, no pad token is ever computed, the last
                # block runs for the pooled rows only; None for any other mask.  The tensors are still on the host here.
                pooled = self.model.pooled_last_token(inputs["input_ids"], inputs["attention_mask"])
            if pooled is not None:
                emb = ops.pool_normalize(pooled[:, None, :], None, "cls", self.normalize_embeddings)
            else:
                hint = {"right_padded": self.model.last_right_padded[0]} if packed_tried else {}
                inputs = {k: v.to(self.device) for k, v in inputs.items()}
                h = self.model(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"],
                               return_dict=True, **hint).last_hidden_state
                emb = ops.pool_normalize(h, inputs["attention_mask"], mode, self.normalize_embeddings)
            all_embeddings.append(emb)
            if n + 1 < len(starts):
                inputs = tokenise(starts[n + 1])                    # the GPU is busy with batch n meanwhile
        out = torch.cat(all_embeddings, dim=0)
        if order is not None:                       # back to the caller's order
            inv = torch.empty(len(order), dtype=torch.int64)
            inv[torch.tensor(order)] = torch.arange(len(order))
            out = out.index_select(0, inv.to(out.device))
        if convert_to_numpy:
            if out.dtype == torch.bfloat16:      # modeling.py:537-538
                out = out.float()
            out = out.cpu().numpy()
        if input_was_string:
            return out[0]
        return out
