"""Drop-in for the hot-path methods of the reference's `RankPOTrainer` (src/rankpo_trainer.py:392-587, 626-645):
`single_forward, concatenated_forward, rankpo_loss, get_batch_loss_metrics, compute_loss, store_metrics, log`.

The HF/TRL plumbing of the reference class (PEFT, DeepSpeed ref-model wrapping, wandb, push_to_hub;
rankpo_trainer.py:67-352) is out of scope; this class only needs a policy encoder, an optional
reference encoder and the loss knobs, whose names and meaning are those of `RankPOArguments`
(arguments.py:692-778).

MI355X-side differences that do not change results: the three scoring stages run in librankpo_hip.so
(rankpo_amd/ops.py); the 9 metrics are produced by the same kernel launch and fetched with ONE device->host copy
instead of 9 `gather_for_metrics(...).mean().item()` round trips (rankpo_trainer.py:496-520).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Any, Dict, List, Literal, Optional, Tuple, Union

import torch
import torch.distributed as dist
from torch import nn

from . import ops
from ._lib import METRIC_KEYS


class RankPOTrainer:
    def __init__(
        self,
        model: nn.Module = None,
        ref_model: Optional[nn.Module] = None,
        *,
        beta: float = 0.1,                  # arguments.py:737
        temperature: float = 1.0,           # :733
        gamma_beta_ratio: float = 0.0,      # :745
        sft_weight: float = 0.0,            # :754
        rankpo_weight: float = 1.0,         # :758
        loss_type: str = "sigmoid",         # :762
        label_smoothing: float = 0.0,       # :766
        reference_free: bool = False,       # :692
        disable_dropout: bool = True,       # :778
        args: Any = None,
    ):
        if args is not None:    # accept a RankPOArguments-like object
            beta, temperature = args.beta, args.temperature
            gamma_beta_ratio, sft_weight, rankpo_weight = args.gamma_beta_ratio, args.sft_weight, args.rankpo_weight
            loss_type, label_smoothing, reference_free = args.loss_type, args.label_smoothing, args.reference_free
            disable_dropout = getattr(args, "disable_dropout", disable_dropout)
        self.model = model
        self.ref_model = ref_model
        self.beta = beta
        self.temperature = temperature
        self.gamma_beta_ratio = gamma_beta_ratio
        self.sft_weight = sft_weight
        self.rankpo_weight = rankpo_weight
        self.loss_type = loss_type
        self.label_smoothing = label_smoothing
        self.reference_free = reference_free
        if disable_dropout:                 # rankpo_trainer.py:209-213
            from .encoder import disable_dropout_in_model
            if self.model is not None:
                disable_dropout_in_model(self.model)
            if self.ref_model is not None:
                disable_dropout_in_model(self.ref_model)
        if self.ref_model is not None:
            self.ref_model.eval()
            for p in self.ref_model.parameters():
                p.requires_grad_(False)
        self._stored_metrics = defaultdict(lambda: defaultdict(list))
        self._pending_metrics = None         # (prefix, device SUM of the metric vectors since the last log, count): O(1) memory

    # -- knobs -> C struct --------------------------------------------------------------------------
    def _cfg(self) -> ops.RankPOConfig:
        return ops.RankPOConfig(beta=self.beta, temperature=self.temperature, gamma_beta_ratio=self.gamma_beta_ratio,
                                label_smoothing=self.label_smoothing, rankpo_weight=self.rankpo_weight,
                                sft_weight=self.sft_weight, loss_type=self.loss_type,
                                reference_free=self.reference_free)

    # -- rankpo_trainer.py:392-418 --------------------------------------------------------------------
    def single_forward(self, model: nn.Module, inputs: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Encoder -> ALWAYS last-token pooling -> ALWAYS normalize (the reference ignores pooling/normalize flags)."""
        if hasattr(model, "pooled_last_token"):       # unpadded fast path (right-padded batches), same pooled rows
            pooled = model.pooled_last_token(inputs["input_ids"], inputs["attention_mask"])
            if pooled is not None:
                return ops.pool_normalize(pooled[:, None, :], None, "cls", True)
        outputs = model(**inputs, return_dict=True)
        return ops.pool_normalize(outputs.last_hidden_state, inputs["attention_mask"], "last", True)

    # -- rankpo_trainer.py:420-445 --------------------------------------------------------------------
    def concatenated_forward(self, model: nn.Module, batch: Dict[str, Any]) -> torch.Tensor:
        """scores[b, g] = <q_b, p_{2b+g}>, unscaled, [B, 2]."""
        q, p = self._embed_pair(model, batch)
        cfg = self._cfg()
        if cfg.loss_type not in ("sigmoid", "hinge"):
            cfg.loss_type = "sigmoid"        # scores do not depend on it
        _, scores, _, _ = ops.rankpo_loss_metrics(q, p, cfg)
        return scores.to(q.dtype)

    # -- rankpo_trainer.py:525-568 --------------------------------------------------------------------
    def rankpo_loss(self, chosen_scores, rejected_scores, ref_chosen_scores=None, ref_rejected_scores=None):
        """Per-sample losses from scores.  Tiny ([B]) elementwise math kept in torch for API parity; the training
        path (`get_batch_loss_metrics`) never calls it -- the fused kernel computes the same numbers."""
        adv = chosen_scores - rejected_scores
        if not self.reference_free:
            rc = 0 if ref_chosen_scores is None else ref_chosen_scores
            rr = 0 if ref_rejected_scores is None else ref_rejected_scores
            adv = adv - (rc - rr)
        adv = adv / self.temperature
        logits = adv - self.gamma_beta_ratio
        if self.loss_type == "sigmoid":
            F = torch.nn.functional
            return (-F.logsigmoid(self.beta * logits) * (1 - self.label_smoothing)
                    - F.logsigmoid(-self.beta * logits) * self.label_smoothing)
        if self.loss_type == "hinge":
            return torch.relu(1 - self.beta * logits)
        raise ValueError(f"Unknown loss type: {self.loss_type}. Should be one of ['sigmoid', 'hinge']")

    # -- rankpo_trainer.py:447-522 --------------------------------------------------------------------
    def _embed_pair(self, model, batch):
        """single_forward(query), single_forward(passage) -- through ONE packed encoder pass when the encoder offers it
        (sequences are independent: same rows, fewer and fuller launches)."""
        if hasattr(model, "pooled_last_token_multi"):
            q, p = batch["query"], batch["passage"]
            pooled = model.pooled_last_token_multi([(q["input_ids"], q["attention_mask"]),
                                                    (p["input_ids"], p["attention_mask"])])
            if pooled is not None:
                both = ops.pool_normalize(torch.cat(pooled, 0)[:, None, :], None, "cls", True)
                nq = pooled[0].shape[0]
                return both[:nq].contiguous(), both[nq:].contiguous()
        return self.single_forward(model, batch["query"]), self.single_forward(model, batch["passage"])

    def get_batch_loss_metrics(self, model, batch: Dict[str, Any], train_eval: Literal["train", "eval"] = "train",
                               sync_metrics: bool = True):
        """Returns (loss, metrics).  With sync_metrics=False `metrics` is a lazily-resolved handle (no host sync
        inside the step); the dict is materialised by `resolve_metrics` / `log`."""
        prefix = "eval_" if train_eval == "eval" else ""
        q, p = self._embed_pair(model, batch)
        ref_c = ref_r = None
        if self.ref_model is not None:                                   # :468-477
            with torch.inference_mode():
                rq, rp = self._embed_pair(self.ref_model, batch)
                cfg0 = self._cfg()
                cfg0.loss_type = "sigmoid" if cfg0.loss_type not in ("sigmoid", "hinge") else cfg0.loss_type
                _, rs, _, _ = ops.rankpo_loss_metrics(rq, rp, cfg0)
            ref_c, ref_r = rs[:, 0].clone(), rs[:, 1].clone()
        loss, scores, losses, mvec = ops.rankpo_loss_metrics(q, p, self._cfg(), ref_c, ref_r)
        if self.rankpo_weight <= 0.0 and self.sft_weight <= 0.0:
            loss = 0                                                      # the reference returns the int 0 (:482)
        handle = (prefix, mvec)
        if sync_metrics:
            return loss, self.resolve_metrics(handle)
        return loss, handle

    def resolve_metrics(self, handle) -> Dict[str, float]:
        prefix, mvec = handle
        m = mvec.detach()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # gather_for_metrics(x).mean() over equal per-rank batches == mean of the per-rank means: ONE all-reduce
            m = m.clone()
            dist.all_reduce(m, op=dist.ReduceOp.SUM)
            m = m / dist.get_world_size()
        vals = m.cpu().tolist()                                           # the single host copy
        out = {}
        for k, v in zip(METRIC_KEYS, vals):
            if k == "rankpo_loss" and not self.rankpo_weight > 0.0:
                continue
            if k == "sft_loss" and not self.sft_weight > 0.0:
                continue
            out[f"{prefix}{k}"] = v
        # same key order as the reference builds its dict (:496-520)
        return out

    # -- rankpo_trainer.py:570-587 --------------------------------------------------------------------
    def compute_loss(self, model, inputs: Dict[str, Any], return_outputs=False):
        """The reference resolves the 9 metrics to Python floats here, every micro-step (9 gathers + 9 `.item()` syncs,
        rankpo_trainer.py:496-520).  Nobody reads them before `log`, so unless the caller asks for them (`return_outputs`)
        the metric vector stays on the device and `log` resolves everything stored since the last log with ONE all-reduce and
        ONE host copy: the training step has no host sync on this path."""
        if return_outputs:
            loss, metrics = self.get_batch_loss_metrics(model, inputs, train_eval="train")
            self.store_metrics(metrics, train_eval="train")
            return (loss, metrics)
        loss, (prefix, mvec) = self.get_batch_loss_metrics(model, inputs, train_eval="train", sync_metrics=False)
        # a running device-side sum (one 9-float add per micro-step): bounded memory however rarely `log` is called
        if self._pending_metrics is None:
            self._pending_metrics = (prefix, mvec.detach().clone(), 1)
        else:
            _, acc, n = self._pending_metrics
            self._pending_metrics = (prefix, acc.add_(mvec.detach()), n + 1)
        return loss

    # -- rankpo_trainer.py:626-645 --------------------------------------------------------------------
    def store_metrics(self, metrics: Dict[str, float], train_eval: Literal["train", "eval"] = "train") -> None:
        for key, value in metrics.items():
            self._stored_metrics[train_eval][key].append(value)

    def _flush_pending(self):
        """Resolve the device-side metric sum of the micro-steps since the last log: the mean over the steps is taken on
        the device first (the mean over ranks and over steps commute), then ONE all-reduce + ONE host copy.
        COLLECTIVE when torch.distributed is up: every rank must get here, see `log`."""
        if self._pending_metrics is None:
            return
        prefix, acc, n = self._pending_metrics
        self._pending_metrics = None
        vals = self.resolve_metrics((prefix, acc / n))
        for key, value in vals.items():                       # `log` averages the stored values: n copies of the mean
            self._stored_metrics["train"][key].extend([value] * n)

    def log(self, logs: Dict[str, float]) -> Dict[str, float]:
        """rankpo_trainer.py:626-645.  Call it on EVERY rank, as the HF Trainer loop the reference copies does (its
        `_maybe_log_save_evaluate` calls `self.log` on all processes; only the printing is rank-0): with more than one rank a
        training log resolves the deferred metrics with one all-reduce (`_flush_pending`) -- the collective the reference issues
        inside every micro-step (9 `gather_for_metrics`) lives here instead -- so a loop that logs on rank 0 only would hang in it."""
        train_eval = "train" if "loss" in logs else "eval"
        if train_eval == "train":
            self._flush_pending()
        for key, metrics in self._stored_metrics[train_eval].items():
            logs[key] = torch.tensor(metrics).mean().item()
        del self._stored_metrics[train_eval]
        return logs
