"""Minimal training step around the hot path ("next" row f2 of SURVEY.md §8; reference: HF Trainer loop copied in
contrastive_trainer.py:487-612 + DeepSpeed ZeRO-1 bf16, scripts/train/run_contrastive.sh:33-44).

  micro-steps:  loss = model(**batch)["loss"]; loss.backward()      (GAS of them, negatives are per micro-batch)
  boundary:     gradient mean over ranks  (FlatGradAllReducer: async bucketed RCCL all-reduce during backward)
                global-norm clip (1.0)   -> one device scalar, no host sync
                AdamW (lr 1e-5, cosine, warmup 0.1) -> ONE rpo_adamw_step launch over the flat parameter space
                (bf16 parameters + f32 master / m / v), gradients zeroed by one memset.

Everything here is stream-ordered; `step()` returns the detached device loss so that logging decides when to sync.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Iterable, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import _lib
from ._lib import RPO_DT_BF16, RPO_DT_F32, check
from .distributed import FlatGradAllReducer


def cosine_with_warmup(step: int, total_steps: int, warmup_steps: int) -> float:
    """transformers' get_cosine_schedule_with_warmup multiplier (lr_scheduler_type cosine, warmup_ratio 0.1)."""
    if step < warmup_steps:
        return step / max(1, warmup_steps)
    prog = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


class FlatAdamW:
    """AdamW over one flat parameter buffer; parameters and gradients are views into flat storage."""

    def __init__(self, params: Iterable[nn.Parameter], lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 max_grad_norm: Optional[float] = 1.0, bucket_mb: float = 512.0, force_collectives: bool = False):
        self.reducer = FlatGradAllReducer(list(params), bucket_mb=bucket_mb, force_collectives=force_collectives)
        r = self.reducer
        dev, dtype = r.flat.device, r.flat.dtype
        if not r.flat.is_cuda:
            raise RuntimeError("FlatAdamW runs on a HIP device only (no CPU fallback)")
        self.dt = RPO_DT_BF16 if dtype == torch.bfloat16 else RPO_DT_F32
        if dtype not in (torch.bfloat16, torch.float32):
            raise TypeError(f"unsupported parameter dtype {dtype}")
        # flat parameter storage in the same layout as the gradients; parameters become views of it
        self.flat_param = torch.zeros(r.numel, dtype=dtype, device=dev)
        for p, o in zip(r.order, r.offsets):
            v = self.flat_param[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
        self.master = self.flat_param.float() if dtype == torch.bfloat16 else None
        self.exp_avg = torch.zeros(r.numel, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(r.numel, dtype=torch.float32, device=dev)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.t = 0
        self._nblk = 1024
        self._partial = torch.empty(self._nblk, dtype=torch.float32, device=dev)
        self.last_grad_norm = None

    def grad_norm(self, pre_scale: float) -> torch.Tensor:
        """||pre_scale * grad||_2 as a device scalar."""
        lib = _lib.load()
        g = self.reducer.flat
        with torch.cuda.device(g.device):
            check(lib.rpo_sumsq_partial(g.data_ptr(), g.numel(), self.dt, self._partial.data_ptr(), self._nblk,
                                        torch.cuda.current_stream(g.device).cuda_stream), "rpo_sumsq_partial")
        return self._partial.sum().sqrt() * pre_scale

    def step(self, grad_scale: float = 1.0, lr_mult: float = 1.0):
        """grad_scale: constant factor on the accumulated gradients (1/GAS/world)."""
        lib = _lib.load()
        r = self.reducer
        self.t += 1
        scale = torch.full((1,), grad_scale, dtype=torch.float32, device=r.flat.device)
        if self.max_grad_norm is not None:
            norm = self.grad_norm(grad_scale)
            self.last_grad_norm = norm
            scale = scale * torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0)   # clip_grad_norm_ semantics
        b1, b2 = self.betas
        with torch.cuda.device(r.flat.device):
            check(lib.rpo_adamw_step(self.flat_param.data_ptr(), None if self.master is None else self.master.data_ptr(),
                                     r.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), r.numel,
                                     self.dt, self.lr * lr_mult, b1, b2, self.eps, self.weight_decay,
                                     1.0 - b1 ** self.t, 1.0 - b2 ** self.t, scale.data_ptr(),
                                     torch.cuda.current_stream(r.flat.device).cuda_stream), "rpo_adamw_step")
        r.zero_()


class TrainStep:
    """loss_fn(batch) -> scalar loss tensor (e.g. `lambda b: model(**b)["loss"]` or `trainer.compute_loss`)."""

    def __init__(self, params, loss_fn: Callable[[Dict], torch.Tensor], *, lr=1e-5, weight_decay=0.0,
                 max_grad_norm=1.0, gradient_accumulation_steps=1, total_steps=1000, warmup_ratio=0.1,
                 bucket_mb=512.0, force_collectives=False):
        self.opt = FlatAdamW(params, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                             bucket_mb=bucket_mb, force_collectives=force_collectives)
        self.loss_fn = loss_fn
        self.gas = gradient_accumulation_steps
        self.total_steps = total_steps
        self.warmup_steps = int(math.ceil(total_steps * warmup_ratio))
        self.global_step = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def micro_step(self, batch, last: bool) -> torch.Tensor:
        if last:
            self.opt.reducer.arm()
        loss = self.loss_fn(batch)
        loss.backward()
        return loss.detach()

    def step(self, batches) -> torch.Tensor:
        """`batches`: a list of GAS micro-batches (or a single batch when GAS == 1)."""
        if isinstance(batches, dict):
            batches = [batches]
        assert len(batches) == self.gas
        tot = None
        for i, b in enumerate(batches):
            l = self.micro_step(b, last=(i == self.gas - 1))
            tot = l if tot is None else tot + l
        inv_world = self.opt.reducer.finish()
        mult = cosine_with_warmup(self.global_step, self.total_steps, self.warmup_steps)   # HF: scheduler steps after the optimizer
        self.opt.step(grad_scale=inv_world / self.gas, lr_mult=mult)
        self.global_step += 1
        return tot / self.gas
