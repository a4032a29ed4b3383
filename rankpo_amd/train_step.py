"""Minimal training step around the hot path ("next" row f2 of SURVEY.md §8; reference: HF Trainer loop copied in
contrastive_trainer.py:487-612 + DeepSpeed ZeRO-1 bf16, scripts/train/run_contrastive.sh:33-44).

  micro-steps:  loss = model(**batch)["loss"]; loss.backward()      (GAS of them, negatives are per micro-batch)
  boundary:     gradient mean over ranks  (FlatGradAllReducer: async bucketed RCCL all-reduce during backward; or, with
                `partition_optimizer`, reduce-scatter + AdamW on this rank's 1 / W of the state + all-gather of the parameters)
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
    """AdamW over one flat parameter buffer; parameters and gradients are views into flat storage.

    partition=True (with more than one rank): the optimizer state is PARTITIONED over the ranks, the xGMI-native form of what the
    reference configures as DeepSpeed ZeRO stage 1 (configs/ds_zero1_config_llama.json:10-12).  Every gradient bucket is
    reduce-scattered instead of all-reduced (rank r gets the summed shard r of every bucket), rank r keeps float32 master / m / v
    for ITS shards only (12 B/parameter / W instead of 12 B/parameter: 90 GB -> 11 GB per GPU for Llama-3-8B at W = 8) and runs
    AdamW on them, then the bf16 parameters are all-gathered bucket by bucket.  Bytes on the wire = the all-reduce's (a ring
    all-reduce IS a reduce-scatter + an all-gather).  Same sums, same element-wise update as the replicated path; the global
    gradient norm is summed shard-wise first, so the two paths agree to float32 summation order (the 2- / 4-rank tests assert
    equal parameters after 3 steps to that tolerance, and identical replicas)."""

    def __init__(self, params: Iterable[nn.Parameter], lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 max_grad_norm: Optional[float] = 1.0, bucket_mb: float = 512.0, force_collectives: bool = False,
                 partition: bool = False):
        params = list(params)
        dist_on = dist.is_available() and dist.is_initialized()
        self.partition = bool(partition) and dist_on and (dist.get_world_size() > 1 or force_collectives)
        self.reducer = FlatGradAllReducer(params, bucket_mb=bucket_mb, force_collectives=force_collectives, shard=self.partition)
        r = self.reducer
        dev, dtype = r.flat.device, r.flat.dtype
        self.dt = RPO_DT_BF16 if dtype == torch.bfloat16 else RPO_DT_F32
        if dtype not in (torch.bfloat16, torch.float32):
            raise TypeError(f"unsupported parameter dtype {dtype}")
        # flat parameter storage in the same layout as the gradients; parameters become views of it
        self.flat_param = torch.zeros(r.numel, dtype=dtype, device=dev)
        for p, o in zip(r.order, r.offsets):
            v = self.flat_param[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
        # optimizer state: the whole flat space, or (partition) this rank's shard of every bucket, back to back
        n_state = r.shard_numel if self.partition else r.numel
        self.state_numel = n_state
        if self.partition:
            own = torch.cat([self._param_shard(b) for b in range(len(r.buckets))])
            self.master = own.float() if dtype == torch.bfloat16 else None
        else:
            self.master = self.flat_param.float() if dtype == torch.bfloat16 else None
        self.exp_avg = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.t = 0
        self._nblk = 1024
        self._partial = torch.empty(self._nblk, dtype=torch.float32, device=dev)
        self.last_grad_norm = None

    def _param_shard(self, b: int) -> torch.Tensor:
        """This rank's slice of bucket b of the flat parameter buffer."""
        r = self.reducer
        s, e, _ = r.buckets[b]
        n = (e - s) // r.world
        return self.flat_param[s + r.rank * n:s + (r.rank + 1) * n]

    # -- the two kernel calls (tensors in, launch on the current stream); the CPU tests of the partition logic replace them ------
    def _sumsq(self, g: torch.Tensor) -> torch.Tensor:
        """sum of squares of g as a device scalar (f32)."""
        if not g.is_cuda:
            raise RuntimeError("FlatAdamW runs on a HIP device only (no CPU fallback)")
        lib = _lib.load()
        with torch.cuda.device(g.device):
            check(lib.rpo_sumsq_partial(g.data_ptr(), g.numel(), self.dt, self._partial.data_ptr(), self._nblk,
                                        torch.cuda.current_stream(g.device).cuda_stream), "rpo_sumsq_partial")
        return self._partial.sum()

    def _adamw(self, param, master, grad, m, v, lr, bc1, bc2, scale):
        if not param.is_cuda:
            raise RuntimeError("FlatAdamW runs on a HIP device only (no CPU fallback)")
        lib = _lib.load()
        b1, b2 = self.betas
        with torch.cuda.device(param.device):
            check(lib.rpo_adamw_step(param.data_ptr(), None if master is None else master.data_ptr(), grad.data_ptr(),
                                     m.data_ptr(), v.data_ptr(), param.numel(), self.dt, lr, b1, b2, self.eps, self.weight_decay,
                                     bc1, bc2, scale.data_ptr(), torch.cuda.current_stream(param.device).cuda_stream),
                  "rpo_adamw_step")

    def grad_norm(self, pre_scale: float) -> torch.Tensor:
        """||pre_scale * grad||_2 of the (rank-summed) gradient as a device scalar."""
        if self.partition:
            # every rank holds 1 / W of the summed gradient: local sum of squares, then ONE scalar all-reduce
            ss = self._sumsq(self.reducer.grad_shards).reshape(1).clone()
            dist.all_reduce(ss, op=dist.ReduceOp.SUM)
            return ss[0].sqrt() * pre_scale
        return self._sumsq(self.reducer.flat).sqrt() * pre_scale

    def step(self, grad_scale: float = 1.0, lr_mult: float = 1.0):
        """grad_scale: constant factor on the accumulated gradients (1/GAS/world)."""
        r = self.reducer
        self.t += 1
        scale = torch.full((1,), grad_scale, dtype=torch.float32, device=r.flat.device)
        if self.max_grad_norm is not None:
            norm = self.grad_norm(grad_scale)
            self.last_grad_norm = norm
            scale = scale * torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0)   # clip_grad_norm_ semantics
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        lr = self.lr * lr_mult
        if not self.partition:
            self._adamw(self.flat_param, self.master, r.flat, self.exp_avg, self.exp_avg_sq, lr, bc1, bc2, scale)
            r.zero_()
            return
        works = []
        for b, (s, e, _) in enumerate(r.buckets):
            o, n = r.shard_offsets[b], (e - s) // r.world
            own = self._param_shard(b)
            self._adamw(own, None if self.master is None else self.master[o:o + n], r.grad_shards[o:o + n],
                        self.exp_avg[o:o + n], self.exp_avg_sq[o:o + n], lr, bc1, bc2, scale)
            works.append(self._all_gather_bucket(b, own))          # in flight while the next bucket's AdamW runs
        self._wait_gathers(works)
        r.zero_()

    def _wait_gathers(self, works):
        """Launch stream waits for the parameter all-gathers (no host sync); a method of its own so that bench.py can bracket it."""
        for w in works:
            if w is not None:
                w.wait()

    def _all_gather_bucket(self, b: int, own: torch.Tensor):
        """Every rank's updated shard of bucket b -> the whole bucket of the flat parameter buffer, on every rank."""
        r = self.reducer
        s, e, _ = r.buckets[b]
        if dist.get_backend() == "nccl":
            # in place: `own` IS slice `rank` of the output (NCCL's in-place all-gather form)
            return dist.all_gather_into_tensor(self.flat_param[s:e], own, async_op=True)
        from .distributed import _all_gather_into
        n = (e - s) // r.world
        _all_gather_into(self.flat_param[s:e].view(r.world, n), own.clone())     # gloo: host tensors, or staged through the host
        return None


class TrainStep:
    """loss_fn(batch) -> scalar loss tensor (e.g. `lambda b: model(**b)["loss"]` or `trainer.compute_loss`)."""

    def __init__(self, params, loss_fn: Callable[[Dict], torch.Tensor], *, lr=1e-5, weight_decay=0.0,
                 max_grad_norm=1.0, gradient_accumulation_steps=1, total_steps=1000, warmup_ratio=0.1,
                 bucket_mb=512.0, force_collectives=False, partition_optimizer=False):
        self.opt = FlatAdamW(params, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                             bucket_mb=bucket_mb, force_collectives=force_collectives, partition=partition_optimizer)
        self.loss_fn = loss_fn
        self.gas = gradient_accumulation_steps
        self.total_steps = total_steps
        self.warmup_steps = int(math.ceil(total_steps * warmup_ratio))
        self.global_step = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._in_optimizer = False       # True while opt.step runs: an exception there leaves half-updated moments (no retry)

    def micro_step(self, batch, last: bool) -> torch.Tensor:
        if last:
            self.opt.reducer.arm()
        loss = self.loss_fn(batch)
        loss.backward()
        return loss.detach()

    def abort_step(self):
        """After an exception inside `step` (e.g. an out-of-memory error the caller answers by checkpointing more blocks): drop
        the half-accumulated gradients and the reducer's armed state, so that the next `step` starts clean.  The optimizer's
        moments and step count are untouched (the failed step never reached `opt.step`).  An exception that came out of `opt.step`
        itself (the clip-norm or all-gather temporaries running out of memory) is NOT retryable -- master weights and moments may
        be half-updated -- and is refused here: the caller's `except` re-raises."""
        if self._in_optimizer:
            raise RuntimeError("TrainStep.abort_step: the step failed inside the optimizer update; its state may be half-updated and "
                               "the step cannot be retried")
        r = self.opt.reducer
        r.reset()
        for p, o in zip(r.order, r.offsets):                   # every .grad is the parameter's view of the flat buffer again
            p.grad = r.flat[o:o + p.numel()].view_as(p)

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
        self._in_optimizer = True
        self.opt.step(grad_scale=inv_world / self.gas, lr_mult=mult)
        self._in_optimizer = False
        self.global_step += 1
        return tot / self.gas
