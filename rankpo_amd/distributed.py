"""Cross-device pieces of the hot path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI
on ROCm; "gloo" in the CPU tests).

* `all_gather_with_local_grad`  -- reference `distributed_gather` semantics (modeling.py:331-404): rank-major
  concatenation; backward hands each rank the gradient of ITS OWN slice, no backward collective.
* `FusedQPGather` / `EmbeddingGather` -- what `ModelForTraining.forward` uses: the gathered matrices are constants for
  autograd (the InfoNCE kernel produces gradients for the local rows directly), so the gather is a plain asynchronous
  collective: ONE all-gather of the `q ‖ p` block when both towers ran as one packed pass (the default), or the passage
  gather in flight during the query tower when the towers run one after the other.
* `FlatGradAllReducer` -- data-parallel gradient mean over flat buckets (replaces DeepSpeed ZeRO-1's reduction;
  SURVEY.md §8e): buckets are all-reduced asynchronously as soon as backward has produced them.
* `rebalance_groups` -- with cross-device negatives every step ends in a gather all ranks wait at, so a step lasts as long as the
  rank with the most tokens; the (query + its passages) groups of the GLOBAL batch are re-dealt to the ranks by packed-token
  cost before the encoder runs.  The global batch, hence the loss and its gradient, is unchanged.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def _all_gather_into(out: torch.Tensor, x: torch.Tensor, async_op=False):
    """out: [W, *x.shape] contiguous.  nccl (= RCCL): one all_gather_into_tensor.  gloo (the CPU tests; and device tensors in
    `bench.py --share-gpu`, a rehearsal of the N > 1 path on one GPU): gloo gathers host tensors only, device tensors are staged
    through the host (synchronous; returns None instead of a work handle)."""
    if dist.get_backend() == "nccl":
        return dist.all_gather_into_tensor(out.view(-1), x.reshape(-1), async_op=async_op)
    if x.is_cuda:
        host = [torch.empty(x.shape, dtype=x.dtype) for _ in range(out.shape[0])]
        dist.all_gather(host, x.cpu())
        out.copy_(torch.stack(host).to(out.device))
        return None
    return dist.all_gather(list(out.unbind(0)), x, async_op=async_op)


def deal_balanced(costs, hands: int) -> List[List[int]]:
    """Deal len(costs) items (a multiple of `hands`) into `hands` hands of EQUAL size so that the most expensive hand is as cheap
    as a greedy pass gets it: items by decreasing cost, each to the cheapest hand that is not full yet (longest-processing-time
    first under a cardinality constraint, then pairwise swaps out of the most expensive hand while they help).  Deterministic (ties by index), so every rank computes the same deal from the same
    costs.  Returns the item indices of each hand in increasing order."""
    n = len(costs)
    if hands <= 0 or n % hands:
        raise ValueError(f"cannot deal {n} items into {hands} equal hands")
    per = n // hands
    total = [0.0] * hands
    out: List[List[int]] = [[] for _ in range(hands)]
    for i in sorted(range(n), key=lambda j: (-float(costs[j]), j)):
        h = min((k for k in range(hands) if len(out[k]) < per), key=lambda k: (total[k], k))
        out[h].append(i)
        total[h] += float(costs[i])
    # refinement: swap one item of the most expensive hand with one of another hand while that lowers the larger of the two totals
    for _ in range(4 * n):
        hi = max(range(hands), key=lambda k: (total[k], -k))
        best = None
        for lo in range(hands):
            if lo == hi:
                continue
            for a in out[hi]:
                for b in out[lo]:
                    d = float(costs[a]) - float(costs[b])
                    if d <= 0.0:
                        continue
                    new_max = max(total[hi] - d, total[lo] + d)
                    if new_max < total[hi] - 1e-12 and (best is None or new_max < best[0]):
                        best = (new_max, lo, a, b)
        if best is None:
            break
        _, lo, a, b = best
        d = float(costs[a]) - float(costs[b])
        out[hi][out[hi].index(a)] = b
        out[lo][out[lo].index(b)] = a
        total[hi] -= d
        total[lo] += d
    return [sorted(h) for h in out]


# cost of one packed sequence of L tokens in units of "token through the GEMMs": the projections and the MLP are linear in L, causal
# attention is L^2 / 2; on the Llama-3.2-1B shape attention is ~19 % of the encoder's time at 4096-token rows
_ATTN_COST_AT = (4096, 0.25)


def sequence_cost(lengths: torch.Tensor) -> torch.Tensor:
    L = lengths.to(torch.float64)
    return L * (1.0 + _ATTN_COST_AT[1] * L / _ATTN_COST_AT[0])


def rebalance_groups(query: dict, passage: dict, pad_token_id: int = 0):
    """query: {'input_ids', 'attention_mask'} [B, Lq]; passage: the same [B G, Lp] with the G passages of query b in rows
    b G .. b G + G - 1 (the collators' layout, data_utils.py).  Every rank holds B such groups; returns this rank's B groups of a
    re-deal of all W B groups that evens out the ranks' packed-token cost, plus the per-rank costs before and after (host
    floats).  Mechanism: the padded widths are made equal over the ranks (all-reduce MAX), ids and masks are all-gathered (a few
    MB of integers per step at the reference's shapes), every rank computes the same deal and keeps its hand -- no second
    exchange.  One host sync (the lengths).  The union of the ranks' batches is the same set of groups as before, so with
    cross-device negatives the loss (a mean over the global batch, modeling.py:287-314) and its gradient do not change; a rank's
    OWN rows do (reference: the sampler's assignment of examples to ranks is random to begin with)."""
    W, r = dist.get_world_size(), dist.get_rank()
    B = query["input_ids"].shape[0]
    n_pass = passage["input_ids"].shape[0]
    G = n_pass // B if B else 0
    # A rank-local problem (an empty batch, passages that are not a whole number of groups) must not raise on THIS rank alone: its
    # peers would sit in the all-reduce below until the process-group timeout.  The verdict travels in the same all-reduce as the
    # widths, so every rank raises together.
    local_ok = B > 0 and G > 0 and n_pass == B * G
    dev = query["input_ids"].device
    # widths to agree on, and the batch shape every rank must share (max of x and of -x = min: every rank sees a mismatch and raises)
    widths = torch.tensor([query["input_ids"].shape[1], passage["input_ids"].shape[1], B, -B, G, -G, 0 if local_ok else 1],
                          dtype=torch.int64, device=dev)
    if dist.get_backend() != "nccl" and widths.is_cuda:
        wh = widths.cpu()
        dist.all_reduce(wh, op=dist.ReduceOp.MAX)
        widths = wh
    else:
        dist.all_reduce(widths, op=dist.ReduceOp.MAX)
    Lq, Lp, Bmax, nBmin, Gmax, nGmin, any_bad = (int(x) for x in widths.tolist())
    if any_bad:
        mine = "" if local_ok else f" (this rank: {n_pass} passages for {B} queries)"
        raise ValueError(f"rebalance_groups: on at least one rank the passages are not a whole number of groups for its queries, or "
                         f"the batch is empty{mine}")
    if Bmax != -nBmin or Gmax != -nGmin:
        raise ValueError(f"rebalance_groups: the ranks hold different batch shapes (queries {-nBmin}..{Bmax}, passages per query "
                         f"{-nGmin}..{Gmax}); the re-deal needs equal per-rank batches (drop_last)")

    def widen(t, width, value):
        if t.shape[1] == width:
            return t.contiguous()
        out = torch.full((t.shape[0], width), value, dtype=t.dtype, device=t.device)
        out[:, :t.shape[1]] = t
        return out
    # one block per rank: [B, Lq + G Lp] ids | the same for the masks
    ids = torch.cat([widen(query["input_ids"], Lq, pad_token_id), widen(passage["input_ids"], Lp, pad_token_id).view(B, G * Lp)], 1)
    msk = torch.cat([widen(query["attention_mask"], Lq, 0), widen(passage["attention_mask"], Lp, 0).view(B, G * Lp)], 1)
    block = torch.stack([ids, msk.to(ids.dtype)]).contiguous()                       # [2, B, Lq + G Lp]
    allb = torch.empty((W,) + tuple(block.shape), dtype=block.dtype, device=dev)
    _all_gather_into(allb, block)
    all_ids = allb[:, 0].reshape(W * B, Lq + G * Lp)
    all_msk = allb[:, 1].reshape(W * B, Lq + G * Lp)
    lens = torch.cat([all_msk[:, :Lq].sum(-1, keepdim=True), all_msk[:, Lq:].view(W * B, G, Lp).sum(-1)], 1)    # [W B, 1 + G]
    cost = sequence_cost(lens).sum(-1).tolist()                                        # the one host sync
    hands = deal_balanced(cost, W)
    before = [sum(cost[k * B:(k + 1) * B]) for k in range(W)]
    after = [sum(cost[i] for i in h) for h in hands]
    mine = torch.tensor(hands[r], dtype=torch.int64, device=dev)
    ids_r, msk_r = all_ids.index_select(0, mine), all_msk.index_select(0, mine)
    q = {"input_ids": ids_r[:, :Lq].contiguous(), "attention_mask": msk_r[:, :Lq].contiguous().to(query["attention_mask"].dtype)}
    p = {"input_ids": ids_r[:, Lq:].reshape(B * G, Lp), "attention_mask": msk_r[:, Lq:].reshape(B * G, Lp).to(passage["attention_mask"].dtype)}
    return q, p, {"cost_before": before, "cost_after": after, "groups": hands[r]}


class _AllGatherLocalGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        W = dist.get_world_size()
        out = torch.empty((W,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        _all_gather_into(out, x.contiguous())
        ctx.n0 = x.shape[0]
        return out.view((W * x.shape[0],) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, g):
        r = dist.get_rank()
        return g[r * ctx.n0:(r + 1) * ctx.n0].contiguous()


def all_gather_with_local_grad(x: torch.Tensor) -> torch.Tensor:
    return _AllGatherLocalGrad.apply(x)


class EmbeddingGather:
    """Asynchronous rank-major all-gather of a [n, d] embedding block (no autograd)."""

    def __init__(self, x: torch.Tensor):
        W = dist.get_world_size()
        self.x = x.detach().contiguous()
        self.out = torch.empty((W,) + tuple(self.x.shape), dtype=x.dtype, device=x.device)
        self.work = _all_gather_into(self.out, self.x, async_op=True)

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()          # NCCL: makes the current stream wait for the collective; no host sync
            self.work = None
        return self.out.view((-1,) + tuple(self.x.shape[1:]))


class FusedQPGather:
    """The reference's two `distributed_gather` calls (modeling.py:289-290) as ONE collective: every rank contributes its
    [B + B G, d] block `q ‖ p` (the packed encoder pass already emits the pooled rows in that order), the all-gather lands in
    one [W, B + B G, d] buffer, and `wait()` returns the rank-major matrices the loss is computed on,
    q_all [W B, d] and p_all [W B G, d] (two small copies out of the buffer: 32 KiB and 192 KiB per rank at cfg 3).
    Asynchronous: issued on RCCL's stream, the launch stream only waits in `wait()`.  No autograd: the gathered matrices
    are constants, the InfoNCE backward produces this rank's rows directly (modeling.py:374-377 semantics)."""

    def __init__(self, qp: torch.Tensor, num_queries: int):
        self.nq = int(num_queries)
        self._g = EmbeddingGather(qp)

    def wait(self):
        out = self._g.out                                    # [W, B + B G, d]
        self._g.wait()
        d = out.shape[-1]
        return out[:, :self.nq].reshape(-1, d), out[:, self.nq:].reshape(-1, d)


def gather_embeddings(q: torch.Tensor, p: torch.Tensor):
    """Gather q and p across ranks with one collective; returns (q_all [W*B, d], p_all [W*BG, d]) as autograd constants."""
    return FusedQPGather(torch.cat([q.detach(), p.detach()], 0), q.shape[0]).wait()


class FlatGradAllReducer:
    """Parameters' .grad tensors are views into ONE flat buffer per dtype; the buffer is cut into buckets of
    ~`bucket_mb` MiB (large: xGMI is point-to-point, few big collectives beat many small ones) that are
    all-reduced (mean) on RCCL's stream as soon as every gradient inside has been accumulated."""

    def __init__(self, params: List[torch.nn.Parameter], bucket_mb: float = 512.0, world_size: Optional[int] = None,
                 force_collectives: bool = False, shard: bool = False, rank: Optional[int] = None):
        """shard=True (optimizer-state partition, `train_step.FlatAdamW(partition=True)`): every bucket is cut into `world` equal
        shards and REDUCE-SCATTERED instead of all-reduced -- rank r receives the summed gradient of shard r of every bucket
        (`grad_shards[b]`), the only part its optimizer slice needs; bucket sizes are padded to a multiple of 8 x world elements."""
        self.params = [p for p in params if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.shard = bool(shard)
        assert self.params, "no trainable parameters"
        dtype, device = self.params[0].dtype, self.params[0].device
        assert all(p.dtype == dtype and p.device == device for p in self.params), "one dtype/device per reducer"
        # backward produces gradients roughly in reverse registration order: lay the flat buffer out that way
        order = list(reversed(self.params))
        # weights tagged as a fuse group (encoder.tag_fuse_group: q|k|v, gate|up) stay adjacent and in group order, so
        # that their row-concatenation is a plain view of the flat buffer
        i = 0
        while i < len(order):
            tag = getattr(order[i], "_rpo_fuse_group", None)
            j = i + 1
            if tag is not None:
                while j < len(order) and getattr(order[j], "_rpo_fuse_group", (None,))[0] is tag[0]:
                    j += 1
                order[i:j] = sorted(order[i:j], key=lambda p: p._rpo_fuse_group[1])
            i = j
        # layout: parameters back to back (every view 16-byte aligned), cut into buckets of >= `per` elements at parameter
        # boundaries; with shard=True a bucket's length is padded up to a multiple of 8 x world so that its `world` shards are
        # equal and 16-byte aligned (the padding elements belong to no parameter and stay zero)
        es = torch.empty((), dtype=dtype).element_size()
        per = max(1, int(bucket_mb * 2 ** 20 / es))
        quantum = 8 * self.world if self.shard else 8
        offs, n = [], 0
        self.buckets = []                              # (start, end, [param indices])
        start, idxs = 0, []
        for i, p in enumerate(order):
            offs.append(n)
            n += (p.numel() + 7) // 8 * 8
            idxs.append(i)
            if n - start >= per or i == len(order) - 1:
                n = start + (n - start + quantum - 1) // quantum * quantum
                self.buckets.append((start, n, idxs))
                start, idxs = n, []
        self.numel = n
        self.flat = torch.zeros(n, dtype=dtype, device=device)
        self.order, self.offsets = order, offs
        for p, o in zip(order, offs):
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        # shard=True: this rank's reduced gradient shards, one contiguous buffer, bucket b at shard_offsets[b]
        self.shard_offsets, self.shard_numel, self.grad_shards = [], 0, None
        if self.shard:
            for s_, e_, _ in self.buckets:
                self.shard_offsets.append(self.shard_numel)
                self.shard_numel += (e_ - s_) // self.world
            self.grad_shards = torch.zeros(self.shard_numel, dtype=dtype, device=device)
        self._bucket_of = {}
        for b, (_, _, ids) in enumerate(self.buckets):
            for i in ids:
                self._bucket_of[i] = b
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._armed = False
        self.late_buckets = 0                      # buckets that finish() had to reduce (diagnostic)
        self._reduce = self.world > 1 or (force_collectives and dist.is_initialized())   # force: 1-rank rehearsal
        if self._reduce:
            for i, p in enumerate(order):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(param):
            # autograd may have replaced .grad; keep the flat view authoritative
            o = self.offsets[i]
            view = self.flat[o:o + param.numel()].view_as(param)
            if param.grad is not None and param.grad.data_ptr() != view.data_ptr():
                view.copy_(param.grad)
                param.grad = view
            if not self._armed:
                return
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._works.append(self._reduce_bucket(b))
        return hook

    def shard_view(self, b: int) -> torch.Tensor:
        """This rank's reduced-gradient shard of bucket b (shard=True)."""
        s, e, _ = self.buckets[b]
        o = self.shard_offsets[b]
        return self.grad_shards[o:o + (e - s) // self.world]

    def _reduce_bucket(self, b: int):
        """Asynchronous sum over ranks of bucket b: all-reduce in place, or (shard=True) reduce-scatter into this rank's shard."""
        s, e, _ = self.buckets[b]
        if not self.shard:
            return dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, async_op=True)
        out = self.shard_view(b)
        if dist.get_backend() == "nccl":
            return dist.reduce_scatter_tensor(out, self.flat[s:e], op=dist.ReduceOp.SUM, async_op=True)
        # gloo (the CPU tests) has no reduce-scatter: all-reduce, then keep the own shard -- the same values
        n = (e - s) // self.world
        dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM)
        out.copy_(self.flat[s + self.rank * n:s + (self.rank + 1) * n])
        return None

    def arm(self):
        """Call before the LAST backward of an accumulation window: buckets then reduce as they complete."""
        self._armed = self._reduce
        self._pending = [len(ids) for _, _, ids in self.buckets]
        self._works = []

    def finish(self):
        """Wait for the bucket all-reduces (stream wait, no host sync).  The 1/world mean factor is folded into
        the optimizer's grad scale (returned).
        A bucket fires from the hooks only when EVERY parameter in it received a gradient in the armed backward; a
        parameter without one (unused branch, frozen sub-path on this step) would otherwise leave its whole bucket
        un-reduced and the ranks would diverge silently.  Such buckets are reduced here, in bucket order (their missing
        gradients are the zeros the flat buffer holds).  The sequence of collectives stays identical on all ranks as long as
        the ranks run the same graph -- the usual data-parallel contract."""
        if self._armed:
            for b, left in enumerate(self._pending):
                if left > 0:
                    self._works.append(self._reduce_bucket(b))
                    self._pending[b] = 0
                    self.late_buckets += 1
        for w in self._works:
            if w is not None:
                w.wait()
        self._works = []
        self._armed = False
        return 1.0 / self.world

    def reset(self):
        """Forget a step that did not complete (an exception between arm() and finish()): disarmed, nothing pending, gradients
        zeroed.  Single-rank use only -- with peers, a rank that leaves a step mid-way has already broken the collective sequence."""
        self._armed = False
        self._pending = [0] * len(self.buckets)
        # bucket reductions the hooks fired before the failure may still be running on the communicator's stream (world == 1 over
        # RCCL, the --force-dist rehearsal, is allowed to retry): they write the flat buffer, so they finish before it is zeroed
        for w in self._works:
            try:
                w.wait()
            except Exception:                   # a reduction that failed with the step: nothing of it may be trusted, nor awaited
                pass
        self._works = []
        if self.flat.is_cuda:
            torch.cuda.synchronize(self.flat.device)
        self.zero_()

    def zero_(self):
        self.flat.zero_()
