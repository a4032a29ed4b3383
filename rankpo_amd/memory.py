"""HBM plan of a training step: which blocks keep their activations and which are checkpointed (the reference's
`--gradient_checkpointing` flag, scripts/train/run_contrastive.sh:39 / run_rankpo.sh, is all-or-nothing: HF checkpoints every block).

288 GB of HBM3E per MI355X changes the question from "checkpoint or not" to "how many blocks": Llama-3.2-1B at the BASELINE batch
(8 x 1280 + 48 x 4096 tokens) keeps every block's activations (198 GiB measured), Llama-3-8B on one GPU keeps 3 of 32, and on 8 GPUs
with the optimizer state partitioned most of them.  `LlamaEncoder.gradient_checkpointing_enable()` (no argument) asks this module
with the measured free HBM, the world size and the padded token count of the batches it actually sees (`plan_for_encoder`);
bench.py drives the same functions explicitly and then corrects the plan once from a MEASURED worst-case step
(`checkpoint_fewer`, `admit_transposed_dgu`).

Everything here is host arithmetic on byte counts: no device work, importable without a GPU (tests/test_host_logic.py).
"""
from __future__ import annotations

from dataclasses import dataclass

PLAN_HBM_FRACTION = 0.85          # of the usable HBM: what the plan lets the MODELLED worst-case peak reach (the allocator's own
#                                   slack on top of it was 13 % on cfg 5: 286 GiB reserved for 253 GiB allocated)
PRESIZE_TIGHT_FRACTION = 0.94     # a MEASURED worst-case peak above this share of the usable HBM counts as "does not fit"
DGU_T_ROOM_FRACTION = 0.90        # the transposed d(gate|up) buffer is admitted while peak + 2 x buffer stays below this share
ACT_KEEP_FRACTION = 0.8           # kept activations of an un-checkpointed block / the upper bound es (6.5 d + 2 ff) per token
WORKING_SET_BLOCKS = 2.4          # recomputed activations + backward temporaries of the block in flight, in kept-block units


def usable_hbm(free_now, reserved_by_this_process, total, ranks_sharing_the_card=1):
    """HBM this rank can really use: what the device reports free NOW (after RCCL created its communicator, with whatever else
    lives on the card) plus what this process already holds; `--share-gpu` rehearsals split one card between the ranks."""
    return min((free_now + reserved_by_this_process) // max(1, ranks_sharing_the_card), total)


def optimizer_state_bytes(nparam, es, world, partitioned):
    """parameters + gradients (2 es B/param) + f32 master / m / v (12 B/param; or 12 / W with the state partitioned over W
    ranks, + the es / W reduced-gradient shard)."""
    return nparam * (es * 2 + ((12 + es) / max(1, world) if partitioned else 12))


def modelled_peak_bytes(free_blocks, nparam, es, hidden, inter, nl, tok_pad, world=1, partitioned=False, block_inputs=2):
    """Worst-case (every row at full length) peak of allocated HBM with `free_blocks` of the nl blocks un-checkpointed:
      states  +  checkpointed blocks x their kept INPUTS  +  free blocks x their kept activations  +  the block in flight.
    A block's input is TWO [tokens, d] tensors -- the residual stream travels as (x, delta) because the add is fused into the
    RMSNorm that follows it (encoder.LlamaLayer.forward(x, delta, ...)) -- which round 4's plan counted once: the 45 GiB nobody
    had accounted for at cfg 5 (profiles/r05_cfg5_memory_summary.txt: 253.4 GiB allocated = 111.8 states + 2 x 50.5 inputs +
    40.6 in flight; allocator slack is on top of that, not inside it).  Kept activations per token of a free block: x, norm(x),
    q|k|v, attention out, x', norm(x'), gate|up (the SwiGLU product is recomputed) <= es (6.5 d + 2 ff), measured 0.75 of that
    on Llama-3.2-1B.  The block in flight (recomputed activations + d(gate|up), d(product), the transposed product ...) was
    calibrated on the cfg-5 peak: 2.4 kept-block units.  cfg 2 (all 16 blocks free): modelled 187 GiB, measured 198.
    block_inputs = 1 when the encoder forms x + delta in front of a checkpointed block (encoder.CKPT_SINGLE_INPUT, round 5)."""
    per_layer = tok_pad * int(ACT_KEEP_FRACTION * es * (6.5 * hidden + 2 * inter))
    return (optimizer_state_bytes(nparam, es, world, partitioned) + (nl - free_blocks) * block_inputs * tok_pad * hidden * es
            + free_blocks * per_layer + WORKING_SET_BLOCKS * per_layer)


def plan_free_blocks(hbm_usable, nparam, es, hidden, inter, nl, tok_pad, world=1, partitioned=False, block_inputs=2):
    """How many of the nl blocks may run WITHOUT activation checkpointing: the largest count whose modelled worst-case peak
    stays within PLAN_HBM_FRACTION of the usable HBM (0 when even full checkpointing does not: the pre-size step decides)."""
    shape = (nparam, es, hidden, inter, nl, tok_pad, world, partitioned, block_inputs)
    free = 0
    while free < nl and modelled_peak_bytes(free + 1, *shape) <= PLAN_HBM_FRACTION * hbm_usable:
        free += 1
    return free


def plan_checkpointing(hbm_usable, nparam, es, hidden, inter, nl, tok_pad, world=1, multi=False, partition_mode="auto",
                       per_block_control=True, block_inputs=2):
    """-> (checkpointed blocks: 0 = none, -1 = all for encoders without per-block control, else the first k; partition the
    optimizer state?).  `auto` partitions exactly when the replicated state would force blocks to be checkpointed."""
    shape = (nparam, es, hidden, inter, nl, tok_pad, world)
    partition = multi and partition_mode == "on"
    if multi and partition_mode == "auto":
        partition = world > 1 and plan_free_blocks(hbm_usable, *shape, partitioned=False, block_inputs=block_inputs) < nl
    free = plan_free_blocks(hbm_usable, *shape, partitioned=partition, block_inputs=block_inputs)
    ckpt = nl - free
    if not per_block_control:
        ckpt = 0 if free == nl else -1
    return ckpt, partition


def presize_is_tight(peak, hbm_usable, oom=False):
    """The worst-case step ran out of memory, or left less than 6 % of the usable HBM to spare."""
    return bool(oom or peak > PRESIZE_TIGHT_FRACTION * hbm_usable)


def checkpoint_more(now_ckpt, nl):
    """A quarter more of the blocks, or None when every block is checkpointed already (nothing left to give)."""
    return None if now_ckpt >= nl else min(nl, now_ckpt + max(1, nl // 4))


def checkpoint_fewer(peak, hbm_usable, now_ckpt, nparam, es, hidden, inter, nl, tok_pad, block_inputs=2, reserve=0):
    """After the worst-case step has been MEASURED (peak bytes, with now_ckpt blocks checkpointed) and was not tight: how many blocks
    to checkpoint instead -- as many fewer as the model says fit under PLAN_HBM_FRACTION of the usable HBM with `reserve` bytes kept
    aside (the transposed d(gate|up) buffer the run is about to take).  The plan is made from a model that errs on the safe side
    (cfg 5 with one kept input per block: modelled 234 GiB, measured 204); this hands the difference back once, and the step is
    measured again afterwards.  Never below 0, never more than now_ckpt."""
    per_free = tok_pad * int(ACT_KEEP_FRACTION * es * (6.5 * hidden + 2 * inter)) - block_inputs * tok_pad * hidden * es
    if now_ckpt <= 0 or per_free <= 0:
        return max(0, now_ckpt)
    room = PLAN_HBM_FRACTION * hbm_usable - peak - reserve
    return now_ckpt - max(0, min(now_ckpt, int(room // per_free)))


def transposed_dgu_bytes(inter, tok_pad, es):
    """[2 ff, T] at the worst-case token count: the SwiGLU backward's transposed d(gate|up) (ops.SWIGLU_DGU_T)."""
    return 2 * inter * tok_pad * es


def admit_transposed_dgu(peak, need, hbm_usable, default_limit):
    """None: the buffer is within ops' static default, nothing to decide.  Else: allowed iff, with the worst-case peak known
    (MEASURED in bench.py; the safe-side model in `plan_for_encoder`), twice its size still leaves 10 % of the usable HBM free
    (cfg 5 on one GPU: 253 GiB of 287 used, 22 GiB more: no; at 8 GPUs with the optimizer state partitioned: yes).  A dynamic
    "is there room right now" rule inside ops let cfg 5 run out of memory in round 4 (gpurun_out/r5h)."""
    if need <= default_limit:
        return None
    return bool(peak + 2 * need < DGU_T_ROOM_FRACTION * hbm_usable)


def may_retry_after_oom(world):
    """An out-of-memory error of ONE rank inside a step that holds collectives cannot be agreed on afterwards (its peers sit
    in the step's all-gather / gradient reduce): with more than one rank it is fatal, `--share-gpu` rehearsals included
    (advisor, round 4: the retry was allowed there and would have mismatched the collectives)."""
    return world == 1


# ----------------------------------------------------------------------------------------------------------------------
# the plan for an encoder in hand (what `gradient_checkpointing_enable()` without an argument resolves to)
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class EncoderPlan:
    checkpoint_blocks: int        # the first k blocks are checkpointed (0: none)
    tokens: int                   # the padded token count the plan was made for (rounded up to 256)
    hbm_usable: int
    modelled_peak: int
    transposed_dgu_limit: int     # what ops.SWIGLU_DGU_T_MAX_BYTES may be raised to (the static default when not admitted)


def pad_tokens(tokens: int) -> int:
    return (int(tokens) + 255) // 256 * 256


def plan_for_shape(hbm_usable, nparam, es, hidden, inter, nl, tokens, world=1, partitioned=False, block_inputs=1,
                   dgu_default_limit=6 * 2 ** 30) -> EncoderPlan:
    """The plan from byte counts alone (host arithmetic; `plan_for_encoder` measures the inputs)."""
    tok = pad_tokens(tokens)
    free = plan_free_blocks(hbm_usable, nparam, es, hidden, inter, nl, tok, world, partitioned, block_inputs)
    peak = int(modelled_peak_bytes(free, nparam, es, hidden, inter, nl, tok, world, partitioned, block_inputs))
    need = transposed_dgu_bytes(inter, tok, es)
    ok = admit_transposed_dgu(peak, need, hbm_usable, dgu_default_limit)
    return EncoderPlan(nl - free, tok, int(hbm_usable), peak, need if ok else dgu_default_limit)


def plan_for_encoder(encoder, tokens: int, world: int = None, partitioned: bool = False, block_inputs: int = 1) -> EncoderPlan:
    """Measures what the plan needs -- free HBM of the encoder's device + what this process already holds, the parameter count
    and element size, the world size of torch.distributed -- and returns the plan for `tokens` padded tokens per micro-step
    (both towers).  On a CPU encoder there is nothing to measure: every block is checkpointed, as HF does.
    The measurement is LOCAL: with several ranks each plans from its own free HBM (no collective is issued from inside a forward;
    checkpointing changes no result and no collective, so ranks may differ by a block); a launcher that wants one plan for all ranks
    takes the minimum of the ranks' usable HBM first and passes explicit counts, as bench.py does."""
    import torch
    cfg = encoder.config
    nl = len(encoder.layers)
    p0 = next(encoder.parameters())
    if p0.device.type != "cuda":
        return EncoderPlan(nl, pad_tokens(tokens), 0, 0, 0)
    if world is None:
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    free_now, total = torch.cuda.mem_get_info(p0.device)
    hbm = usable_hbm(free_now, torch.cuda.memory_reserved(p0.device), total)
    nparam = sum(p.numel() for p in encoder.parameters())
    from . import ops
    return plan_for_shape(hbm, nparam, p0.element_size(), cfg.hidden_size, cfg.intermediate_size, nl, tokens, world, partitioned,
                          block_inputs, dgu_default_limit=ops.SWIGLU_DGU_T_DEFAULT_MAX_BYTES)
