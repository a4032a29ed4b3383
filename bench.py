#!/usr/bin/env python3
"""bench.py -- query-passage pairs/sec of the contrastive training step on N MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch per GPU, i.e. a full training micro-step:
encode B queries + B*(1+K) passages (Llama-3.2-1B architecture, bf16, random init) -> last-token pool + L2 normalise
(HIP) -> [N > 1: RCCL all-gather of embeddings, overlapped with the query tower] -> similarity + InfoNCE (HIP)
-> backward (HIP scoring backward, encoder backward under PyTorch-ROCm, bucketed gradient all-reduce) -> clip +
AdamW (HIP, one launch).  value = N * B * (1 + K) * steps / time.  Weak scaling: per-GPU work is fixed.

Workload at N = 1: BASELINE.json configs[1] (q_len 1280, p_len 4096, K = 5, B = 8, in-batch negatives, T = 0.02).
The JSON line also carries `roofline` (dominant hand-written kernel of the timed step, HIP-event timed live),
`kernels` (every hand-written entry point in the step), `roofline_sweep` (the similarity+InfoNCE kernel on the
scaled shapes of SURVEY.md §8d where it is MFMA-bound) and `cpu_baseline` (the oracle's step on the host cores).
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The host driver of this pool supports dmabuf IPC only: without this RCCL's peer-to-peer setup fails with
# "hipIpcGetMemHandle: invalid argument".  The ROCm runtime reads it when it initialises, i.e. at the first HIP call, so it is
# set here, before torch is even imported (round 3 set it after torch.cuda.set_device: too late had the shell not exported it).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak
MFMA_F32_PEAK_TFLOPS = 157.3
INFINITY_CACHE_BYTES = 256 * 2 ** 20   # MI355X_MICROARCH.md: 256 MB memory-side cache in front of HBM
PMC_FILE = "profiles/r06_pmc_traffic.json"

# MFMA products an entry point EXECUTES per algorithmic product (recomputation it does by design), so that the line shows both
# rates: what the matrix pipe does and what the caller gets.
EXECUTED_OVER_ALGO = {
    "rpo_flash_attn_bwd": (7, 5, "two launches, no atomics: the dQ kernel recomputes S and dP (3 products: S, dP, dQ), the dK/dV "
                                 "kernel computes S, dP, dV, dK (4); 5 are algorithmic"),
}

WORKLOADS = {
    # name: (arch, B, K, Lq, Lp, temperature, dtype).  K = 1 + "rankpo": chosen/rejected pairs (BASELINE configs[3]).
    "cfg2": ("llama-3.2-1b", 8, 5, 1280, 4096, 0.02, "bf16"),
    "cfg1": ("bge-small", 8, 5, 128, 256, 0.02, "f32"),
    "cfg4": ("llama-3.2-1b", 8, 1, 1280, 4096, 0.1, "bf16"),      # RankPO: reference_free, sigmoid, beta 2.0
    "cfg5": ("llama-3-8b", 8, 5, 1280, 4096, 0.02, "bf16"),       # 8B contrastive (per-GPU part of configs[4])
    "cfg5r": ("llama-3-8b", 8, 1, 1280, 4096, 0.1, "bf16"),       # 8B RankPO + 0.5 x SFT (InfoNCE) loss: the other half of configs[4]
    "tiny": ("llama-tiny", 8, 5, 160, 512, 0.02, "bf16"),
    # not a training step: ModelForInference.encode + exact top-k search, measured by bench_inference.py (SURVEY §8 a13 + f3)
    "encode": ("llama-3.2-1b", 64, 0, 1280, 4096, 1.0, "bf16"),
    "encode-tiny": ("llama-tiny", 64, 0, 1280, 4096, 1.0, "bf16"),     # the same code path on a small model (rehearsals)
}


# ----------------------------------------------------------------------------------------------------------
# live HIP-event timing of the C entry points (events are recorded on the stream the kernels are launched on)
# ----------------------------------------------------------------------------------------------------------
def _es(dt):
    return 4 if dt == 0 else 2          # RPO_DT_F32 = 0; bf16 = 1 and fp16 = 2 are two bytes


_LAST_T = [0]
_ATTN_PAIRS = {}    # packed token count T (as the C entry points receive it) -> sum over sequences of len (len + 1) / 2


def hook_attn_tables():
    """Flash attention's algorithmic flops depend on the sequence lengths, which its C arguments only hold on the device.
    Every packed encoder pass builds its query-tile table from the host copy of the lengths it hands to the kernels --
    INCLUDING the filler sequence that rounds the packed token count up to a multiple of 256 (encoder.py,
    `pooled_last_token_multi`) -- so the bench records them there, keyed by T = sum(lens), the `T` argument of
    rpo_flash_attn_fwd / _bwd.  A miss is an error (`_algo`), not a silent 0 (round 1's driver line had exactly that)."""
    from rankpo_amd import ops
    real = ops.attn_tile_table

    def recording(lens, device, *a, **kw):
        n = np.asarray(lens, dtype=np.int64)
        _ATTN_PAIRS[int(n.sum())] = int((n * (n + 1) // 2).sum())
        _LAST_T[0] = int(n.sum())                     # the packed token count of the pass in flight (rpo_lastq_attn_* get no T)
        return real(lens, device, *a, **kw)
    ops.attn_tile_table = recording


def _attn_pairs(T):
    if T not in _ATTN_PAIRS:
        raise KeyError(f"bench: no sequence lengths registered for a flash-attention call with T={T} packed tokens "
                       f"(known: {sorted(_ATTN_PAIRS)}); its algorithmic flops would read 0")
    return _ATTN_PAIRS[T]


def _algo(name, a):
    """(algorithmic bytes, flops) of one call, from its C arguments (SURVEY.md §8d figures)."""
    if name == "rpo_infonce_fwd":
        Q, P, d, dt, mode = a[2], a[3], a[4], a[5], a[7]
        s = _es(dt)
        ns = Q * P if mode == 0 else P
        return (Q + P) * d * s + ns * s + 4 * Q + 4, 2 * (Q * P if mode == 0 else P) * d
    if name == "rpo_infonce_bwd":
        Q, P, d, dt, mode = a[5], a[6], a[7], a[8], a[10]
        qr, pr = a[12], a[14]
        s = _es(dt)
        ns = Q * P if mode == 0 else P
        return (Q + P) * d * s + ns * s + (qr + pr) * d * s, 2 * ((qr * P + pr * Q) if mode == 0 else 2 * P) * d
    if name == "rpo_pool_normalize_fwd":
        N, L, d, dt = a[4], a[5], a[6], a[7]
        return N * L * 8 + 2 * N * d * _es(dt), 3 * N * d
    if name == "rpo_pool_normalize_bwd":
        N, L, d, dt = a[4], a[5], a[6], a[7]
        dense = a[10] is not None
        return (N * L * d * _es(dt) if dense else N * d * _es(dt)) + 2 * N * d * _es(dt), 4 * N * d
    if name == "rpo_adamw_step":
        n, dt = a[5], a[6]
        return n * (2 * _es(dt) + 24), 12 * n
    if name == "rpo_sumsq_partial":
        n, dt = a[1], a[2]
        return n * _es(dt), 2 * n
    if name == "rpo_swiglu_fwd":
        n = a[3] * a[4]
        return 3 * n * _es(a[7]), 5 * n
    if name == "rpo_swiglu_bwd":
        n = a[6] * a[7]
        return (6 if a[5] is not None else 5) * n * _es(a[12]), 13 * n
    if name == "rpo_swiglu_bwd_t":
        n = a[7] * a[8]
        return (8 if a[6] is not None else 6) * n * _es(a[13]), 13 * n
    if name == "rpo_rope":
        rows, H, hd, dt = a[5], a[6], a[7], a[9]
        return 2 * rows * H * hd * _es(dt) + rows * hd * 4, 3 * rows * H * hd
    if name == "rpo_add_rmsnorm_fwd":
        rows, d, dt = a[7], a[8], a[9]
        return rows * d * _es(dt) * (4 if a[1] is not None else 2) + rows * 4, 4 * rows * d
    if name == "rpo_add_rmsnorm_bwd":
        rows, d, dt = a[7], a[8], a[9]
        return rows * d * _es(dt) * (4 if a[4] is not None else 3) + rows * 4, 8 * rows * d
    if name == "rpo_flash_attn_fwd":
        T, nh, nkv, hd = a[10], a[11], a[12], a[13]          # a[9] = tile_cols
        pairs = _attn_pairs(T)                            # causal (query, key) pairs of this packed batch
        fold = (2 * T * nh * hd + 4 * T * hd) if a[19] is not None else 0      # rotary fold: rotated q written back, tables read
        return 2 * T * (2 * nh + 2 * nkv) * hd + 4 * T * nh + fold, 4 * hd * pairs * nh
    if name == "rpo_flash_attn_bwd":
        T, nh, nkv, hd = a[18], a[19], a[20], a[21]        # a[13] = q_tile_cols, a[16] = key_block, a[17] = sweep_down
        pairs = _attn_pairs(T)
        fold = 4 * T * hd if a[31] is not None else 0                          # rotary fold: cos / sin rows read by the epilogues
        return 2 * T * (4 * nh + 4 * nkv) * hd + 12 * T * nh + fold, 10 * hd * pairs * nh
    if name in ("rpo_lastq_attn_fwd", "rpo_lastq_attn_bwd"):
        N, nh, nkv, hd = a[7], a[8], a[9], a[10]
        T = _LAST_T[0]
        kvb = 2 * T * nkv * hd * 2                                   # K and V rows, read once
        if name.endswith("fwd"):
            return kvb + 2 * N * nh * hd * 2 + 4 * N * nh, 4 * T * nh * hd
        return 2 * kvb + 4 * N * nh * hd * 2 + 4 * N * nh, 10 * T * nh * hd
    if name == "rpo_transpose":
        rows, cols, dt = a[2], a[3], a[6]
        return 2 * rows * cols * _es(dt), 0
    if name in ("rpo_topk_merge", "rpo_topk_merge_split"):
        rows, cols, k, dt = a[2], a[3], a[5], a[6]
        return rows * cols * _es(dt) + 2 * rows * k * 12, rows * cols
    if name == "rpo_sim_topk_filter":           # the fused search step: both operands once, a threshold per row; survivors are a few dozen per row
        Q, P, d = a[2], a[3], a[4]
        return (Q + P) * d * 2 + 12 * Q, 2 * Q * P * d
    if name == "rpo_sim_scores_f32":
        Q, P, d = a[2], a[3], a[4]
        return (Q + P) * d * 2 + 4 * Q * P, 2 * Q * P * d
    if name == "rpo_topk_merge_candidates":     # counters + the winners in and out (+ the few candidates)
        rows, k = a[3], a[5]
        return rows * (4 + 2 * k * 12), 0
    if name == "rpo_rankpo_fwd":
        B, d, dt = a[4], a[5], a[6]
        return 3 * B * d * _es(dt), 4 * B * d
    if name == "rpo_rankpo_bwd":
        B, d, dt = a[4], a[5], a[6]
        return 6 * B * d * _es(dt), 6 * B * d
    return 0, 0


class TimedLib:
    """Proxy of the ctypes library that brackets every kernel entry point with HIP events while `enabled`."""

    def __init__(self, real):
        self._real = real
        self.enabled = False
        self.records = []     # (name, start_event, end_event, bytes, flops)

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("rpo_") or name in ("rpo_version", "rpo_status_string", "rpo_last_hip_error", "rpo_infonce_workspace_bytes", "rpo_add_rmsnorm_waves",
                                               "rpo_sim_topk_filter_ok", "rpo_build_flags"):
            return fn

        def wrapped(*a):
            if not self.enabled:
                return fn(*a)
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            rc = fn(*a)
            e1.record(s)
            b, f = _algo(name, a)
            hd = a[13] if name == "rpo_flash_attn_fwd" else a[21] if name == "rpo_flash_attn_bwd" else 0
            self.records.append((name, e0, e1, b, f, hd))
            return rc
        return wrapped

    def summary(self):
        agg = {}
        for name, e0, e1, b, f, hd in self.records:
            ms = e0.elapsed_time(e1)
            d = agg.setdefault(name, dict(calls=0, ms=0.0, bytes=0, flops=0))
            if hd:
                d["head_dim"] = hd
            d["calls"] += 1
            d["ms"] += ms
            d["bytes"] += b
            d["flops"] += f
        out = []
        for name, d in agg.items():
            avg_us = 1e3 * d["ms"] / d["calls"]
            gbs = d["bytes"] / d["calls"] / (avg_us * 1e-6) / 1e9
            tfs = d["flops"] / d["calls"] / (avg_us * 1e-6) / 1e12
            # the roofline that bounds the entry point: MFMA when its flops / byte exceed the machine balance
            bound = "mfma" if d["flops"] * HBM_PEAK_GBS * 1e9 > d["bytes"] * MFMA_BF16_PEAK_TFLOPS * 1e12 else "hbm"
            row = dict(entry=name, calls=d["calls"], avg_us=round(avg_us, 2), total_ms=round(d["ms"], 3),
                       algo_bytes=d["bytes"] // d["calls"], algo_flops=d["flops"] // d["calls"], bound=bound,
                       achieved_GBs=round(gbs, 2), frac_hbm=round(gbs / HBM_PEAK_GBS, 5),
                       achieved_TFLOPs=round(tfs, 2), frac_mfma=round(tfs / MFMA_BF16_PEAK_TFLOPS, 5))
            if bound == "hbm" and d["bytes"] // d["calls"] <= INFINITY_CACHE_BYTES:
                # the working set of one call fits the 256 MB Infinity Cache (and a producer has usually just written it): the
                # bytes per second of such a call are cache traffic, not HBM traffic -- cfg 1's AdamW read 7.0 TB/s "of 8" in
                # round 3 -- so the fraction is labelled for what it is
                row["cache_resident"] = True
                row["frac_hbm_note"] = "working set <= 256 MB Infinity Cache: this rate is not HBM traffic"
            if name in EXECUTED_OVER_ALGO:
                ex, al, why = EXECUTED_OVER_ALGO[name]
                row.update(executed_TFLOPs=round(tfs * ex / al, 2), frac_mfma_executed=round(tfs * ex / al / MFMA_BF16_PEAK_TFLOPS, 5),
                           executed_note=f"{ex} MFMA products executed per {al} algorithmic: {why}")
            if d.get("head_dim"):
                row["head_dim"] = d["head_dim"]
            out.append(row)
        out.sort(key=lambda r: -r["total_ms"])
        return out


# ----------------------------------------------------------------------------------------------------------
# first contact with N GPUs: every rank says where it is, and a rank that sits in one phase past its budget ends the job
# ----------------------------------------------------------------------------------------------------------
class Watchdog:
    """Per-rank phase log + dead-man switch.  `phase(name, budget_s)` prints `[bench rR HH:MM:SS] phase: name` to stderr (every
    rank: when an N-GPU run dies, the last line of each rank says where) and re-arms a timer; a daemon thread ends the process
    with exit code 3 and a line naming the phase when the phase outlives its budget.  A collective that never completes (a rank
    that died, a mismatched sequence of collectives, an RCCL bootstrap that cannot reach a peer) otherwise sits until the
    driver's own limit kills the job without a word; with this -- and the process-group timeout next to it -- the run ends
    non-zero within minutes, naming the phase.  No re-exec, no signal games: `os._exit` of this process only; under
    torch.distributed.run the agent then ends the other ranks."""

    def __init__(self, rank, scale=1.0, exit_fn=None):
        self.rank, self.scale = rank, float(scale)
        self._lock = threading.Lock()
        self._name, self._t0, self._budget = "start", time.monotonic(), None
        self._exit = exit_fn or (lambda code: os._exit(code))
        self.history = []
        if self.scale > 0:
            threading.Thread(target=self._run, name="bench-watchdog", daemon=True).start()

    def phase(self, name, budget_s):
        now = time.monotonic()
        with self._lock:
            self.history.append((self._name, round(now - self._t0, 2)))
            self._name, self._t0 = name, now
            self._budget = budget_s * self.scale if self.scale > 0 else None
        print(f"[bench r{self.rank} {time.strftime('%H:%M:%S')}] phase: {name}"
              + (f" (budget {budget_s * self.scale:.0f} s)" if self.scale > 0 else ""), file=sys.stderr, flush=True)

    def done(self):
        with self._lock:
            self.history.append((self._name, round(time.monotonic() - self._t0, 2)))
            self._name, self._budget = "done", None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self._lock:
                name, t0, budget = self._name, self._t0, self._budget
            if budget is not None and time.monotonic() - t0 > budget:
                print(f"[bench r{self.rank} {time.strftime('%H:%M:%S')}] WATCHDOG: phase '{name}' has run {time.monotonic() - t0:.0f} s, "
                      f"past its budget of {budget:.0f} s -- this rank is stuck (or a peer is, and this rank waits for it in a "
                      f"collective); exiting with code 3", file=sys.stderr, flush=True)
                self._exit(3)
                return


def llama_step_flops(cfg, lens_per_micro_batch):
    """Algorithmic FLOP of training micro-steps of a Llama encoder on REAL tokens (forward + backward = 3 x forward for the
    GEMMs, 3.5 x for causal attention: backward recomputes nothing algorithmically but has 5 products for the forward's 2):
      required: what the loss needs -- every block but the last on all tokens; the last block's K / V projections on all tokens
                and everything else of it (q, o, MLP, one-query attention) on the POOLED rows only (last-token pooling,
                modeling.py:224-230, consumes one row per sequence);
      model:    the reference's own count: all blocks on all tokens (what HF's LlamaModel executes before pooling).
    Gradient-checkpoint recomputation, pad tokens and the packed path's filler sequence are not algorithmic and not counted.
    Embedding lookup / its scatter-add backward: no GEMM.  Returns dict(gemm_required, attn_required, gemm_model, attn_model)."""
    d, nh = cfg.hidden_size, cfg.num_attention_heads
    nkv = getattr(cfg, "num_key_value_heads", None) or nh
    hd = getattr(cfg, "head_dim", None) or d // nh
    ff, nl = cfg.intermediate_size, cfg.num_hidden_layers
    kv = 2 * d * nkv * hd                                   # k_proj + v_proj weights
    blk = d * nh * hd + kv + nh * hd * d + 3 * d * ff       # all linear weights of one block
    out = dict(gemm_required=0, attn_required=0, gemm_model=0, attn_model=0)
    for lens in lens_per_micro_batch:
        n = np.asarray(lens, dtype=np.int64)
        T, rows, pairs = int(n.sum()), int(len(n)), int((n * (n + 1) // 2).sum())
        out["gemm_model"] += 6 * T * blk * nl
        out["gemm_required"] += 6 * (T * blk * (nl - 1) + T * kv + rows * (blk - kv))
        out["attn_model"] += int(3.5 * 4 * hd * nh * pairs) * nl
        out["attn_required"] += int(3.5 * 4 * hd * nh * pairs) * (nl - 1) + int(3.5 * 4 * hd * nh * T)
    return out


# ----------------------------------------------------------------------------------------------------------
def build_config(arch):
    from rankpo_amd import encoder as PE
    if arch == "llama-3.2-1b":
        return PE.llama_3_2_1b_config()
    if arch == "llama-3-8b":
        return PE.llama_3_8b_config()
    if arch == "bge-small":
        return PE.bge_small_config()
    if arch == "llama-tiny":
        return PE.llama_config(vocab_size=32000, hidden_size=512, intermediate_size=1536, num_hidden_layers=4,
                               num_attention_heads=8, num_key_value_heads=4, pad_token_id=0)
    raise ValueError(arch)


def synth_batch(cfg, B, K, Lq, Lp, seed, device):
    """SURVEY.md §8d synthetic inputs: ids in [1000, vocab-1000), right padded, lengths in [L/2, L], one full row."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lo, hi = (1000, cfg.vocab_size - 1000) if cfg.vocab_size > 4000 else (1, cfg.vocab_size)
    pad = cfg.pad_token_id if cfg.pad_token_id is not None else 0

    def side(N, L):
        ids = torch.randint(lo, hi, (N, L), generator=g)
        lens = torch.randint(L // 2, L + 1, (N,), generator=g)
        lens[0] = L
        m = (torch.arange(L)[None, :] < lens[:, None]).long()
        ids = ids * m + pad * (1 - m)
        return {"input_ids": ids.to(device), "attention_mask": m.to(device)}
    out = {"query": side(B, Lq), "passage": side(B * (1 + K), Lp)}
    return out


def sweep(lib_timed, device):
    """Similarity + InfoNCE forward (rpo_infonce_fwd: tile kernel + finalize) on the scaled shapes where it is
    MFMA-bound, called through the C ABI with preallocated buffers, 3 interleaved rounds."""
    from rankpo_amd import _lib
    lib = _lib.load()
    res = []
    st = torch.cuda.current_stream().cuda_stream
    # SURVEY.md 8d asks Q = P in {1024, 4096, 16384} x d in {2048, 4096}; 2048^2 and 8192^2 at d = 2048 complete the curve
    for Q, d in ((1024, 2048), (2048, 2048), (4096, 2048), (8192, 2048), (16384, 2048), (1024, 4096), (4096, 4096), (16384, 4096)):
        P = Q
        q = torch.nn.functional.normalize(torch.randn(Q, d, device=device), dim=-1).to(torch.bfloat16)
        p = torch.nn.functional.normalize(torch.randn(P, d, device=device), dim=-1).to(torch.bfloat16)
        scores = torch.empty(Q, P, device=device, dtype=torch.bfloat16)
        lse = torch.empty(Q, device=device)
        loss = torch.empty((), device=device)
        nws = lib.rpo_infonce_workspace_bytes(Q, P, d, 1)
        ws = torch.empty(nws, dtype=torch.uint8, device=device)
        call = lambda: lib.rpo_infonce_fwd(q.data_ptr(), p.data_ptr(), Q, P, d, 1, 0.02, 0, scores.data_ptr(),
                                           lse.data_ptr(), loss.data_ptr(), ws.data_ptr(), nws, st)
        for _ in range(5):
            assert call() == 0
        torch.cuda.synchronize()
        best = None
        reps = 20 if Q <= 8192 else 8
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            best = ms if best is None else min(best, ms)
        fl = 2.0 * Q * P * d
        entry = dict(kernel="rpo_infonce_fwd", Q=Q, P=P, d=d, dtype="bf16", ms=round(best, 4),
                     achieved_TFLOPs=round(fl / best / 1e9, 1),
                     frac_mfma=round(fl / best / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4))
        entry.update(_sweep_spot_check(q, p, scores, lse, 0.02))
        entry.update(_sweep_backward(lib, q, p, scores, lse, 0.02, best, st))
        res.append(entry)
        del q, p, scores, ws
    return res


def _sweep_backward(lib, q, p, scores, lse, temperature, fwd_ms, st):
    """The scoring BACKWARD at the sweep point (SURVEY 8d counts the path as 6 Q P d FLOP forward + backward): rpo_infonce_ds
    (dS and dS^T from the stored scores + lse) and the two products dq = dS p, dp = dS^T q -- timed with both arms of the products,
    the forward kernel's own MFMA frame (rpo_sim_gemm_nt on transposed embeddings; transposes inside the timed region) and
    hipBLASLt through torch.matmul; the arm `ops.INFONCE_BWD_GEMM` names is the one the product runs and `fwd_bwd_ms` counts.
    The two arms' gradients are compared (both round dS to bf16 first; they differ by accumulation order only)."""
    from rankpo_amd import ops
    Q, d = q.shape
    P = p.shape[0]
    gl = torch.ones((), device=q.device)
    ds = torch.empty(Q, P, device=q.device, dtype=torch.bfloat16)
    dst = torch.empty(P, Q, device=q.device, dtype=torch.bfloat16)
    out = {}

    def run(arm):
        assert lib.rpo_infonce_ds(scores.data_ptr(), lse.data_ptr(), gl.data_ptr(), Q, P, 1, temperature, 0, Q, 0, P,
                                  ds.data_ptr(), dst.data_ptr(), st) == 0
        if arm == "hip":
            out[arm] = (ops.sim_gemm_nt(ds, ops.transpose2d(p)), ops.sim_gemm_nt(dst, ops.transpose2d(q)))
        else:
            out[arm] = (ds @ p, dst @ q)
    times = {}
    reps = 10 if Q <= 8192 else 4
    for arm in ("hip", "blaslt"):
        for _ in range(2):
            run(arm)
        torch.cuda.synchronize()
    for _ in range(3):
        for arm in ("hip", "blaslt"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(arm)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            times[arm] = min(times.get(arm, ms), ms)
    dq_h, dp_h = out["hip"]
    dq_b, dp_b = out["blaslt"]
    rel = max(((dq_h.float() - dq_b.float()).abs().max() / dq_b.float().abs().max()).item(),
              ((dp_h.float() - dp_b.float()).abs().max() / dp_b.float().abs().max()).item())
    if not rel <= 2.0 ** -6:
        raise SystemExit(f"roofline sweep: the two arms of the scoring backward disagree at Q = P = {Q}, d = {d}: {rel:.3e}")
    arm = ops.INFONCE_BWD_GEMM
    if arm == "auto":
        arm = "hip" if min(Q, P) >= ops.INFONCE_BWD_HIP_MIN_K else "blaslt"
    fl = 2.0 * Q * P * d
    tot = fwd_ms + times[arm]
    return {"bwd_ms_hip": round(times["hip"], 4), "bwd_ms_blaslt": round(times["blaslt"], 4), "bwd_arm": arm,
            "bwd_arms_rel_diff": float(f"{rel:.3g}"),
            "frac_mfma_bwd_hip": round(2 * fl / times["hip"] / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4),
            "frac_mfma_bwd_blaslt": round(2 * fl / times["blaslt"] / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4),
            "fwd_bwd_ms": round(tot, 4), "frac_mfma_fwd_bwd": round(3 * fl / tot / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4)}


def _sweep_spot_check(q, p, scores, lse, temperature, rows=64):
    """The timed kernel's VALUES on one block of query rows from the middle of the matrix (advisor, round 4: the sweep is the
    only caller that runs the tile kernels' LDS-DMA rings 32-64 K-steps deep and it asserted rc == 0 only): scores against a
    float32 matmul with the kernel's two bf16 rounding points (dot -> bf16, / T -> bf16; <= 2 ulps, the tolerance of
    tests/test_gpu_kernels.py), lse against logsumexp of the RETURNED scores.  A failure raises: a fast wrong kernel is not a
    sweep point."""
    Q = q.shape[0]
    r0 = (Q // 2) // rows * rows
    raw = q[r0:r0 + rows].float() @ p.float().t()
    exp = (raw.to(torch.bfloat16).float() / temperature).to(torch.bfloat16).float()
    got = scores[r0:r0 + rows].float()
    ulps = ((got - exp).abs() / (exp.abs().clamp_min(1e-2) * 2.0 ** -7)).max().item()
    lse_err = (torch.logsumexp(got, dim=-1) - lse[r0:r0 + rows]).abs().max().item()
    if not (ulps <= 2.0 + 1e-6 and lse_err <= 2e-4):
        raise SystemExit(f"roofline sweep: rpo_infonce_fwd at Q = P = {Q}, d = {q.shape[1]} returns wrong values "
                         f"(scores {ulps:.2f} bf16 ulps off a float32 matmul, lse error {lse_err:.2e}, rows {r0}..{r0 + rows - 1})")
    return {"checked_rows": [r0, r0 + rows], "scores_max_ulps": round(ulps, 3), "lse_max_abs_err": float(f"{lse_err:.3g}")}


def attention_standalone(timed, cfg, batch_lens, device, reps=6):
    """The attention entry points of the timed steps, launched back to back ON THEIR OWN in this same process on the same packed
    shapes (every timed batch's 56 sequence lengths + the filler, rotary fold on, the encoder's own work lists), bracketed by the
    same HIP events: what a call costs when nothing else has run since the previous call, against what it cost inside the step
    (`kernels[]`).  Rounds 2-4 quoted stand-alone rates from other boxes and other shapes (48 passages, no rotary) and read the
    difference as an in-step loss of 12 % / 7 %."""
    import rankpo_amd.encoder
    from rankpo_amd import ops
    nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    W = (nh + 2 * nkv) * hd
    kb = ops.ATTN_KEY_BLOCK if hd == 64 else ops.ATTN_KEY_BLOCK_HD128
    scale = 1.0 / hd ** 0.5
    n0 = len(timed.records)
    was = timed.enabled
    for lens in batch_lens:
        lens = [int(x) for x in lens]
        fill = (-sum(lens)) % 256
        if fill:
            lens = lens + [fill]
        T = sum(lens)
        qkv = torch.randn(T, W, device=device).to(torch.bfloat16)
        pos = torch.cat([torch.arange(n) for n in lens]).to(device).float()
        fr = torch.outer(pos, 1.0 / (float(cfg.rope_theta) ** (torch.arange(0, hd, 2, device=device).float() / hd)))
        rope = (fr.cos().contiguous(), fr.sin().contiguous())
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=device)
        tiles = ops.attn_tile_table(lens, device, nh, nkv)
        ft = ops.attn_fwd_tile_table(lens, device, nh, nkv, hd) if rankpo_amd.encoder.FWD128_ONE_WAVE else None
        kt = ops.attn_key_tile_table(lens, device, nkv, kb)      # the defaults the encoder's packed path uses
        views = lambda t: (t[:, :nh * hd].unflatten(1, (nh, hd)), t[:, nh * hd:(nh + nkv) * hd].unflatten(1, (nkv, hd)),
                           t[:, (nh + nkv) * hd:].unflatten(1, (nkv, hd)))
        qv, kv_, vv = views(qkv)
        dqkv = torch.empty_like(qkv)
        go = torch.randn(T, nh, hd, device=device).to(torch.bfloat16)
        for on in (False, True):                                 # one untimed warm-up pair, then `reps` timed pairs
            timed.enabled = on
            for _ in range(reps if on else 1):
                out, lse = ops.flash_attn_varlen_fwd(qv, kv_, vv, cu, tiles if ft is None else ft, scale, rope=rope,
                                                     q_block=128 if ft is None else 64)
                ops.flash_attn_varlen_bwd(qv, kv_, vv, out, go, lse, cu, tiles, kt, scale, grads=views(dqkv), key_block=kb, rope=rope)
        torch.cuda.synchronize()
        del qkv, dqkv, go, out, lse
    timed.enabled = was
    recs, timed.records = timed.records[n0:], timed.records[:n0]
    res = {}
    for name in ("rpo_flash_attn_fwd", "rpo_flash_attn_bwd"):
        ms = [e0.elapsed_time(e1) for (n, e0, e1, *_rest) in recs if n == name]
        fl = [r[4] for r in recs if r[0] == name]
        res[name] = {"calls": len(ms), "avg_us": round(1e3 * sum(ms) / max(1, len(ms)), 2),
                     "frac_mfma": round(sum(fl) / max(1e-9, sum(ms) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 5)}
    return res


# the checker legs (CPU baseline of the oracle, step-loss parity against it) live in bench_parity.py
from bench_parity import (PARITY_GRADS, _cpu_model, usable_cores, cpu_baseline, oracle_step, step_parity,  # noqa: E402,F401
                          headline_parity)


def select_library(path):
    """`--lib`: make another build of librankpo_hip.so the one every op calls.  Importing the package has already loaded (and
    cached) the in-tree build, so changing `_lib.LIB_PATH` alone changes NOTHING -- round 3's `--lib` did exactly that, and its
    one in-step A/B (s_setprio around the forward's MFMA clusters: "no difference") compared the in-tree library with itself.
    The cache is dropped and the other file loaded; the line's `config.library` says which file ran."""
    from rankpo_amd import _lib
    path = os.path.abspath(path)
    if not os.path.exists(path):
        raise SystemExit(f"--lib {path}: no such file")
    _lib.LIB_PATH = path
    _lib._lib = None
    lib = _lib.load()
    assert os.path.samefile(lib._name, path), (lib._name, path)
    return lib


class _StdoutToStderr:
    """RCCL prints a version banner on stdout when the communicator is created; the driver wants ONE JSON line
    there.  Route file descriptor 1 to stderr while the process group comes up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


COMM_KEYS = ("backend", "ranks_seen", "world_size", "collectives_per_step", "allgather_calls_per_step", "allgather_bytes_per_rank",
             "allgather_wait_us", "allgather_wait_us_max", "allreduce_bytes", "allreduce_buckets", "allreduce_exposed_ms",
             "allreduce_exposed_ms_max", "late_buckets", "optimizer_state_partitioned", "param_allgather_exposed_ms", "loss_min_over_ranks", "loss_max_over_ranks", "loss_equal_over_ranks",
             "params_in_sync", "ms_per_step_min_over_ranks", "ms_per_step_max_over_ranks", "rebalance", "timer")


class CommProbe:
    """Brackets the two places where the launch stream WAITS for a collective -- `EmbeddingGather.wait` (the q||p all-gather,
    modeling.py:287-290 of the reference) and `FlatGradAllReducer.finish` (the bucketed gradient mean) -- with HIP events on the
    launch stream (wall-clock stamps in the CPU rehearsal): the bracket measures what the collective costs the step, i.e.
    its EXPOSED time; a collective that finished under the kernels queued before the wait reads ~0."""

    def __init__(self, use_events: bool):
        self.use_events = use_events
        self.enabled = False
        self.gather, self.reduce, self.param_gather = [], [], []          # (start, end) pairs
        self.gather_bytes = 0
        self.collectives = 0

    def _stamp(self):
        if self.use_events:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            return e
        return time.perf_counter()

    def install(self):
        from rankpo_amd import distributed as D
        probe = self
        from rankpo_amd import train_step as TS
        g_init, g_wait, r_finish = D.EmbeddingGather.__init__, D.EmbeddingGather.wait, D.FlatGradAllReducer.finish
        o_wait = TS.FlatAdamW._wait_gathers

        def wait_gathers(o, works):
            if not probe.enabled:
                return o_wait(o, works)
            a = probe._stamp()
            o_wait(o, works)
            probe.param_gather.append((a, probe._stamp()))
        TS.FlatAdamW._wait_gathers = wait_gathers
        real_ar, real_ag, real_agt = dist.all_reduce, dist.all_gather, dist.all_gather_into_tensor

        def count(fn):
            def counted(*a, **k):
                if probe.enabled:
                    probe.collectives += 1
                return fn(*a, **k)
            return counted
        dist.all_reduce, dist.all_gather, dist.all_gather_into_tensor = count(real_ar), count(real_ag), count(real_agt)
        dist.reduce_scatter_tensor = count(dist.reduce_scatter_tensor)

        def init(g, x):
            g_init(g, x)
            if probe.enabled:
                probe.gather_bytes = g.x.numel() * g.x.element_size()

        def wait(g):
            if not probe.enabled:
                return g_wait(g)
            a = probe._stamp()
            out = g_wait(g)
            probe.gather.append((a, probe._stamp()))
            return out

        def finish(r):
            if not probe.enabled:
                return r_finish(r)
            a = probe._stamp()
            out = r_finish(r)
            probe.reduce.append((a, probe._stamp()))
            return out
        D.EmbeddingGather.__init__, D.EmbeddingGather.wait, D.FlatGradAllReducer.finish = init, wait, finish

    def _ms(self, pairs):
        if self.use_events:
            return [a.elapsed_time(b) for a, b in pairs]
        return [1e3 * (b - a) for a, b in pairs]


def comm_block(probe, device, reducer, last_loss, elapsed, steps, flat_param=None, balance=None, segments=None):
    """The N > 1 part of the JSON line (also emitted by the world-1 `--force-dist` rehearsal and, on gloo, by
    `--rehearse-launch`): what the communicator saw and what the collectives cost, measured in this run.  Collective calls:
    every rank must call this."""
    def red(x, op):
        t = torch.tensor([float(x)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=op)
        return t.item()
    ones = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(ones)                                              # what the backend's communicator spans
    g_us = [1e3 * m for m in probe._ms(probe.gather)]
    r_ms = probe._ms(probe.reduce)
    p_ms = probe._ms(probe.param_gather)
    loss = float(last_loss)
    lmin, lmax = red(loss, dist.ReduceOp.MIN), red(loss, dist.ReduceOp.MAX)
    in_sync = None
    if flat_param is not None:                                        # replicas still hold the same parameters after the steps
        # a checksum of the parameter BITS (no float temporary: the 8B model's flat buffer is 15 GB): integer sum of the words
        bits = flat_param.view({1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[flat_param.element_size()])
        cs = bits.sum(dtype=torch.int64).reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(lo.item() == hi.item())
    ms = 1e3 * elapsed / max(1, steps)
    mean = lambda v: (sum(v) / len(v)) if v else None
    rnd = lambda v, n=3: None if v is None else round(v, n)
    return {
        "backend": dist.get_backend(), "ranks_seen": int(round(ones.item())), "world_size": dist.get_world_size(),
        "collectives_per_step": round(probe.collectives / max(1, steps), 2),
        "allgather_calls_per_step": round(len(probe.gather) / max(1, steps), 2), "allgather_bytes_per_rank": probe.gather_bytes,
        "allgather_wait_us": rnd(mean(g_us), 1), "allgather_wait_us_max": rnd(max(g_us) if g_us else None, 1),
        "allreduce_bytes": reducer.flat.numel() * reducer.flat.element_size(), "allreduce_buckets": len(reducer.buckets),
        "allreduce_exposed_ms": rnd(mean(r_ms)), "allreduce_exposed_ms_max": rnd(max(r_ms) if r_ms else None),
        "late_buckets": reducer.late_buckets, "optimizer_state_partitioned": bool(getattr(reducer, "shard", False)),
        "param_allgather_exposed_ms": rnd(mean(p_ms)) if p_ms else 0.0,
        "loss_min_over_ranks": lmin, "loss_max_over_ranks": lmax, "loss_equal_over_ranks": bool(lmin == lmax),
        "params_in_sync": in_sync,
        "ms_per_step_min_over_ranks": rnd(red(ms, dist.ReduceOp.MIN)), "ms_per_step_max_over_ranks": rnd(red(ms, dist.ReduceOp.MAX)),
        # groups of the global batch re-dealt to the ranks by packed-token cost (distributed.rebalance_groups): the most loaded
        # rank's cost over the mean, as the per-rank batches came and as they ran, averaged over the timed micro-steps
        "rebalance": (None if not balance else {
            "micro_steps": len(balance),
            "max_over_mean_as_sampled": rnd(mean([max(b["cost_before"]) * len(b["cost_before"]) / sum(b["cost_before"]) for b in balance]), 4),
            "max_over_mean_as_run": rnd(mean([max(b["cost_after"]) * len(b["cost_after"]) / sum(b["cost_after"]) for b in balance]), 4),
            # the timed region's segments (bench.py --balance auto with N > 1: first half re-dealt, second half as sampled), each
            # bracketed by a barrier, max over ranks: the mechanism's effect measured inside ONE run
            **({"segments": segments} if segments else {})}),
        "timer": ("HIP events on the launch stream around EmbeddingGather.wait / FlatGradAllReducer.finish" if probe.use_events
                  else "host clock (CPU rehearsal)"),
    }


def _under_profiler():
    """rocprofv3 (and rocprof) preload their tool library into the program they run; with counters it initialises the GPU before
    main() starts.  Spawning rank processes from such a process is the exec-after-GPU-init hop this pool forbids."""
    pre = os.environ.get("LD_PRELOAD", "")
    return ("rocprof" in pre.lower() or any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ))


def self_launch(n):
    """Spawn `n` rank processes of this script under torch.distributed.run (one per GPU, RCCL rendezvous on 127.0.0.1) and
    relay their output; rank 0 prints the JSON line.  Returns the launcher's exit code."""
    import socket
    import subprocess
    if _under_profiler():
        print("bench.py: refusing to start rank processes from under a profiler (its preloaded tool library may already have "
              "initialised the GPU in this process).  Profile the single-process form instead: "
              "rocprofv3 ... -- python3 bench.py --gpus 1 [--force-dist]", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.run(cmd, env=env).returncode


def rehearse_launch(rank, world, args):
    """What every rank does around the timed region, without a GPU: process group up (gloo, with the bench's timeout), barrier,
    MAX over ranks, and the `comm` block of the N > 1 line built by the same code from a toy replica (one all-gather + a
    bucketed gradient mean per step through the product's own EmbeddingGather / FlatGradAllReducer), under the same per-rank
    phase log and watchdog as the real run (`--rehearse-stall R`: rank R stops answering in step 1)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    wd = Watchdog(rank, args.watchdog_scale)
    wd.phase("process group init (gloo)", args.pg_timeout + 60)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.pg_timeout))
    dist.barrier()
    wd.phase("rehearsal steps", 40)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    from rankpo_amd.distributed import EmbeddingGather, FlatGradAllReducer, rebalance_groups
    probe = CommProbe(use_events=False)
    probe.install()
    torch.manual_seed(0)
    lin = torch.nn.Linear(16, 4)
    red_ = FlatGradAllReducer(list(lin.parameters()), bucket_mb=1e-4)
    probe.enabled = True
    steps, t0, loss, bal = 3, time.perf_counter(), None, []
    for i in range(steps):
        # token batches whose size grows with the rank, re-dealt by cost as the N > 1 bench does before every micro-step
        Lq, Lp = 4 + rank, 8 + 4 * rank
        qm = (torch.arange(Lq)[None, :] < torch.tensor([[Lq], [2]])).long()
        pm = (torch.arange(Lp)[None, :] < torch.tensor([[Lp], [3], [Lp - 1], [1]])).long()
        if i == 1 and rank == args.rehearse_stall:
            time.sleep(3600)                                       # a rank that stopped answering (the watchdog test)
        _, _, info = rebalance_groups({"input_ids": qm * 7, "attention_mask": qm}, {"input_ids": pm * 9, "attention_mask": pm})
        bal.append(info)
        x = torch.full((6, 16), float(rank + i + 1))
        emb = lin(x)
        allrows = EmbeddingGather(emb).wait()                    # [W * 6, 4]
        loss = allrows.detach().pow(2).mean()                    # over the gathered rows: the same value on every rank
        red_.arm()
        (emb.pow(2).mean()).backward()
        scale = red_.finish()
        with torch.no_grad():
            for p_ in lin.parameters():
                p_ -= 0.1 * scale * p_.grad
        red_.zero_()
    probe.enabled = False
    wd.phase("comm block", 60)
    flat = torch.cat([p_.detach().reshape(-1) for p_ in lin.parameters()])
    comm = comm_block(probe, torch.device("cpu"), red_, loss, time.perf_counter() - t0, steps, flat_param=flat, balance=bal)
    if rank == 0:
        print(json.dumps({"rehearsal": "launch", "n_gpus": world, "max_over_ranks": t.item(), "comm": comm}), flush=True)
    wd.phase("final barrier", 60)
    dist.barrier()
    dist.destroy_process_group()
    wd.done()


# ----------------------------------------------------------------------------------------------------------
# Memory guard: the checkpointing plan and its corrections as PURE functions of (usable bytes, measured peak, model shape,
# world), so that they are tested without a GPU (tests/test_host_logic.py::test_memory_guard_*; round 4's cfg-5 run died of a
# decision that only a GPU run could exercise).  bench.main feeds them what the device reports.
# ----------------------------------------------------------------------------------------------------------
# The HBM plan (which blocks are checkpointed, whether the transposed d(gate|up) buffer is admitted, the OOM retry rule) lives in
# the PACKAGE -- rankpo_amd/memory.py; `ModelForTraining.gradient_checkpointing_enable()` resolves to it for every user -- and is
# driven from here explicitly: plan -> MEASURED worst-case step -> one correction (`checkpoint_fewer`, `admit_transposed_dgu`).
from rankpo_amd.memory import (PLAN_HBM_FRACTION, PRESIZE_TIGHT_FRACTION, DGU_T_ROOM_FRACTION, ACT_KEEP_FRACTION,  # noqa: E402,F401
                               WORKING_SET_BLOCKS, usable_hbm, optimizer_state_bytes, modelled_peak_bytes, plan_free_blocks,
                               plan_checkpointing, presize_is_tight, checkpoint_more, checkpoint_fewer, transposed_dgu_bytes,
                               admit_transposed_dgu, may_retry_after_oom)


def inference_main(args, device, wd, rank, arch):
    """--workload encode: ModelForInference.encode and the exact top-k search, ONE JSON line (bench_inference.py)."""
    import bench_inference as BI
    from rankpo_amd import _lib
    from rankpo_amd.encoder import build_encoder
    cfg = build_config(arch)
    torch.manual_seed(0)
    with torch.device(device):
        enc = build_encoder(cfg)
    enc = enc.to(torch.bfloat16)
    timed = TimedLib(_lib.load())
    _lib._lib = timed
    hook_attn_tables()
    note = lambda m: print(f"[bench r{rank} {time.strftime('%H:%M:%S')}] {m}", file=sys.stderr, flush=True)
    small = arch != "llama-3.2-1b"
    wd.phase("encode()", 1500)
    enc_block = BI.encode_block(cfg, enc, device, note, reps=max(1, args.steps // 2), n_query=128 if small else 256,
                                n_passage=128 if small else 256, oracle=not args.no_cpu_baseline, cores=usable_cores()[0])
    wd.phase("search", 600)
    search = BI.search_block(device, timed, note, ntotal=100_000 if small else 1_000_000)
    wd.done()
    pas = enc_block["passages"]
    line = {"metric": "sentences/sec of ModelForInference.encode (passages <= 4096 tokens, batch 64), tokeniser included",
            "value": pas["end_to_end"]["sentences_per_s"], "unit": "sentences/s", "n_gpus": 1, "steps": max(1, args.steps // 2),
            "warmup": 1, "ms_per_step": round(1e3 * pas["end_to_end"]["seconds"] / (pas["sentences"] / pas["batch_size"]), 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"encode: {arch} architecture, bf16, 256 queries <= 1280 tokens + 256 passages <= 4096 tokens in batches "
                                   "of 64 (scripts/evaluate/run_evaluate.sh); search: 10^6 x 2048 bf16 corpus, k = 100",
                       "note": "NOT the headline metric of BASELINE.json (that is the default --workload cfg2); a step = one batch of 64"},
            "roofline": {"bound": "mfma", "achieved": pas["pre_tokenised"]["achieved_TFLOPs"], "peak": MFMA_BF16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": pas["pre_tokenised"]["frac_mfma"], "traffic": None,
                         "of": "the whole forward (library GEMMs + hand-written attention + fused elementwise) on pre-tokenised passages, "
                               "algorithmic FLOP on real tokens"},
            "cpu_baseline": enc_block.get("cpu_baseline"), "encode": enc_block, "search": search}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--ckpt-layers", type=int, default=-2,
                    help="checkpoint the first k blocks (-2 = as few as fit in 72%% of HBM, -1 = all, 0 = none)")
    ap.add_argument("--padded", action="store_true", help="run the encoder on padded batches (reference behaviour)")
    ap.add_argument("--no-fill", action="store_true", help="A/B: no filler sequence rounding the packed token count to 256")
    ap.add_argument("--no-linear-tn", action="store_true", help="A/B: torch's own operand layout for the input-gradient GEMMs")
    ap.add_argument("--no-wgrad-mixed", action="store_true", help="A/B: weight-gradient GEMMs as autograd issues them")
    ap.add_argument("--no-prod-t", action="store_true",
                    help="A/B: SwiGLU backward writes the recomputed product row-major (round 2-3) instead of transposed (ops.SWIGLU_PROD_T)")
    ap.add_argument("--no-dgu-t", action="store_true",
                    help="A/B: no transposed copy of d(gate|up) from the SwiGLU backward (ops.SWIGLU_DGU_T)")
    ap.add_argument("--wgrad-split", type=int, default=None,
                    help="A/B: chunks of the token reduction for the q|k|v and o weight gradients (ops.WGRAD_SPLIT_T; 1 = one GEMM)")
    ap.add_argument("--fold-rope", type=int, default=2, choices=(0, 1, 2),
                    help="A/B: 2 = rotary folded into the attention forward (q) and backward epilogues, 1 = backward only, 0 = separate passes")
    ap.add_argument("--fwd128", default="onewave", choices=("onewave", "classic"),
                    help="A/B: head_dim-128 attention forward: the one-wave-per-SIMD kernel on its own 64-query x 4-head list, or the 128-query kernel of rounds 3-4")
    ap.add_argument("--ckpt-inputs", type=int, default=1, choices=(1, 2),
                    help="A/B: tensors a checkpointed block keeps as its input: 1 = x + delta formed in front of the checkpoint (round 5), 2 = the (x, delta) pair of rounds 2-4")
    ap.add_argument("--recompute-output", action="store_true",
                    help="A/B: a recomputed (checkpointed) block also recomputes its own OUTPUT -- the SwiGLU product and the down projection -- as rounds 2-4 did (ops.SKIP_RECOMPUTED_OUTPUT = False)")
    ap.add_argument("--lib", default=None,
                    help="A/B: another build of librankpo_hip.so (tools/exp/build_variant.sh) instead of the in-tree one, for an A/B of "
                         "kernel variants INSIDE the training step (stand-alone kernel A/Bs have ranked schedules the step did not)")
    ap.add_argument("--dkdv-heaviest-first", action="store_true",
                    help="A/B: round 1's dK/dV work list (heaviest blocks first, ascending sweep) instead of group order with a heaviest-first tail")
    ap.add_argument("--dkdv-tail", type=float, default=None,
                    help="A/B: fraction of every XCD's dK/dV work that runs heaviest-first behind the group-ordered part (ops.ATTN_GROUP_TAIL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the N > 1 code path (RCCL init, cross-device gather, grad all-reduce) at N = 1")
    ap.add_argument("--gas", type=int, default=1,
                    help="micro-batches per optimizer step (the reference's scripts use 4; negatives are per micro-batch). "
                         "A timed step is then `gas` micro-steps + one gradient all-reduce + one AdamW launch")
    ap.add_argument("--partition-optimizer", default="auto", choices=("auto", "on", "off"),
                    help="N > 1: optimizer state partitioned over the ranks (reduce-scatter + AdamW shard + parameter all-gather, "
                         "the reference's ZeRO-1) instead of replicated (all-reduce); auto = on when the replicated state would "
                         "force blocks to be checkpointed")
    ap.add_argument("--balance", default="auto", choices=("auto", "on", "off"),
                    help="N > 1: re-deal the (query + its passages) groups of the global batch to the ranks by packed-token cost "
                         "before every micro-step (one small all-gather of token ids; the global batch, loss and gradient are "
                         "unchanged): every step ends in a gather all ranks wait at, so it lasts as long as the rank with the most "
                         "tokens.  auto = on when there is more than one rank")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r uses device r %% device_count and the process group "
                         "runs on gloo (RCCL refuses two ranks on one device); exercises the whole N > 1 code path -- launch, "
                         "cross-device forward with this rank's row window, bucketed gradient mean, comm block -- on the HIP kernels; "
                         "the numbers it prints are not a measurement")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="CPU rehearsal of the N-rank launch only: gloo process group, barrier, max-over-ranks, one JSON line")
    ap.add_argument("--rehearse-stall", type=int, default=-1,
                    help="with --rehearse-launch: this rank stops answering in the second step (a test of the watchdog: the job "
                         "must end non-zero, naming the phase, instead of hanging)")
    ap.add_argument("--pg-timeout", type=float, default=180.0,
                    help="seconds a collective may take before torch.distributed aborts the process group (default 180; "
                         "torch's own default is 10 minutes for nccl, 30 for gloo: longer than a driver's patience)")
    ap.add_argument("--watchdog-scale", type=float, default=1.0,
                    help="multiplies every phase budget of the per-rank watchdog (0 = no watchdog, phases are still logged)")
    ap.add_argument("--attn-standalone", action="store_true",
                    help="after the timed steps: the attention entry points back to back on their own, on the timed batches' shapes "
                         "(same process, same events) -> `attention_in_step_vs_standalone` in the line")
    ap.add_argument("--memory-summary", default=None,
                    help="write torch.cuda.memory_summary() + the allocator's totals after the allocator pre-size step to this file "
                         "(what the worst-case peak is made of: profiles/r05_cfg5_memory_summary.txt)")
    ap.add_argument("--headline-parity", default="auto", choices=("auto", "on", "off"),
                    help="N = 1: second step-parity sample at the headline size (2 queries <= q_len + 6 passages <= p_len tokens "
                         "through ALL blocks, float32 oracle on the device pinned to the host); auto = on for the Llama-3.2-1B "
                         "workloads, where it fits beside the training state")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: start N fresh rank processes under torch.distributed.run.  This parent has
        # made no GPU call (importing torch does not initialise HIP) and only waits for the children: a process that has
        # touched the GPU is never re-exec'd.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or run `python bench.py --gpus {args.gpus}` outside torchrun")
    if args.rehearse_launch:
        return rehearse_launch(rank, world, args)
    if args.share_gpu:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    wd = Watchdog(rank, args.watchdog_scale)
    wd.phase("device", 300)                      # the first HIP call of a fresh box pages the runtime in
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        # a collective that cannot complete aborts the process group after --pg-timeout seconds (torch's own default would sit
        # for 10 minutes); the watchdog's budget for this phase lies behind it, so torch's message -- which names the
        # collective -- comes first when both apply
        wd.phase(f"process group init ({'gloo' if args.share_gpu else 'nccl = RCCL'}, world {world})", args.pg_timeout + 120)
        pg_timeout = datetime.timedelta(seconds=args.pg_timeout)
        with _StdoutToStderr():
            if args.share_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=pg_timeout)
            dist.barrier()                       # creates the RCCL communicator now (banner goes to stderr)
            torch.cuda.synchronize()
    wd.phase("build model", 600)

    import rankpo_amd
    from rankpo_amd import _lib
    if args.lib:
        select_library(args.lib)
    from rankpo_amd.encoder import build_encoder
    from rankpo_amd.train_step import TrainStep

    arch, B, K, Lq, Lp, temperature, dtn = WORKLOADS[args.workload]
    if args.workload.startswith("encode"):
        if world != 1:
            raise SystemExit("--workload encode measures one GPU (encode() shards trivially: run N processes on N slices of the corpus)")
        return inference_main(args, device, wd, rank, arch)
    dtype = torch.bfloat16 if dtn == "bf16" else torch.float32
    cfg = build_config(arch)
    torch.manual_seed(0)                      # identical weights on every rank (data parallel replicas)
    with torch.device(device):
        enc = build_encoder(cfg)
    enc = enc.to(dtype)
    if args.no_fill:
        enc.pack_fill = False
    if args.no_linear_tn:
        rankpo_amd.ops.LINEAR_TN = False
    if args.no_wgrad_mixed:
        rankpo_amd.ops.WGRAD_MIXED = False
    if args.no_prod_t:
        rankpo_amd.ops.SWIGLU_PROD_T = False
    if args.no_dgu_t:
        rankpo_amd.ops.SWIGLU_DGU_T = False
    if args.wgrad_split is not None:
        rankpo_amd.ops.WGRAD_SPLIT_T = args.wgrad_split
    rankpo_amd.encoder.FOLD_ROPE = args.fold_rope
    rankpo_amd.encoder.FWD128_ONE_WAVE = args.fwd128 == "onewave"
    rankpo_amd.encoder.CKPT_SINGLE_INPUT = args.ckpt_inputs == 1
    rankpo_amd.ops.SKIP_RECOMPUTED_OUTPUT = not args.recompute_output
    if args.dkdv_heaviest_first:
        rankpo_amd.ops.ATTN_SWEEP_DOWN = rankpo_amd.ops.ATTN_SWEEP_DOWN_HD128 = False
    if args.dkdv_tail is not None:
        rankpo_amd.ops.ATTN_SWEEP_DOWN = rankpo_amd.ops.ATTN_SWEEP_DOWN_HD128 = True
        rankpo_amd.ops.ATTN_GROUP_TAIL = args.dkdv_tail
    model = rankpo_amd.ModelForTraining(encoder=enc, temperature=temperature, use_inbatch_neg=True,
                                        negatives_cross_device=multi, unpad=not args.padded).train()
    hook_attn_tables()
    wd.phase("synthetic batches", 300)
    gas = max(1, args.gas)
    nb = (args.steps + args.warmup) * gas
    batches = [synth_batch(cfg, B, K, Lq, Lp, 1234 + rank * 1000 + i, device) for i in range(nb)]
    # real (unpadded) row lengths of every micro-batch, on the host: token counts, the step's algorithmic FLOP, the ranks' costs
    batch_lens = [torch.cat([b["query"]["attention_mask"].sum(-1), b["passage"]["attention_mask"].sum(-1)]).tolist() for b in batches]
    tok_real = [int(sum(l)) for l in batch_lens]
    tok_pad = B * Lq + B * (1 + K) * Lp
    ckpt = args.ckpt_layers
    es = 2 if dtype == torch.bfloat16 else 4
    nl = cfg.num_hidden_layers
    nparam = sum(p.numel() for p in enc.parameters())

    # HBM this rank can really use: what the device reports free NOW -- after RCCL created its communicator and buffers, with
    # whatever else lives on the card -- plus what this process already holds (the bf16 parameters), MIN over the ranks, so that
    # every rank derives the same checkpointing plan (round 3 budgeted 0.72 of the nominal 288 GB whatever was free)
    free_now, total_hbm = torch.cuda.mem_get_info(device)
    hbm_usable = usable_hbm(free_now, torch.cuda.memory_reserved(device), total_hbm, world if args.share_gpu else 1)
    if multi:
        hu = torch.tensor([float(hbm_usable)], dtype=torch.float64, device=device if not args.share_gpu else "cpu")
        dist.all_reduce(hu, op=dist.ReduceOp.MIN)
        hbm_usable = int(hu.item())
    # the plan itself: pure functions above (tested without a GPU)
    plan_ckpt, partition = plan_checkpointing(hbm_usable, nparam, es, cfg.hidden_size, cfg.intermediate_size, nl, tok_pad,
                                              world=world, multi=multi, partition_mode=args.partition_optimizer,
                                              per_block_control=hasattr(enc, "layers"), block_inputs=args.ckpt_inputs)
    if ckpt == -2:
        ckpt = plan_ckpt
    if ckpt != 0 and hasattr(enc, "layers"):
        model.gradient_checkpointing_enable(layers="all" if ckpt < 0 else ckpt)
    elif ckpt != 0:
        model.gradient_checkpointing_enable(layers="all")

    timed = TimedLib(_lib.load())
    if not args.no_kernel_timing:
        _lib._lib = timed
    probe = None
    if multi:
        probe = CommProbe(use_events=True)
        probe.install()
    balance = multi and (args.balance == "on" or (args.balance == "auto" and world > 1))
    # `auto` with more than one rank: the FIRST half of the timed steps runs with the re-deal, the second half without, inside
    # this one run -- a single N-GPU record then carries the mechanism's gain (or loss) instead of a default nobody measured
    split_balance = multi and args.balance == "auto" and world > 1 and args.steps >= 2
    bal_state = {"on": balance}
    bal_log = []
    pad_id = cfg.pad_token_id if cfg.pad_token_id is not None else 0

    def dealt(b):
        """This rank's micro-batch, after the global batch's groups were re-dealt by packed-token cost (inside the timed step)."""
        if not bal_state["on"]:
            return b
        from rankpo_amd.distributed import rebalance_groups
        q, p, info = rebalance_groups(b["query"], b["passage"], pad_id)
        if probe is not None and probe.enabled:
            bal_log.append(info)
        return {"query": q, "passage": p}
    rankpo_wl = args.workload in ("cfg4", "cfg5r")
    if rankpo_wl:
        # RankPO stage (rankpo_trainer.py:570-587): policy = the bare encoder, no reference model (reference_free),
        # metrics stay on the device (one host copy per LOG step, not per micro-step)
        trainer = rankpo_amd.RankPOTrainer(enc, None, beta=2.0, temperature=temperature, loss_type="sigmoid",
                                           reference_free=True, rankpo_weight=1.0,
                                           sft_weight=0.5 if args.workload == "cfg5r" else 0.0)
        loss_fn = lambda b: trainer.get_batch_loss_metrics(enc, dealt(b), "train", sync_metrics=False)[0]
    else:
        loss_fn = lambda b: model(**dealt(b))["loss"]
    ts = TrainStep(model.parameters(), loss_fn, lr=1e-5, max_grad_norm=1.0,
                   gradient_accumulation_steps=gas, total_steps=max(10, args.steps + args.warmup), warmup_ratio=0.1,
                   force_collectives=args.force_dist, partition_optimizer=partition)
    micro = lambda i: batches[i] if gas == 1 else batches[i * gas:(i + 1) * gas]

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    note(f"model + {nb} synthetic batches ready ({arch}, world {world}); warmup {args.warmup} steps")
    losses = []
    mem_guard = {"hbm_usable_GiB": round(hbm_usable / 2 ** 30, 1), "hbm_total_GiB": round(total_hbm / 2 ** 30, 1),
                 "checkpointed_blocks_planned": (nl if ckpt < 0 else ckpt), "retries": 0}
    if not args.padded and args.warmup > 0:
        # Packed mode changes the activation sizes every step.  One untimed step on a full-length batch first sizes
        # the caching allocator for the worst case (later, smaller steps reuse/split those blocks instead of
        # calling hipMalloc inside the timed region) and proves that the worst case fits in HBM.
        full = {k: {"input_ids": v["input_ids"], "attention_mask": torch.ones_like(v["attention_mask"])}
                for k, v in batches[0].items()}
        wd.phase("allocator pre-size step (full-length batch)", 900)
        while True:
            # Memory guard: the plan above is an estimate.  If the worst-case step does not fit (out of memory), or fits with
            # less than 6 % of the usable HBM to spare on some rank, checkpoint more blocks and take the step again rather
            # than die in the timed region (cfg 5 at N = 1 peaked at 93 % in round 3).  Every rank must take the same
            # decision: the verdict is all-reduced (MAX); an OOM of one rank inside a step that holds collectives cannot be
            # agreed on afterwards, so with more than one rank it stays fatal (the watchdog names the phase).
            oom = False
            try:
                ts.step(full if gas == 1 else [full] * gas)
                torch.cuda.synchronize()
            except torch.OutOfMemoryError:
                if not may_retry_after_oom(world):
                    raise
                oom = True
                ts.abort_step()                  # the step died half way: gradients zeroed, reducer disarmed, no pending work
            peak = torch.cuda.max_memory_allocated(device)
            tight = presize_is_tight(peak, hbm_usable, oom)
            if multi:
                tg = torch.tensor([1.0 if tight else 0.0], device=device)
                dist.all_reduce(tg, op=dist.ReduceOp.MAX)
                tight = bool(tg.item() > 0)
            now_ckpt = nl if ckpt < 0 else ckpt
            more = checkpoint_more(now_ckpt, nl) if hasattr(enc, "layers") else None
            can_more = more is not None
            note(f"allocator pre-sized on a full-length batch: peak mem {peak / 2**30:.1f} GiB of {hbm_usable / 2**30:.1f} usable"
                 + (" -- OUT OF MEMORY" if oom else "") + (f"; tight: checkpointing more blocks ({now_ckpt} of {nl} so far)" if tight and can_more else ""))
            if not tight and not oom and hasattr(enc, "layers") and args.ckpt_layers == -2 and not mem_guard.get("loosened") \
                    and mem_guard["retries"] == 0 and now_ckpt > 0:
                # the measured peak leaves room under the plan's own budget: hand it back, once, and measure again
                dgu_need = transposed_dgu_bytes(getattr(cfg, "intermediate_size", 0), tok_pad, es)
                reserve = 2 * dgu_need if ("llama" in arch and not args.no_dgu_t and not args.no_prod_t
                                           and dgu_need > rankpo_amd.ops.SWIGLU_DGU_T_MAX_BYTES) else 0
                fewer = checkpoint_fewer(peak, hbm_usable, now_ckpt, nparam, es, cfg.hidden_size, cfg.intermediate_size, nl, tok_pad,
                                         args.ckpt_inputs, reserve)
                if multi:
                    fw = torch.tensor([float(fewer)], device=device)
                    dist.all_reduce(fw, op=dist.ReduceOp.MAX)
                    fewer = int(fw.item())
                mem_guard["loosened"] = now_ckpt - fewer
                if fewer < now_ckpt:
                    note(f"the measured peak leaves room under the plan: {fewer} blocks checkpointed instead of {now_ckpt}; measuring again")
                    ckpt = fewer
                    model.gradient_checkpointing_enable(layers=ckpt)
                    ts.opt.reducer.zero_()
                    torch.cuda.empty_cache()
                    torch.cuda.reset_peak_memory_stats(device)
                    continue
            if not tight or not can_more:
                if oom:
                    raise SystemExit(f"bench: the worst-case step does not fit in {hbm_usable / 2**30:.1f} GiB of HBM with every block checkpointed")
                break
            ckpt = more
            model.gradient_checkpointing_enable(layers=ckpt)
            mem_guard["retries"] += 1
            ts.opt.reducer.zero_()
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(device)
        # The transposed d(gate|up) of the SwiGLU backward (ops.SWIGLU_DGU_T) is taken up to 6 GiB by default; the Llama-3-8B shape
        # needs 13 GB.  With the worst-case peak MEASURED, it is allowed when twice its size still leaves 10 % of the usable HBM
        # free on every rank (cfg 5 on one GPU: 88 % used, no; at 8 GPUs with the optimizer state partitioned: yes), and the
        # pre-size step runs once more so that the allocator holds the buffer before the timed region.
        need = transposed_dgu_bytes(getattr(cfg, "intermediate_size", 0), tok_pad, es)
        ok = admit_transposed_dgu(torch.cuda.max_memory_allocated(device), need, hbm_usable, rankpo_amd.ops.SWIGLU_DGU_T_MAX_BYTES)
        if "llama" in arch and not args.no_dgu_t and not args.no_prod_t and ok is not None:
            if multi:
                okt = torch.tensor([0.0 if ok else 1.0], device=device)
                dist.all_reduce(okt, op=dist.ReduceOp.MAX)
                ok = bool(okt.item() == 0)
            if ok:
                rankpo_amd.ops.SWIGLU_DGU_T_MAX_BYTES = need
                wd.phase("allocator pre-size step again (transposed d(gate|up) enabled)", 900)
                ts.step(full if gas == 1 else [full] * gas)
                torch.cuda.synchronize()
            mem_guard["transposed_dgu_buffer_GiB"] = round(need / 2 ** 30, 1)
            mem_guard["transposed_dgu_enabled"] = bool(ok)
        if args.memory_summary and rank == 0:
            st_ = torch.cuda.memory_stats(device)
            with open(args.memory_summary, "w") as f:
                f.write(f"# {args.workload} ({arch}), world {world}, after the allocator pre-size step (every row at full length: {tok_pad} tokens)\n"
                        f"# usable HBM {hbm_usable / 2**30:.1f} GiB; peak allocated {st_['allocated_bytes.all.peak'] / 2**30:.1f} GiB, peak reserved "
                        f"{st_['reserved_bytes.all.peak'] / 2**30:.1f} GiB, allocated now {st_['allocated_bytes.all.current'] / 2**30:.1f} GiB "
                        f"(= states that outlive a step), reserved now {st_['reserved_bytes.all.current'] / 2**30:.1f} GiB\n"
                        f"# optimizer_state_bytes (model) {optimizer_state_bytes(nparam, es, world, ts.opt.partition) / 2**30:.1f} GiB; "
                        f"checkpointed blocks {nl if ckpt < 0 else ckpt} of {nl}; block inputs kept {tok_pad * cfg.hidden_size * es * nl / 2**30:.1f} GiB\n")
                f.write(torch.cuda.memory_summary(device))
        if "llama" in arch:
            run_ckpt = nl if ckpt < 0 else ckpt
            mem_guard["modelled_peak_GiB"] = round(modelled_peak_bytes(nl - run_ckpt, nparam, es, cfg.hidden_size, cfg.intermediate_size,
                                                                       nl, tok_pad, world, ts.opt.partition, args.ckpt_inputs) / 2 ** 30, 1)
            mem_guard["checkpoint_inputs_per_block"] = args.ckpt_inputs
        mem_guard.update(checkpointed_blocks_run=(nl if ckpt < 0 else ckpt),
                         presize_peak_GiB=round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
                         presize_peak_over_usable=round(torch.cuda.max_memory_allocated(device) / hbm_usable, 3))
    t_step = None
    for i in range(args.warmup):
        wd.phase(f"warm-up step {i}", 600)
        tw = time.perf_counter()
        losses.append(ts.step(micro(i)))
        torch.cuda.synchronize()
        t_step = time.perf_counter() - tw
        note(f"warmup step {i} done in {t_step:.2f} s, loss {float(losses[-1]):.4f}, peak mem {torch.cuda.max_memory_allocated(device) / 2**30:.1f} GiB")
    # the ranks' costs as the batches were sampled, for every timed micro-step (host data, one object gather BEFORE the timed
    # region): what the un-dealt half of a split run is judged by without adding a collective or a host sync to its steps
    cost_sampled = None
    if multi:
        from rankpo_amd.distributed import sequence_cost
        mine = [float(sequence_cost(torch.tensor(l, dtype=torch.float64)).sum()) for l in batch_lens]
        allc = [None] * world
        dist.all_gather_object(allc, mine)
        cost_sampled = [[allc[r][i] for r in range(world)] for i in range(nb)]
    # segments of the timed region: (re-deal on?, first step, one past the last step)
    first, last = args.warmup, args.warmup + args.steps
    if split_balance:
        mid = first + (args.steps + 1) // 2
        segments = [(True, first, mid), (False, mid, last)]
    else:
        segments = [(balance, first, last)]
    wd.phase(f"timed steps ({args.steps})", max(300.0, 60.0 + 4.0 * args.steps * (t_step or 30.0)))
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    timed.enabled = not args.no_kernel_timing
    if probe is not None:
        probe.enabled = True
    t0 = time.perf_counter()
    seg_times, t_local = [], 0.0
    for si, (on, a, b) in enumerate(segments):
        bal_state["on"] = on
        ts0 = time.perf_counter()
        for i in range(a, b):
            losses.append(ts.step(micro(i)))
        torch.cuda.synchronize()
        t_local += time.perf_counter() - ts0        # this rank's own steps (before it waits for the slowest rank)
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        seg_times.append(time.perf_counter() - ts0)
    elapsed = time.perf_counter() - t0
    timed.enabled = False
    bal_state["on"] = balance
    note(f"timed {args.steps} steps in {elapsed:.3f} s" + (f" (re-deal on: {seg_times[0]:.3f} s, off: {seg_times[1]:.3f} s)" if split_balance else ""))
    comm = None
    if multi:
        wd.phase("comm block", 300)
        probe.enabled = False
        tmax = torch.tensor([elapsed] + seg_times, dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, seg_times = tmax[0].item(), tmax[1:].tolist()
        mom = lambda rows: (sum(max(c) * len(c) / sum(c) for c in rows) / len(rows)) if rows else None
        seg_rep = []
        for (on, a, b), st in zip(segments, seg_times):
            sampled = [cost_sampled[i] for i in range(a * gas, b * gas)]
            seg_rep.append({"redeal": bool(on), "steps": b - a, "ms_per_step": round(1e3 * st / max(1, b - a), 3),
                            "pairs_per_s": round(world * B * (1 + K) * gas * (b - a) / st, 3),
                            "max_over_mean_cost_as_sampled": round(mom(sampled), 4),
                            # the most loaded rank's packed tokens per step, as sampled: the two halves run different batches
                            "cost_max_rank_mean_as_sampled": round(sum(max(c) for c in sampled) / len(sampled), 1)})
        comm = comm_block(probe, device, ts.opt.reducer, losses[-1], t_local, args.steps, flat_param=ts.opt.flat_param, balance=bal_log,
                          segments=seg_rep if (split_balance or balance) else None)
    pairs = world * B * (1 + K) * gas * args.steps
    peak_mem = torch.cuda.max_memory_allocated(device) / 2 ** 30
    # algorithmic FLOP of the timed steps, all ranks (the re-deal moves groups between ranks, the global sum stays)
    flops = None
    if "llama" in arch:
        fl = llama_step_flops(cfg, batch_lens[args.warmup * gas:(args.warmup + args.steps) * gas])
        if rankpo_wl and getattr(trainer, "ref_model", None) is not None:
            fl = {k: v + v // 3 for k, v in fl.items()}          # a reference model adds one forward (none here: reference_free)
        keys = sorted(fl)
        ft = torch.tensor([float(fl[k]) for k in keys], dtype=torch.float64, device=device)
        if multi:
            dist.all_reduce(ft, op=dist.ReduceOp.SUM)
        flops = dict(zip(keys, ft.tolist()))

    if rank == 0:
        kernels = [] if args.no_kernel_timing else timed.summary()
        out = {
            "metric": "query-passage pairs/sec", "value": round(pairs / elapsed, 3), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtn, "data": "synthetic",
            "config": {"workload": (f"{args.workload}: {arch} RankPO (reference_free, sigmoid, beta=2.0"
                                    + (", sft_weight=0.5" if args.workload == "cfg5r" else "") + f"), B={B}/GPU, "
                                    f"chosen+rejected, q_len={Lq}, p_len={Lp}, T={temperature}") if rankpo_wl else
                                   (f"{args.workload}: {arch} contrastive, B={B}/GPU, K={K}, q_len={Lq}, p_len={Lp}, "
                                    f"T={temperature}, in-batch negs" + (", cross-device negs" if multi else "")),
                       "global_batch": world * B, "pairs_per_step": world * B * (1 + K) * gas,
                       "parallelism": f"dp{world}" + (", groups re-dealt to ranks by packed-token cost every micro-step" if balance else ""),
                       "optimizer": f"AdamW(flat, HIP) + clip 1.0, GAS={gas}", "micro_steps_per_step": gas,
                       "optimizer_state": (f"partitioned over {world} ranks (reduce-scatter, AdamW on 1/{world}, parameter all-gather)"
                                           if ts.opt.partition else "replicated (bucketed all-reduce)"),
                       "grad_checkpointing": "all blocks" if ckpt < 0 else f"first {ckpt} blocks",
                       "padding": "padded batches" if args.padded else "pad tokens skipped (packed varlen encoder)",
                       "tokens_per_step_per_gpu": {"padded": tok_pad, "real_mean": int(sum(tok_real) / len(tok_real))},
                       "weights": "random init (seed 0)",
                       "library": os.path.relpath(_lib.LIB_PATH, ROOT),
                       "timed_region": ("every call of a hand-written entry point is bracketed by two HIP events on its launch stream INSIDE "
                                        "the timed steps (that is where `roofline` and `kernels` come from): `value` includes their cost "
                                        "(--no-kernel-timing: without them)") if not args.no_kernel_timing else
                                       "no per-call HIP events (--no-kernel-timing)",
                       "memory_guard": mem_guard,
                       "dropout": {"hidden": float(getattr(cfg, "hidden_dropout_prob", 0.0) or 0.0),
                                   "attention": float(getattr(cfg, "attention_probs_dropout_prob", getattr(cfg, "attention_dropout", 0.0)) or 0.0)}},
            "loss_first": round(float(losses[0]), 5), "loss_last": round(float(losses[-1]), 5),
            "peak_mem_GiB": round(peak_mem, 2),
        }
        if comm is not None:
            out["comm"] = comm
        if kernels:
            top = kernels[0]
            # HBM traffic per launch: PMC counters cannot be read from inside this process; the committed rocprofv3 passes
            # (profiles/r03_pmc_traffic.json, taken at the commit it names: 2 x FETCH_SIZE + WRITE_SIZE per kernel, separate
            # passes, tools/pmc_workload.py at the cfg-2 / cfg-5 shapes) give the traffic / algorithmic ratio of THIS entry
            # point's kernels, which is applied to this run's algorithmic bytes.
            traffic, tsrc = None, None
            try:
                pmc = json.load(open(os.path.join(ROOT, PMC_FILE)))
                key = top["entry"] + ("@hd128" if top.get("head_dim") == 128 else "")
                if key in pmc and pmc[key].get("traffic_over_algorithmic"):
                    ratio = pmc[key]["traffic_over_algorithmic"]
                    traffic = int(top["algo_bytes"] * ratio)
                    tsrc = (f"{PMC_FILE} (commit {pmc.get('_meta', {}).get('head')}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                            f"passes, FETCH x2 gfx950 correction, kernels {pmc[key]['kernels']} on {pmc[key]['shape']}: ratio "
                            f"{ratio:.3f} applied to this run's algorithmic bytes")
            except Exception:
                pass
            if top["bound"] == "mfma":
                out["roofline"] = {"kernel": top["entry"], "bound": "mfma", "achieved": top["achieved_TFLOPs"],
                                   "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": top["frac_mfma"],
                                   "traffic": traffic, "traffic_source": tsrc, "avg_us": top["avg_us"],
                                   "algo_flops": top["algo_flops"], "algo_bytes": top["algo_bytes"]}
                if "executed_TFLOPs" in top:
                    out["roofline"].update({k: top[k] for k in ("executed_TFLOPs", "frac_mfma_executed", "executed_note")})
            else:
                out["roofline"] = {"kernel": top["entry"], "bound": "hbm", "achieved": top["achieved_GBs"],
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": top["frac_hbm"], "traffic": traffic,
                                   "traffic_source": tsrc, "avg_us": top["avg_us"], "algo_bytes": top["algo_bytes"]}
            out["kernels"] = kernels
        if flops is not None:
            # north_star: "pairs/sec ... with achieved fraction of bf16 MFMA roofline" -- the WHOLE step against the dense bf16 peak
            # of the GPUs it ran on.  `frac` uses the FLOP the loss requires (last block on the pooled rows only); the
            # reference-model count (every block on every token, what HF's LlamaModel executes) is next to it.
            req = flops["gemm_required"] + flops["attn_required"]
            mod = flops["gemm_model"] + flops["attn_model"]
            hand_ms = sum(k_["total_ms"] for k_ in kernels) / args.steps if kernels else None
            step_ms = 1e3 * elapsed / args.steps
            out["step_roofline"] = {
                "bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_BF16_PEAK_TFLOPS * world,
                "achieved": round(req / elapsed / 1e12, 1), "frac": round(req / elapsed / 1e12 / (MFMA_BF16_PEAK_TFLOPS * world), 4),
                "algo_flops_per_step": int(req / args.steps), "gemm_flops_per_step": int(flops["gemm_required"] / args.steps),
                "attention_flops_per_step": int(flops["attn_required"] / args.steps),
                "gemm_share_of_flops": round(flops["gemm_required"] / req, 4),
                "model_flops_per_step": int(mod / args.steps), "achieved_model_flops": round(mod / elapsed / 1e12, 1),
                "frac_model_flops": round(mod / elapsed / 1e12 / (MFMA_BF16_PEAK_TFLOPS * world), 4),
                "counting": "real (unpadded) tokens of the timed steps, all ranks; GEMMs 6 x tokens x linear weights (forward + "
                            "backward), causal attention 3.5 x 4 hd nh sum len (len + 1) / 2 per block; algo = every block but the "
                            "last on all tokens + the last block's K/V projections on all tokens and the rest of it on the pooled "
                            "rows only; model = all blocks on all tokens (the reference's HF encoder).  Not counted: checkpoint "
                            "recomputation, pad tokens, the packed path's filler sequence, embedding lookups, the optimizer"}
            if hand_ms is not None and world == 1:
                # rank 0's HIP-event brackets: hand-written entry points vs everything else on the stream (hipBLASLt GEMMs +
                # PyTorch glue); the GEMM rate is a lower bound because the remainder is not only GEMM time
                lib_ms = step_ms - hand_ms
                out["step_roofline"].update({
                    "hand_written_kernel_ms_per_step": round(hand_ms, 2), "library_and_glue_ms_per_step": round(lib_ms, 2),
                    "library_gemm_TFLOPs_at_least": round(flops["gemm_required"] / args.steps / (lib_ms * 1e-3) / 1e12, 1),
                    "gemm_share_of_step_time_at_most": round(lib_ms / step_ms, 4)})
        if world == 1 and args.attn_standalone and kernels and "llama" in arch:
            wd.phase("attention entry points stand-alone on the timed shapes", 600)
            alone = attention_standalone(timed, cfg, batch_lens[args.warmup * gas:(args.warmup + args.steps) * gas], device)
            instep = {k_["entry"]: k_ for k_ in kernels}
            out["attention_in_step_vs_standalone"] = {
                name: {"in_step_avg_us": instep[name]["avg_us"], "in_step_frac_mfma": instep[name]["frac_mfma"],
                       "standalone_avg_us": a["avg_us"], "standalone_frac_mfma": a["frac_mfma"], "standalone_calls": a["calls"],
                       "in_step_over_standalone": round(instep[name]["avg_us"] / a["avg_us"], 4)}
                for name, a in alone.items() if name in instep}
        if world == 1 and not args.no_sweep:
            wd.phase("roofline sweep (similarity + InfoNCE kernel)", 600)
            out["roofline_sweep"] = sweep(timed, device)
            note("sweep done")
        if world == 1 and not args.no_cpu_baseline:
            wd.phase("cpu baseline (oracle on the host cores)", 1200)
            note("cpu baseline (oracle on host cores) ...")
            dt, times, toks, (cores, cores_how), sample_batch, ref, fit = cpu_baseline(
                model, cfg, temperature, note=note, long_sample=(256, 1024, 2, 1) if Lp >= 1024 else None)
            wd.phase("step parity (short sample, oracle on the host)", 900)
            note("step parity: fast path and stock-eager control vs the float32 oracle ...")
            # the timed steps ran with the configuration's dropout (BERT family: HF's 0.1, as the reference trains it); the
            # oracle has none, so the parity leg -- outside the timed region -- compares with every nn.Dropout at p = 0
            rankpo_amd.encoder.disable_dropout_in_model(model.model)
            out["step_loss_parity"] = step_parity(model, cfg, temperature, sample_batch, ref, device, dtype)
            toks_per_pair = Lp + Lq / (1 + K)
            nq_s, np_s = sample_batch["query"]["input_ids"].shape, sample_batch["passage"]["input_ids"].shape
            out["cpu_baseline"] = {"value": round(toks / dt / toks_per_pair, 5), "unit": "pairs/s", "cores": cores,
                                   "kind": "port", "cpu_model": _cpu_model(), "cores_from": cores_how,
                                   "step_seconds": [round(t, 3) for t in times], "best_seconds": round(dt, 3),
                                   "sample": f"oracle (eager torch f32, {cores} threads) fwd+bwd of {nq_s[0]} queries x {nq_s[1]} "
                                             f"tok + {np_s[0]} passages x {np_s[1]} tok (padded, as the reference runs them) "
                                             f"through the same {arch} weights: {toks} tokens, fastest of {len(times)} timed "
                                             f"steps after 1 untimed; pairs/s extrapolated linearly in tokens to "
                                             f"{toks_per_pair:.0f} tokens per full-length pair"}
            if fit is not None:
                # the full workload as the reference runs it (padded rows): B queries of Lq + B (1 + K) passages of Lp tokens
                t_full = fit["a_s_per_token"] * B * (Lq + (1 + K) * Lp) + fit["b_s_per_token2"] * B * (Lq ** 2 + (1 + K) * Lp ** 2)
                out["cpu_baseline"].update({
                    "value_with_attention_term": round(B * (1 + K) / t_full, 5),
                    "attention_term": {**{k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in fit.items()},
                                       "model": "seconds = a * padded tokens + b * sum over rows of L_pad^2 (the oracle's eager "
                                                "attention), fitted on the two samples, evaluated at the full batch "
                                                f"({B} x {Lq} + {B * (1 + K)} x {Lp} tokens): {t_full:.0f} s per step"}})
            try:        # what the port's time is worth in REFERENCE time (tools/time_reference.py, build container)
                rc = json.load(open(os.path.join(ROOT, "profiles", "ref_cpu_container.json")))
                # the ratio measured on THIS architecture family: the Llama-3.2-1B architecture on the very sample timed above
                # (weight-bound, short rows), or the BGE-small cfg-1 step for the BERT family
                leg = rc["full_step_llama_3_2_1b_sample"] if "llama" in arch else rc["full_step_cfg1"]
                por = leg["port_over_reference"]
                out["cpu_baseline"].update({
                    "port_over_reference": por,
                    "reference_equivalent_value": round(out["cpu_baseline"]["value"] * por, 5),
                    "port_over_reference_source": ("profiles/ref_cpu_container.json: the reference itself (imported, unmodified) and this "
                                                   f"port timed on the same step in the build container ({rc['host']['threads']} threads of "
                                                   f"{rc['host']['cpu_model']}; {leg['case'][:60]}...): reference {leg['reference_median_s']} s, "
                                                   f"port {leg['port_median_s']} s per step; the reference cannot travel to the GPU box")})
            except Exception:
                pass
            # second sample: the headline's own size (BASELINE.json configs[1]): full-length rows through ALL blocks
            want_hp = args.headline_parity == "on" or (args.headline_parity == "auto" and arch == "llama-3.2-1b"
                                                       and dtype == torch.bfloat16)
            if want_hp and "llama" in arch:
                wd.phase("step parity (headline size: full-length rows, all blocks, oracle on the device)", 1200)
                del ref
                hp = headline_parity(model, cfg, temperature, Lq, Lp, device, dtype, note=note)
                out["step_loss_parity"] = {"pass": bool(out["step_loss_parity"]["pass"] and hp["pass"]),
                                           "failed": out["step_loss_parity"]["failed"] + ["headline:" + f for f in hp["failed"]],
                                           "short_sample": out["step_loss_parity"], "headline_sample": hp,
                                           "samples": "short_sample: the cpu_baseline rows (oracle on the host end to end); "
                                                      "headline_sample: one full-length query and passage row among 2 + 6 rows "
                                                      "through all blocks (oracle on the device, pinned to the host)"}
            if not out["step_loss_parity"]["pass"]:
                print(json.dumps(out), flush=True)
                raise SystemExit("step_loss_parity FAILED: " + ", ".join(out["step_loss_parity"]["failed"]))
        print(json.dumps(out), flush=True)
    if multi:
        wd.phase("final barrier", 300 if world > 1 else 2400)
        dist.barrier()
        dist.destroy_process_group()
    wd.done()


if __name__ == "__main__":
    main()
