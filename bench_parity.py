"""bench.py's CHECKER legs, outside the timed region: the oracle's step on the host cores (`cpu_baseline`, the
reported CPU baseline), and step-loss parity of the product's fast path against the float32 oracle with its stock-precision controls
(`step_parity`: the rule the GPU tests share; `headline_parity`: the same at the size the metric is quoted on).  Only bench.py and
the tests import this module; the oracle is used as the checker, never timed as the product."""
from __future__ import annotations

import os
import time

import numpy as np  # noqa: F401
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """(threads to use, how that was decided): the affinity mask, cut down to the cgroup CPU quota when there is one -- a GPU
    box hands one GPU's job a share of a many-core host, its affinity mask still lists every core, and running one thread
    per listed core under a 16-core quota only thrashes (a first version of this leg then ran for more than 7 minutes)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    how = f"affinity mask {n}"
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                  # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())           # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n = max(1, int(quota + 0.5))
        how += f", cgroup cpu quota {quota:.1f} -> {n} threads"
    return n, how


def cpu_baseline(model, cfg, temperature, sample=(20, 42, 8, 3), repeats=3, note=lambda m: None, budget_s=150.0,
                 long_sample=(256, 1024, 2, 1)):
    """The oracle's training micro-step (encoder fwd+bwd + scoring, float32, eager) on the host cores, on a bounded
    sample of the workload: Q_s queries padded to Lq_s tokens + Q_s * G_s passages padded to Lp_s tokens (right-padded rows
    of random length, first row full: the reference's padded batches) through the SAME architecture and weights; median of
    `repeats` timed steps after one untimed step of the same shape (the FASTEST of them is reported), on every core this process may use.  pairs/s is
    extrapolated linearly in (padded) tokens to the full-length pair -- optimistic for the CPU, the quadratic attention term is
    ignored -- and, from a second sample with longer rows, with that term fitted (`fit`)."""
    from oracle import encoder_ref as E
    Lq_s, Lp_s, Q_s, G_s = sample
    cores, how = usable_cores()
    torch.set_num_threads(cores)
    note(f"cpu baseline: {cores} threads ({how}), {_cpu_model()}")
    w = {k: v.detach().to("cpu", torch.float32).requires_grad_(True) for k, v in model.model.state_dict().items()}
    cd = cfg.to_dict()
    g = torch.Generator().manual_seed(7)
    lo, hi = (1000, cfg.vocab_size - 1000) if cfg.vocab_size > 4000 else (1, cfg.vocab_size)
    pad = cfg.pad_token_id if cfg.pad_token_id is not None else 0

    def mk(N, L):
        lens = torch.randint(L // 2, L + 1, (N,), generator=g)
        lens[0] = L
        m = (torch.arange(L)[None, :] < lens[:, None]).long()
        ids = torch.randint(lo, hi, (N, L), generator=g)
        return {"input_ids": ids * m + pad * (1 - m), "attention_mask": m}
    batch = {"query": mk(Q_s, Lq_s), "passage": mk(Q_s * G_s, Lp_s)}
    times, t_start = [], time.perf_counter()
    for i in range(repeats + 1):
        t0 = time.perf_counter()
        ref = oracle_step(w, cd, batch, temperature)
        dt = time.perf_counter() - t0
        note(f"cpu baseline: oracle step {i}{' (untimed warm-up)' if i == 0 else ''} {dt:.2f} s")
        if i:                                   # step 0 warms the allocator and the thread pool
            times.append(dt)
        elif dt > budget_s / 2:                 # a host this slow gets ONE measured step: the warm-up itself
            times.append(dt)
            break
        if time.perf_counter() - t_start > budget_s and times:
            break
    # the FASTEST timed step, not the median: the host is shared (round 5's review: 0.0446 -> 0.0316 pairs/s between two runs of
    # unchanged code), a noisy neighbour only ever adds time, and the best step is the CPU's capability -- the figure that flatters
    # the baseline, not the GPU
    med = min(times)
    # A second, LONGER-row sample fits the quadratic (attention) term the token-linear extrapolation ignores: the oracle's eager
    # attention materialises [N, heads, L, L] scores, so t = a * tokens + b * sum(L_pad^2) over the padded rows (what the reference
    # runs).  Two samples, two unknowns; one timed step of the second one (the pool and the allocator are warm by now).
    fit = None
    if long_sample is not None and time.perf_counter() - t_start < budget_s:
        Lq2, Lp2, Q2, G2 = long_sample
        b2 = {"query": mk(Q2, Lq2), "passage": mk(Q2 * G2, Lp2)}
        t0 = time.perf_counter()
        oracle_step(w, cd, b2, temperature)
        dt2 = time.perf_counter() - t0
        note(f"cpu baseline: oracle step on the long-row sample ({Q2} x {Lq2} + {Q2 * G2} x {Lp2} tokens) {dt2:.2f} s")
        tok1, quad1 = Q_s * (Lq_s + G_s * Lp_s), Q_s * (Lq_s ** 2 + G_s * Lp_s ** 2)
        tok2, quad2 = Q2 * (Lq2 + G2 * Lp2), Q2 * (Lq2 ** 2 + G2 * Lp2 ** 2)
        det = tok1 * quad2 - tok2 * quad1
        a = (med * quad2 - dt2 * quad1) / det
        bq = (tok1 * dt2 - tok2 * med) / det
        if a > 0 and bq >= 0:
            fit = {"a_s_per_token": a, "b_s_per_token2": bq, "long_sample_seconds": round(dt2, 3),
                   "long_sample": f"{Q2} queries x {Lq2} tok + {Q2 * G2} passages x {Lp2} tok"}
    return med, times, Q_s * (Lq_s + G_s * Lp_s), (cores, how), batch, ref, fit


PARITY_GRADS = ("embed_tokens.weight", "layers.0.self_attn.q_proj.weight", "layers.0.self_attn.k_proj.weight")


def oracle_step(w, cd, batch, temperature, block_checkpoint=False):
    """One float32 training micro-step of the oracle (oracle/encoder_ref.py: contrastive_step + backward) on a dict of
    float32 leaf weights: what `step_parity` compares against and what `cpu_baseline` times.  `block_checkpoint`: the same
    arithmetic with every block recomputed in backward instead of stored (the headline-size parity leg, on the device)."""
    from oracle import encoder_ref as E
    for t in w.values():
        t.grad = None
    loss, scores, q, p = E.contrastive_step(w, cd, batch, temperature, block_checkpoint=block_checkpoint)
    loss.backward()
    grads = {k: w[k].grad.clone() for k in PARITY_GRADS if k in w and w[k].grad is not None}
    return dict(loss=float(loss.detach()), scores=scores.detach().float(), q=q.detach(), p=p.detach(), grads=grads)


def step_parity(model, cfg, temperature, sample_batch, ref, device, dtype, block_checkpoint=False):
    """Step-loss parity on identical tokens: the cpu_baseline sample through (i) the product's fast path
    (packed tokens, hand-written attention, fused ops, storage dtype of the run) and two CONTROLS in the same storage dtype on
    the GPU: (ii) the oracle's own eager arithmetic (HF eager semantics, modeling.py:219), and (iii) the same encoder with
    PyTorch's stock flash-attention kernels both ways (`hand_attention = False`) -- the reference trains with
    attn_implementation="flash_attention_2" (scripts/train/run_contrastive.sh), and every flash-attention backward takes
    delta = rowsum(dO o O) from the bf16-ROUNDED output where eager autograd sums P dP inside the softmax backward, which
    shows on the q / k projection gradients only (`tools/grad_error_map.py`: 0.016-0.023 there against eager's 0.009, the
    same with PyTorch's kernels as with the hand-written ones).  All three are compared with the float32 oracle; the fast
    path passes when its error is at most 1.5x the LARGER control error (every statistic, the max included) (the tolerance IS the reduced-precision error of
    the stock paths, measured here, not a guessed constant).  Statistics: RMS and max of the cosine errors over all [Q, P]
    scores, the loss, and the relative error of two weight gradients."""
    from oracle import encoder_ref as E
    dev_batch = {k: {kk: vv.to(device) for kk, vv in v.items()} for k, v in sample_batch.items()}
    names = list(ref["grads"])
    params = dict(model.model.named_parameters())

    def stats(loss, scores, grads):
        dc = (scores.float().cpu() - ref["scores"]) * temperature
        out = {"loss": float(loss), "loss_abs_err": abs(float(loss) - ref["loss"]),
               "cos_rms_err": float(dc.pow(2).mean().sqrt()), "cos_max_err": float(dc.abs().max())}
        for n in names:
            gr = ref["grads"][n]
            out["grad_rel_err:" + n] = float((grads[n].float().cpu() - gr).norm() / gr.norm().clamp_min(1e-30))
        return out

    def product_run():
        out = model(**dev_batch)
        got = torch.autograd.grad(out["loss"], [params[n] for n in names]) if names else ()   # .grad buffers stay untouched
        return stats(out["loss"].detach(), out["scores"].detach(), dict(zip(names, got)))

    fast = product_run()
    controls = {}
    if getattr(model.model, "hand_attention", False) and dtype == torch.bfloat16:
        model.model.hand_attention = False
        try:
            controls["control_stock_flash"] = product_run()
        finally:
            model.model.hand_attention = True
    wd = {k: v.detach().to(device, dtype).requires_grad_(k in names) for k, v in model.model.state_dict().items()}
    loss_c, s_c = E.contrastive_step(wd, cfg.to_dict(), dev_batch, temperature, dtype=dtype, block_checkpoint=block_checkpoint)[:2]
    got_c = torch.autograd.grad(loss_c, [wd[n] for n in names]) if names else ()
    controls["control_stock_eager"] = stats(loss_c.detach(), s_c.detach(), dict(zip(names, got_c)))
    ctrl = {k: max(c[k] for c in controls.values()) for k in controls["control_stock_eager"] if k != "loss"}
    # floors: float32 round-off (both paths are then ~1e-7 on a cosine and the ratio of two round-off errors means nothing).
    # One factor, 1.5, for every statistic.  (Round 2 ran the max at 2.0 after a red run at a ratio of 1.7; its cause -- Q
    # pre-scaled and re-rounded in the backward kernels -- was removed afterwards, and the driver's run of that round measured
    # 1.04 on the max with the RMS ratio below 1, so the wider factor had nothing left to cover.)
    tol = {"cos_rms_err": 1.5 * ctrl["cos_rms_err"] + 5e-6, "cos_max_err": 1.5 * ctrl["cos_max_err"] + 5e-6,
           "loss_abs_err": 1.5 * max(ctrl["loss_abs_err"], ctrl["cos_rms_err"] / temperature) + 5e-6 / temperature}
    for n in names:
        tol["grad_rel_err:" + n] = 1.5 * ctrl["grad_rel_err:" + n] + 1e-4
    failed = [k for k, t in tol.items() if not fast[k] <= t]
    # The two controls must also agree WITH EACH OTHER: the stock-flash control runs the product's own encoder code around
    # PyTorch's attention kernels, so a defect in that shared code inflates the tolerance together with the fast path's error
    # and the rule above goes blind (round 3 found one exactly so: rotary frequencies rounded to bf16 by `module.to(bf16)`,
    # 27 x the eager control's cosine error on 4096-token rows, "pass").  The eager control is the oracle's own code and shares
    # nothing with the product.  Measured ratios flash / eager on healthy code: 1.0-1.3 (cosines), <= 1.7 (q-projection
    # gradient: flash attention takes delta from the rounded output); 2.5 is the line.
    if "control_stock_flash" in controls:
        cf, ce = controls["control_stock_flash"], controls["control_stock_eager"]
        failed += [f"controls_disagree:{k}" for k in cf if k not in ("loss", "loss_abs_err") and not cf[k] <= 2.5 * ce[k] + 1e-5]
    rnd = lambda d: {k: round(v, 7) for k, v in d.items()}
    # the loss rule, spelled out with this run's numbers: the loss is a mean over Q rows of (logsumexp_j s_ij - s_i,target) with
    # s = cos / T, so a cosine error e moves a logit by e / T and the loss by up to that; the controls' OWN loss errors scatter
    # (the row errors partly cancel in the mean), which is why the tolerance takes the larger of the two terms
    loss_rule = {"fast_path_loss_abs_err": fast["loss_abs_err"],
                 "control_loss_abs_err": {k: c["loss_abs_err"] for k, c in controls.items()},
                 "control_cos_rms_err_over_T": ctrl["cos_rms_err"] / temperature,
                 "tolerance": tol["loss_abs_err"],
                 "tolerance_is": "1.5 x max(largest control loss error, largest control cosine RMS error / T)",
                 "fast_over_largest_control_loss_err": fast["loss_abs_err"] / max(ctrl["loss_abs_err"], 1e-12)}
    loss_rule = {k: ({kk: round(vv, 7) for kk, vv in v.items()} if isinstance(v, dict) else round(v, 7) if isinstance(v, float) else v)
                 for k, v in loss_rule.items()}
    return {"oracle_f32_loss": round(ref["loss"], 6), "fast_path": rnd(fast), **{k: rnd(v) for k, v in controls.items()},
            "tolerance": rnd(tol), "loss_rule": loss_rule, "pass": not failed, "failed": failed,
            "rule": "fast-path error <= 1.5 x the larger error of the stock paths in the same storage dtype (eager attention; "
                    "PyTorch's flash-attention kernels), all against the float32 oracle on the same tokens and weights "
                    "(loss: 1.5 x max(control loss error, control cosine RMS error / T)); and the stock-flash control (product encoder "
                    "code + PyTorch's attention) within 2.5 x of the eager control (oracle code) on every statistic but the loss"}


def headline_parity(model, cfg, temperature, Lq, Lp, device, dtype, w_host=None, note=lambda m: None, seed=77):
    """Step-loss parity AT THE SIZE THE METRIC IS QUOTED ON (BASELINE.json configs[1]; reference modeling.py:206-238, 281-314):
    2 queries of <= Lq tokens + 6 passages of <= Lp tokens (G = 3, one FULL-LENGTH row each, the others of random length in
    [L/2, L]) through ALL blocks of the benchmarked model, with the weights the timed steps left behind.  The float32 oracle is
    oracle/encoder_ref.py's own code executed on the device (torch float32 matmul, TF32-style shortcuts off, every block under
    torch.utils.checkpoint: same arithmetic, recomputed instead of stored -- on 8 host cores one such step takes ~10 minutes);
    that execution is PINNED first to the host execution of the same code on the full-length query row (`oracle_pin`).  Then the
    rule of `step_parity` with its two bf16 controls.  The short sample next to it keeps the oracle on the host end to end."""
    from oracle import encoder_ref as E
    assert not torch.backends.cuda.matmul.allow_tf32
    t_start = time.perf_counter()
    g = torch.Generator().manual_seed(seed)
    lo, hi = (1000, cfg.vocab_size - 1000) if cfg.vocab_size > 4000 else (1, cfg.vocab_size)
    pad = cfg.pad_token_id if cfg.pad_token_id is not None else 0

    def mk(N, L):
        lens = torch.randint(L // 2, L + 1, (N,), generator=g)
        lens[0] = L
        m = (torch.arange(L)[None, :] < lens[:, None]).long()
        ids = torch.randint(lo, hi, (N, L), generator=g)
        return {"input_ids": ids * m + pad * (1 - m), "attention_mask": m}
    batch = {"query": mk(2, Lq), "passage": mk(6, Lp)}
    tot = int(batch["query"]["attention_mask"].sum() + batch["passage"]["attention_mask"].sum())
    if tot % 256 == 0:                      # the filler sequence of the packed path must have something to do
        m = batch["query"]["attention_mask"]
        last = int(m[1].sum()) - 1
        m[1, last] = 0
        batch["query"]["input_ids"][1, last] = pad
        tot -= 1
    cd = cfg.to_dict()
    sd = model.model.state_dict()
    torch.cuda.empty_cache()                # the timed steps' cached activation blocks go back: the oracle needs ~40 GB in one piece
    w_dev = {k: v.detach().to(device, torch.float32).requires_grad_(k in PARITY_GRADS) for k, v in sd.items()}
    # (1) pin: oracle code on the device (f32) == oracle code on the host (f32), full-length query row, all blocks
    row = {k: v[:1] for k, v in batch["query"].items()}
    if w_host is None:
        w_host = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items()}
    with torch.no_grad():
        t0 = time.perf_counter()
        e_host = E.embed({k: v.detach() for k, v in w_host.items()}, cd, row)
        t_host = time.perf_counter() - t0
        e_dev = E.embed(w_dev, cd, {k: v.to(device) for k, v in row.items()}).cpu()
    pin = float((e_host - e_dev).abs().max())
    note(f"headline parity: oracle on device vs on host, one {Lq}-token row through {cfg.num_hidden_layers} blocks: max |diff| {pin:.2e} "
         f"(host {t_host:.1f} s)")
    # (2) the float32 oracle step on the device, block-checkpointed
    dev_batch = {k: {kk: vv.to(device) for kk, vv in v.items()} for k, v in batch.items()}
    t0 = time.perf_counter()
    ref = oracle_step(w_dev, cd, dev_batch, temperature, block_checkpoint=True)
    torch.cuda.synchronize()
    t_oracle = time.perf_counter() - t0
    ref = {"loss": ref["loss"], "scores": ref["scores"].cpu(), "q": ref["q"].cpu(), "p": ref["p"].cpu(),
           "grads": {k: v.cpu() for k, v in ref["grads"].items()}}
    for t in w_dev.values():
        t.grad = None
    del w_dev
    torch.cuda.empty_cache()
    note(f"headline parity: float32 oracle step on the device {t_oracle:.1f} s; fast path + two bf16 controls ...")
    rep = step_parity(model, cfg, temperature, batch, ref, device, dtype, block_checkpoint=True)
    torch.cuda.empty_cache()
    pin_tol = 1e-4                          # float32 round-off of two summation orders through all blocks; the bf16 errors judged are >= 3e-4
    if not pin <= pin_tol:
        rep["pass"] = False
        rep["failed"] = rep["failed"] + [f"oracle_pin:{pin:.2e}>{pin_tol:.0e}"]
    lens = [int(x) for x in torch.cat([batch["query"]["attention_mask"].sum(-1), batch["passage"]["attention_mask"].sum(-1)])]
    rep["sample"] = {"queries": 2, "passages": 6, "row_lengths": lens, "longest_row": max(lens), "tokens": tot,
                     "blocks": cfg.num_hidden_layers, "weights": "the benchmarked model after the timed steps"}
    rep["oracle"] = ("oracle/encoder_ref.py executed on the device in float32 (no TF32), blocks checkpointed; pinned to its host "
                     "execution on the full-length query row")
    rep["oracle_pin"] = {"max_abs_diff_unit_embedding": float(f"{pin:.3g}"), "tolerance": pin_tol, "row_tokens": Lq,
                         "host_seconds": round(t_host, 2)}
    rep["seconds"] = round(time.perf_counter() - t_start, 1)
    return rep
