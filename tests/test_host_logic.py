"""CPU tests of the host side: the C-ABI library loads and exports every symbol of include/rankpo_hip.h, collators
reproduce the reference's output (golden), argument validation mirrors the reference's errors, the N > 1 gather and
gradient-reduction paths work on 2 gloo ranks.  No compute call into the HIP library happens here."""
import ctypes
import json
import math
import os
import random
import re

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    from rankpo_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "rankpo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rpo_[a-z0-9_]+)\s*\(", hdr))
    assert {"rpo_pool_normalize_fwd", "rpo_pool_normalize_bwd", "rpo_infonce_fwd", "rpo_infonce_bwd",
            "rpo_rankpo_fwd", "rpo_rankpo_bwd", "rpo_adamw_step", "rpo_infonce_workspace_bytes"} <= declared
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/rankpo_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes SIGNATURES and the header disagree"
    assert lib.rpo_version() >= 100
    assert lib.rpo_status_string(0) == b"ok"
    assert lib.rpo_infonce_workspace_bytes(8, 48, 2048, 1) >= 256
    assert ctypes.sizeof(_lib.RankPOParams) == 32


def test_missing_library_fails_loudly(monkeypatch):
    from rankpo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librankpo_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    from rankpo_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.infonce_loss(torch.randn(2, 8), torch.randn(4, 8), 0.02)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rankpo_loss_metrics(torch.randn(2, 8), torch.randn(4, 8), ops.RankPOConfig())


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "rankpo_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("oracle/", ""), f"{fn} mentions the oracle"


def test_collators_match_reference(golden):
    from rankpo_amd.data_utils import ContrastiveDataCollatorWithPadding, RankPODataCollatorWithPadding
    g = golden("collators")
    meta = json.loads(str(g["meta"]))
    o = RankPODataCollatorWithPadding(pad_token_id=128004)(meta["rankpo_examples"])
    for a in ("query", "passage"):
        for b in ("input_ids", "attention_mask"):
            np.testing.assert_array_equal(o[a][b].numpy(), g[f"rankpo_{a}_{b}"])
            assert o[a][b].dtype == torch.int64
    # the docstring example of the reference (data_utils.py:144-172)
    np.testing.assert_array_equal(o["passage"]["input_ids"].numpy(),
                                  [[4, 5, 128004], [6, 128004, 128004], [9, 128004, 128004], [10, 11, 12]])
    random.seed(meta["python_random_seed"])
    o = ContrastiveDataCollatorWithPadding(pad_token_id=meta["pad_token_id"],
                                           num_negatives=meta["num_negatives"])(meta["contrastive_features"])
    for a in ("query", "passage"):
        for b in ("input_ids", "attention_mask"):
            np.testing.assert_array_equal(o[a][b].numpy(), g[f"contrastive_{a}_{b}"])
    with pytest.raises(AssertionError, match="key: 'chosen' is missing"):
        RankPODataCollatorWithPadding()([{"query": {}, "rejected": {}}])


def test_model_for_training_argument_errors():
    """Same ValueErrors as modeling.py:189-196."""
    import rankpo_amd
    from rankpo_amd import encoder as PE
    cfg = PE.llama_config(vocab_size=32, hidden_size=16, intermediate_size=32, num_hidden_layers=1,
                          num_attention_heads=2, num_key_value_heads=1)
    with pytest.raises(ValueError, match="Temperature should be smaller than 1.0"):
        rankpo_amd.ModelForTraining(config=cfg, temperature=1.0)
    m = rankpo_amd.ModelForTraining(config=cfg, temperature=0.7, normalize_embeddings=False)
    assert m.temperature == 1.0                                   # reset when not normalizing (modeling.py:186-188)
    with pytest.raises(ValueError, match="Distributed training has not been initialized"):
        rankpo_amd.ModelForTraining(config=cfg, temperature=0.02, negatives_cross_device=True)
    with pytest.raises(ValueError, match="Cannot use fp16 and bf16"):
        rankpo_amd.ModelForInference(config=cfg, use_fp16=True, use_bf16=True)
    m = rankpo_amd.ModelForTraining(config=cfg, temperature=0.02)
    assert m.pooling_mode == "last" and m.embed(None) is None
    out = rankpo_amd.ModelOutput(loss=torch.tensor(1.0))
    assert list(out.keys()) == ["loss"] and out["loss"] == out.loss and out.scores is None


def test_rankpo_trainer_knobs_and_errors():
    import rankpo_amd
    tr = rankpo_amd.RankPOTrainer(None, None, beta=2.0, temperature=0.1, reference_free=True)
    c, r = torch.tensor([.8, .2, .5]), torch.tensor([.3, .6, .5])
    np.testing.assert_allclose(tr.rankpo_loss(c, r).numpy(), [4.5399e-05, 8.000335, 0.693147], rtol=1e-4)
    tr.loss_type = "hinge"
    np.testing.assert_allclose(tr.rankpo_loss(c, r).numpy(), [0, 9, 1], atol=1e-5)
    tr.loss_type = "bogus"
    with pytest.raises(ValueError, match="Unknown loss type: bogus"):
        tr.rankpo_loss(c, r)
    with pytest.raises(ValueError, match="Unknown loss type: bogus"):
        tr._cfg().to_c()
    tr.store_metrics({"rewards/chosen": 1.0})
    tr.store_metrics({"rewards/chosen": 3.0})
    assert tr.log({"loss": 0.5})["rewards/chosen"] == 2.0


def test_cosine_schedule_matches_transformers():
    from transformers import get_cosine_schedule_with_warmup
    from rankpo_amd.train_step import cosine_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=3, num_training_steps=20)
    for s in range(20):
        assert abs(sch.get_last_lr()[0] - cosine_with_warmup(s, 20, 3)) < 1e-9
        opt.step()
        sch.step()


# ------------------------------------------------------------------------------------------- 2-rank gloo
def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rankpo_amd.distributed import EmbeddingGather, FlatGradAllReducer, all_gather_with_local_grad
        g = np.load(os.path.join(ROOT, "tests", "golden", "crossdevice.npz"))
        q = torch.tensor(g[f"w{world}_r{rank}_q"], requires_grad=True)
        p = torch.tensor(g[f"w{world}_r{rank}_p"], requires_grad=True)
        # reference distributed_gather semantics: rank-major cat, gradient only to the own slice
        qa, pa = all_gather_with_local_grad(q), all_gather_with_local_grad(p)
        assert np.array_equal(qa.detach().numpy(), g[f"w{world}_r{rank}_q_reps"])
        assert np.array_equal(pa.detach().numpy(), g[f"w{world}_r{rank}_p_reps"])
        s = qa @ pa.T / 0.02
        G = pa.shape[0] // qa.shape[0]
        loss = torch.nn.functional.cross_entropy(s, torch.arange(qa.shape[0]) * G)
        loss.backward()
        ok = (abs(loss.item() - float(g[f"w{world}_r{rank}_loss"])) < 1e-9
              and np.allclose(q.grad.numpy(), g[f"w{world}_r{rank}_dq"], rtol=1e-8, atol=1e-12)
              and np.allclose(p.grad.numpy(), g[f"w{world}_r{rank}_dp"], rtol=1e-8, atol=1e-12))
        # the asynchronous, autograd-free gather used by ModelForTraining.forward
        eg = EmbeddingGather(p)
        ok = ok and np.array_equal(eg.wait().numpy(), g[f"w{world}_r{rank}_p_reps"])
        # bucketed gradient mean
        torch.manual_seed(0)
        lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
        red = FlatGradAllReducer(list(lin.parameters()), bucket_mb=1e-4)
        assert len(red.buckets) >= 2
        x = torch.full((4, 7), float(rank + 1))
        red.arm()
        lin(x).sum().backward()
        scale = red.finish()
        got = red.flat.clone() * scale
        # expectation: mean over ranks of the per-rank gradients
        exp = torch.zeros_like(red.flat)
        for r in range(world):
            l2 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
            l2.load_state_dict(lin.state_dict())
            l2(torch.full((4, 7), float(r + 1))).sum().backward()
            for prm, o in zip(reversed(list(l2.parameters())), red.offsets):
                exp[o:o + prm.numel()] += prm.grad.reshape(-1) / world
        ok = ok and torch.allclose(got, exp, rtol=1e-5, atol=1e-6)
        ok = ok and all(prm.grad.data_ptr() == red.flat[o:o + prm.numel()].data_ptr()
                        for prm, o in zip(red.order, red.offsets))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_gloo_gather_and_grad_reduce(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29711, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _torch_infonce(q_local, p_local, temperature, use_inbatch_neg=True, q_all=None, p_all=None, q_row0=0, p_row0=0):
    """Stand-in for ops.infonce_loss on CPU with the SAME contract: the loss over the gathered matrices, gradients to the
    local rows only (the HIP kernel writes exactly those rows)."""
    if q_all is None:
        qa, pa = q_local, p_local
    else:
        qa, pa = q_all.clone(), p_all.clone()
        qa = torch.cat([qa[:q_row0], q_local, qa[q_row0 + q_local.shape[0]:]], 0)
        pa = torch.cat([pa[:p_row0], p_local, pa[p_row0 + p_local.shape[0]:]], 0)
    s = qa @ pa.T / temperature
    G = pa.shape[0] // qa.shape[0]
    return torch.nn.functional.cross_entropy(s, torch.arange(qa.shape[0]) * G), s.detach()


def _forward_worker(rank, world, port, ret):
    """ModelForTraining.forward(negatives_cross_device=True) on real ranks (modeling.py:287-290, 331-404): row offsets, gather
    order, returned gathered reps, loss and own-row gradients against the reference's 2-rank gloo run (crossdevice.npz).
    The encoder and the scoring kernel are stubbed by torch ops (no GPU here): everything between them is the product."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rankpo_amd
        from rankpo_amd import encoder as PE, modeling, ops
        g = np.load(os.path.join(ROOT, "tests", "golden", "crossdevice.npz"))
        cfg = PE.llama_config(vocab_size=16, hidden_size=8, intermediate_size=8, num_hidden_layers=1,
                              num_attention_heads=2, num_key_value_heads=1, pad_token_id=0)
        ok = True
        # arms: "one_pass" = both towers in one packed pass -> ONE q||p all-gather; "two_towers" = configured (unpad=False) to run
        # the towers one after the other -> passage gather in flight during the query tower; "mixed" = one-pass configuration in
        # which ONLY RANK 1's batch falls off the packed path (a left-padded / holed row): the collective sequence must not
        # depend on the batch, so that rank still joins the ONE all-gather its peers issue (round 2 issued two there: a hang
        # or silently mis-sized gather on RCCL)
        for arm in ("one_pass", "two_towers", "mixed"):
            model = rankpo_amd.ModelForTraining(encoder=PE.LlamaEncoder(cfg), temperature=0.02,
                                                negatives_cross_device=True, normalize_embeddings=True,
                                                unpad=arm != "two_towers").train()
            assert (model.process_rank, model.world_size) == (rank, world)
            assert model._one_pass_configured() == (arm != "two_towers")
            q = torch.tensor(g[f"w{world}_r{rank}_q"], requires_grad=True)
            p = torch.tensor(g[f"w{world}_r{rank}_p"], requires_grad=True)
            model.embed = lambda x: x["reps"]
            calls = []
            real_ag, real_agt = dist.all_gather, dist.all_gather_into_tensor
            dist.all_gather = lambda *a, **k: (calls.append("all_gather"), real_ag(*a, **k))[1]
            dist.all_gather_into_tensor = lambda *a, **k: (calls.append("all_gather_into_tensor"), real_agt(*a, **k))[1]
            if arm == "one_pass" or (arm == "mixed" and rank != 1):
                model._embed_both = lambda qd, pd: torch.cat([qd["reps"], pd["reps"]], 0)
            else:
                model._embed_both = lambda qd, pd: None
            real = ops.infonce_loss
            ops.infonce_loss = _torch_infonce
            try:
                out = model(query={"reps": q, "input_ids": q}, passage={"reps": p, "input_ids": p})
            finally:
                ops.infonce_loss = real
                dist.all_gather, dist.all_gather_into_tensor = real_ag, real_agt
            # every rank issued the same number of collectives: one (one-pass configuration) or two (two towers)
            ncoll = torch.tensor([len(calls)])
            lo, hi = ncoll.clone(), ncoll.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            ok = ok and int(lo) == int(hi) == (2 if arm == "two_towers" else 1)
            out["loss"].backward()
            ok = ok and abs(out.loss.item() - float(g[f"w{world}_r{rank}_loss"])) < 1e-9
            ok = ok and np.allclose(out.scores.numpy(), g[f"w{world}_r{rank}_scores"], rtol=1e-9, atol=1e-9)
            ok = ok and np.array_equal(out.q_reps.detach().numpy(), g[f"w{world}_r{rank}_q_reps"])
            ok = ok and np.array_equal(out.p_reps.detach().numpy(), g[f"w{world}_r{rank}_p_reps"])
            ok = ok and np.allclose(q.grad.numpy(), g[f"w{world}_r{rank}_dq"], rtol=1e-8, atol=1e-12)
            ok = ok and np.allclose(p.grad.numpy(), g[f"w{world}_r{rank}_dp"], rtol=1e-8, atol=1e-12)
        # a parameter that gets no gradient must not leave its bucket un-reduced (FlatGradAllReducer.finish)
        from rankpo_amd.distributed import FlatGradAllReducer
        torch.manual_seed(0)
        used, unused = torch.nn.Linear(6, 4), torch.nn.Linear(4, 2)
        params = list(used.parameters()) + list(unused.parameters())
        red = FlatGradAllReducer(params, bucket_mb=1e-5)
        assert len(red.buckets) >= 3
        red.arm()
        used(torch.full((3, 6), float(rank + 1))).sum().backward()       # `unused` never fires its hooks
        scale = red.finish()
        assert red.late_buckets >= 1
        wg = used.weight.grad * scale
        exp = sum(torch.full((4, 6), 3.0 * (r + 1)) for r in range(world)) / world
        ok = ok and torch.allclose(wg, exp) and float(unused.weight.grad.abs().sum()) == 0.0
        # the same reduced values on every rank
        chk = red.flat.clone()
        dist.all_reduce(chk, op=dist.ReduceOp.MAX)
        ok = ok and torch.equal(chk, red.flat)
        # RankPOTrainer.resolve_metrics: ONE all-reduce of the 9-metric vector = the reference's nine
        # gather_for_metrics(x).mean() calls over equal per-rank batches (rankpo_trainer.py:496-520); sft_loss is dropped
        # when its weight is 0, key order as the reference builds its dict
        from rankpo_amd.rankpo_trainer import METRIC_KEYS
        tr = rankpo_amd.RankPOTrainer(PE.LlamaEncoder(cfg), None, reference_free=True, rankpo_weight=1.0, sft_weight=0.0)
        mvec = torch.arange(len(METRIC_KEYS), dtype=torch.float32) + 10.0 * rank
        got = tr.resolve_metrics(("eval_", mvec))
        want = {f"eval_{k}": i + 10.0 * (world - 1) / 2 for i, k in enumerate(METRIC_KEYS) if k != "sft_loss"}
        ok = ok and list(got) == list(want) and all(abs(got[k] - want[k]) < 1e-6 for k in want)
        ok = ok and torch.equal(mvec, torch.arange(len(METRIC_KEYS), dtype=torch.float32) + 10.0 * rank)   # caller's vector untouched
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_model_forward_cross_device(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_forward_worker, args=(world, 29713 + world, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` outside torchrun starts N fresh rank processes (scripts/train/run_contrastive.sh:27-30 does
    the same with torchrun) and relays ONE JSON line; under torch.distributed.run it runs as a rank.  CPU rehearsal of the
    launch + barrier + max-over-ranks only (no GPU here)."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--rehearse-launch"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert {k: line[k] for k in ("rehearsal", "n_gpus", "max_over_ranks")} == {"rehearsal": "launch", "n_gpus": 2, "max_over_ranks": 2.0}
    # the `comm` block of the N > 1 line (built by the same code, on gloo): every field present and populated, the backend saw
    # both ranks, one all-gather + one bucket all-reduce per step, the redundantly computed loss identical on both ranks, the
    # replicas' parameters still identical after the steps
    import importlib
    comm = line["comm"]
    assert tuple(comm) == importlib.import_module("bench").COMM_KEYS
    assert all(v is not None for v in comm.values()), comm
    assert comm["backend"] == "gloo" and comm["ranks_seen"] == comm["world_size"] == 2
    # per step: the q||p gather + one all-reduce per bucket + the two collectives of the batch re-deal (widths, token ids)
    assert comm["allgather_calls_per_step"] == 1.0 and comm["collectives_per_step"] == 3.0 + comm["allreduce_buckets"]
    rb = comm["rebalance"]
    assert rb["micro_steps"] == 3 and 1.0 <= rb["max_over_mean_as_run"] < rb["max_over_mean_as_sampled"]
    assert comm["late_buckets"] == 0 and comm["loss_equal_over_ranks"] is True and comm["params_in_sync"] is True
    assert comm["loss_min_over_ranks"] == comm["loss_max_over_ranks"] > 0
    assert 0 < comm["ms_per_step_min_over_ranks"] <= comm["ms_per_step_max_over_ranks"]
    assert comm["allgather_wait_us"] >= 0 and comm["allreduce_exposed_ms"] >= 0
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29717", bench, "--gpus", "2", "--rehearse-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert [json.loads(ln)["n_gpus"] for ln in r.stdout.splitlines() if ln.startswith("{")] == [2]
    # from under a profiler (its tool library is preloaded and may have initialised the GPU) the parent must not spawn ranks
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--rehearse-launch"], capture_output=True, text=True,
                       timeout=120, env=dict(env, ROCPROF_REHEARSAL_MARKER="1"))   # a key of the family rocprofv3 exports, harmless by itself
    assert r.returncode == 2 and "refusing to start rank processes from under a profiler" in r.stderr
    # a world-size mismatch is refused, not silently run
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--rehearse-launch"], capture_output=True, text=True,
                       timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_attention_flops_follow_the_kernel_arguments():
    """bench.py prices flash attention from the lengths the encoder hands to the kernels (filler sequence included), keyed by
    the packed token count the C call receives; an unknown count is an error, never 0 flops (round 1's driver line)."""
    import importlib
    bench = importlib.import_module("bench")
    from rankpo_amd import ops
    real = ops.attn_tile_table
    try:
        bench._ATTN_PAIRS.clear()
        bench.hook_attn_tables()
        lens = [1280, 700, 4096, 3000, 140]                   # 9216 = 36 * 256 after a 0-token filler? no: add one below
        lens = lens + [(-sum(lens)) % 256 or 256]
        ops.attn_tile_table(lens, "cpu")
        T = sum(lens)
        assert T % 256 == 0
        a = [None] * 35
        a[16:22] = 256, 0, T, 32, 8, 64
        nbytes, flops = bench._algo("rpo_flash_attn_bwd", a)
        assert flops == 10 * 64 * 32 * sum(n * (n + 1) // 2 for n in lens) > 0
        a[18] = T - 1
        with pytest.raises(KeyError, match="no sequence lengths registered"):
            bench._algo("rpo_flash_attn_bwd", a)
    finally:
        ops.attn_tile_table = real


def test_compute_metrics_matches_reference(golden):
    from rankpo_amd.retrieval import compute_metrics
    g = golden("metrics")
    meta = json.loads(str(g["meta"]))
    m = compute_metrics(g["preds"], g["scores"], meta["labels"], cutoffs=meta["cutoffs"])
    assert list(m) == list(meta["metrics"])            # same keys, same order
    for k, v in meta["metrics"].items():
        assert (np.isnan(v) and np.isnan(m[k])) or abs(m[k] - v) < 1e-12, k      # AUC with one class is nan in both


def test_fused_weight_is_a_view_of_the_flat_buffer():
    """encoder.fused_weight: q|k|v and gate|up tagged groups are laid out back to back by FlatGradAllReducer, their
    row-concatenation is then a zero-copy view, and the backward hands each weight its own rows."""
    from rankpo_amd import encoder as E
    from rankpo_amd.distributed import FlatGradAllReducer
    torch.manual_seed(0)
    cfg = E.llama_config(vocab_size=128, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                         num_attention_heads=4, num_key_value_heads=2, pad_token_id=0)
    model = E.LlamaEncoder(cfg)
    red = FlatGradAllReducer(list(model.parameters()), world_size=1)
    off = {id(p): o for p, o in zip(red.order, red.offsets)}
    att, mlp = model.layers[0].self_attn, model.layers[0].mlp
    q, k, v = att.q_proj.weight, att.k_proj.weight, att.v_proj.weight
    assert off[id(k)] == off[id(q)] + q.numel() and off[id(v)] == off[id(k)] + k.numel()
    assert off[id(mlp.up_proj.weight)] == off[id(mlp.gate_proj.weight)] + mlp.gate_proj.weight.numel()
    # every parameter still has exactly one slot
    assert len(set(off.values())) == len(red.order) == len(list(model.parameters()))
    # move the parameters into a flat buffer with the same layout (what FlatAdamW does on the device)
    flat = torch.zeros(red.numel)
    for p, o in zip(red.order, red.offsets):
        view = flat[o:o + p.numel()].view_as(p)
        view.copy_(p.data)
        p.data = view
    w = E.fused_weight([q, k, v])
    assert w.data_ptr() == q.data_ptr() and w.shape == (q.shape[0] + k.shape[0] + v.shape[0], q.shape[1])
    assert torch.equal(w, torch.cat([q, k, v], 0))
    g = torch.randn_like(w)
    w.backward(g)
    for t, sl in zip((q, k, v), g.split([q.shape[0], k.shape[0], v.shape[0]], 0)):
        assert torch.equal(t.grad, sl)
    # not adjacent (fresh tensors): falls back to a copy with the same values and gradients
    a, b = torch.randn(3, 8, requires_grad=True), torch.randn(5, 8, requires_grad=True)
    w2 = E.fused_weight([a, b])
    assert torch.equal(w2, torch.cat([a, b], 0))
    w2.sum().backward()
    assert torch.equal(a.grad, torch.ones_like(a)) and torch.equal(b.grad, torch.ones_like(b))


def test_generated_dkdv_slice_body_is_in_sync():
    """The hand-placed slice body of fa_bwd_dkdv4_kernel in attention.hip is generated text: it must be what
    tools/gen/gen_dkdv4_body.py emits today (edit the generator, not the macro)."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "gen", "gen_dkdv4_body.py")], capture_output=True,
                         text=True, check=True).stdout
    src = open(os.path.join(root, "rankpo_amd", "csrc", "attention.hip")).read()
    norm = lambda t: [re.sub(r"\s*\\$", "", ln.rstrip()) for ln in t.splitlines() if ln.strip()]
    a = src.index("// generated by tools/gen/gen_dkdv4_body.py")
    b = src.index("template <bool DOWN>\n__global__ __launch_bounds__(256, 1) void fa_bwd_dkdv4_kernel(")
    assert norm(src[a:b]) == norm(gen)


def test_key_block_table_properties():
    """ops.attn_key_tile_table (the dK/dV kernel's work list): every (sequence, kv head, 256-key block) exactly once, a
    (sequence, head) group on ONE XCD eighth, heaviest blocks first inside an eighth, eighths padded to equal length with
    entries that the kernel skips."""
    from rankpo_amd import ops
    rs = np.random.RandomState(0)
    for bn, nkv, lens in ((256, 8, rs.randint(1, 4097, size=37).tolist()), (256, 2, [5, 300, 257, 256, 1]), (256, 1, [1000]),
                          (128, 8, rs.randint(1, 4097, size=23).tolist()), (128, 2, [5, 300, 129, 128, 1])):   # 128: head_dim 128
        t = ops.attn_key_tile_table(lens, "cpu", nkv, block_n=bn, group_order=False).numpy()
        assert t.shape[0] % 8 == 0
        per = t.shape[0] // 8
        real = t[t[:, 2] < (1 << 30)]
        want = {(s, h, k) for s, n in enumerate(lens) for h in range(nkv) for k in range(0, n, bn)}
        assert len(real) == len(want) and set(map(tuple, real.tolist())) == want
        owner = {}
        for x in range(8):
            chunk = t[x * per:(x + 1) * per]
            work = [lens[s] - k for s, h, k in chunk.tolist() if k < (1 << 30)]
            assert work == sorted(work, reverse=True)                      # heaviest first, padding last
            assert (chunk[len(work):, 2] == (1 << 30)).all()
            for s, h, k in chunk[:len(work)].tolist():
                assert owner.setdefault((s, h), x) == x


def test_key_block_table_group_order():
    """The group-ordered dK/dV work list (pairs with sweep_down; the default at head_dim 128): every (sequence, kv head, key block)
    exactly once, a group on ONE XCD eighth and CONTIGUOUS there with its key blocks ascending (the heaviest first), eighths padded
    with entries the kernel skips.  With a tail fraction f (an A/B knob): the same up to the last f of an eighth's work, which runs
    heaviest-first."""
    from rankpo_amd import ops
    rs = np.random.RandomState(2)
    for bn, nkv, lens in ((128, 8, rs.randint(1, 4097, size=19).tolist()), (128, 2, [5, 300, 129, 128, 1]), (256, 4, [700, 256, 3])):
        f = 0.35
        d = ops.attn_key_tile_table(lens, "cpu", nkv, block_n=bn, group_order=f).numpy()
        g = ops.attn_key_tile_table(lens, "cpu", nkv, block_n=bn, group_order=True if bn == 256 else None).numpy()   # 128: the default
        assert d.shape == g.shape and sorted(map(tuple, d.tolist())) == sorted(map(tuple, g.tolist()))
        per_ = d.shape[0] // 8
        for x in range(8):
            dc, gc = d[x * per_:(x + 1) * per_], g[x * per_:(x + 1) * per_]
            nreal = int((gc[:, 2] < (1 << 30)).sum())
            assert sorted(map(tuple, dc[:nreal].tolist())) == sorted(map(tuple, gc[:nreal].tolist()))     # the same eighth
            work = np.array([lens[s_] - k_ for s_, _, k_ in gc[:nreal].tolist()], dtype=np.int64)
            cut = int(np.searchsorted(np.cumsum(work), (1.0 - f) * work.sum())) if nreal else 0
            assert (dc[:cut] == gc[:cut]).all()
            wt = np.array([lens[s_] - k_ for s_, _, k_ in dc[cut:nreal].tolist()], dtype=np.int64)
            assert (np.diff(wt) <= 0).all()
        t = g
        assert t.shape[0] % 8 == 0
        per = t.shape[0] // 8
        real = t[t[:, 2] < (1 << 30)]
        want = {(s, h, k) for s, n in enumerate(lens) for h in range(nkv) for k in range(0, n, bn)}
        assert len(real) == len(want) and set(map(tuple, real.tolist())) == want
        owner = {}
        for x in range(8):
            chunk = t[x * per:(x + 1) * per]
            nreal = int((chunk[:, 2] < (1 << 30)).sum())
            assert (chunk[nreal:, 2] == (1 << 30)).all()
            seen, last, lastk = set(), None, -1
            for s_, h_, k_ in chunk[:nreal].tolist():
                assert owner.setdefault((s_, h_), x) == x
                if (s_, h_) != last:
                    assert (s_, h_) not in seen
                    seen.add((s_, h_))
                    last, lastk = (s_, h_), -1
                assert k_ > lastk
                lastk = k_


def test_query_tile_table_properties():
    """ops.attn_tile_table in its XCD-dealt form (the forward / dQ kernels' work list): every (sequence, 128-query tile, head)
    exactly once, a (sequence, kv head) group inside ONE eighth and contiguous there (its blocks stream the same K / V through
    one L2), latest tiles of a group first, eighths padded to equal length with entries the kernels skip."""
    from rankpo_amd import ops
    rs = np.random.RandomState(1)
    for nh, nkv, lens in ((32, 8, rs.randint(1, 4097, size=29).tolist()), (8, 2, [5, 300, 257, 256, 1]), (4, 1, [1000]),
                          (4, 4, [129, 128])):
        t = ops.attn_tile_table(lens, "cpu", nh, nkv).numpy()
        assert t.shape[1] == 3 and t.shape[0] % 8 == 0
        per = t.shape[0] // 8
        real = t[t[:, 1] < (1 << 30)]
        want = {(s, q0, h) for s, n in enumerate(lens) for q0 in range(0, n, 128) for h in range(nh)}
        assert len(real) == len(want) and set(map(tuple, real.tolist())) == want
        rep = nh // nkv
        owner = {}
        for x in range(8):
            chunk = t[x * per:(x + 1) * per]
            nreal = int((chunk[:, 1] < (1 << 30)).sum())
            assert (chunk[nreal:, 1] == (1 << 30)).all()                      # padding last
            groups = [(s, h // rep) for s, q0, h in chunk[:nreal].tolist()]
            for gkey in groups:
                assert owner.setdefault(gkey, x) == x                           # one eighth per group
            # contiguous: a group never re-appears after another one started
            seen, last = set(), None
            for gkey in groups:
                if gkey != last:
                    assert gkey not in seen
                    seen.add(gkey)
                    last = gkey
            # inside a group: first query rows non-increasing
            for gkey in seen:
                q0s = [q0 for s, q0, h in chunk[:nreal].tolist() if (s, h // rep) == gkey]
                assert q0s == sorted(q0s, reverse=True)
    legacy = ops.attn_tile_table([300, 5], "cpu").numpy()
    assert legacy.shape == (4, 2) and legacy[0, 1] == 256
    # round 5: blocks of `block_m` queries x `heads_per_block` consecutive q heads of one kv head (the head_dim-128 forward's
    # <WQ, HEADS> instantiations): every (sequence, tile, head) is covered exactly once by the blocks' head ranges
    for block_m, hpb in ((64, 2), (128, 2), (64, 4), (256, 1)):
        lens = [300, 64, 65, 1000]
        t = ops.attn_tile_table(lens, "cpu", 32, 8, block_m=block_m, heads_per_block=hpb).numpy()
        real = t[t[:, 1] < (1 << 30)]
        assert (real[:, 2] % hpb == 0).all() and (real[:, 1] % block_m == 0).all()
        got = sorted((s, q0, h + j) for s, q0, h in real.tolist() for j in range(hpb))
        assert got == sorted((s, q0, h) for s, n in enumerate(lens) for q0 in range(0, n, block_m) for h in range(32))
        assert all((h // 4) == ((h + hpb - 1) // 4) for _, _, h in real.tolist())     # a block's heads share one kv head
    with pytest.raises(ValueError, match="heads_per_block"):
        ops.attn_tile_table([300], "cpu", 32, 8, heads_per_block=3)
    # the forward's own list exists exactly where the one-wave kernel applies: head_dim 128, a multiple of 4 q heads per kv head
    ft = ops.attn_fwd_tile_table([300, 64], "cpu", 32, 8, 128)
    assert ft is not None and torch.equal(ft, ops.attn_tile_table([300, 64], "cpu", 32, 8, block_m=64, heads_per_block=4))
    assert ops.attn_fwd_tile_table([300], "cpu", 16, 2, 128) is not None                  # 8 per kv head: two entries per tile
    assert ops.attn_fwd_tile_table([300], "cpu", 32, 8, 64) is None                       # head_dim 64: the kernel exists, and loses
    assert ops.attn_fwd_tile_table([300], "cpu", 32, 8, 64, force=True) is not None
    for nh, nkv, hd in ((32, 8, 96), (32, 16, 128), (12, 12, 128), (6, 2, 64)):
        assert ops.attn_fwd_tile_table([300], "cpu", nh, nkv, hd, force=True) is None


def test_rotary_frequencies_survive_a_dtype_cast():
    """`encoder.to(torch.bfloat16)` must not round the rotary frequencies (HF keeps inv_freq in float32 whatever the model dtype;
    modeling.py:175-178 loads the encoder through it): rounded to bf16 they turn position 4096 by radians, not ulps.  Round 3
    found the product doing exactly that (inv_freq was a module buffer) through the parity test at real sequence length."""
    from rankpo_amd import encoder as PE
    cfg = PE.llama_3_2_1b_config(vocab_size=64, num_hidden_layers=1, hidden_size=128, num_attention_heads=2, num_key_value_heads=1,
                                 intermediate_size=64)
    enc = PE.LlamaEncoder(cfg)
    want = PE._rope_inv_freq(cfg)
    for e in (enc, enc.to(torch.bfloat16), enc.to(torch.float32)):
        assert e.inv_freq.dtype == torch.float32 and torch.equal(e.inv_freq, want)
        t = e._rope(torch.tensor([4095]))
        assert torch.equal(t.cos32, torch.outer(torch.tensor([4095.0]), want).cos())
    assert "inv_freq" not in dict(enc.named_buffers()) and not any("inv_freq" in k for k in enc.state_dict())
    # what the rounding would have cost at the end of a 4096-token row: > 1 rad on the fastest frequencies
    err = (want.to(torch.bfloat16).float() - want).abs() * 4095
    assert err.max() > 1.0


def test_bert_dropout_follows_the_config_and_disable_dropout():
    """HF BertModel trains with hidden / attention-probability dropout 0.1 by default and the reference loads the checkpoint's
    config as it is (modeling.py:175-178); RankPOTrainer switches every dropout off when `disable_dropout` is set, its default
    (arguments.py:778-779, rankpo_trainer.py:209-213).  Here: eval mode ignores p; train mode with p > 0 is reproducible under a
    seed and differs from p = 0; a checkpoint's non-zero rates survive load_encoder; disable_dropout zeroes them; a Llama config
    that asks for attention dropout is refused."""
    import tempfile
    import rankpo_amd
    from rankpo_amd import encoder as PE
    torch.manual_seed(0)
    kw = dict(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=32)
    enc = PE.BertEncoder(PE.bert_config(**kw))
    assert PE.bert_config().hidden_dropout_prob == 0.1 and PE.bert_config().attention_probs_dropout_prob == 0.1      # HF defaults
    drops = [m for m in enc.modules() if isinstance(m, torch.nn.Dropout)]
    assert len(drops) == 1 + 3 * 2 and all(m.p == 0.1 for m in drops)
    ids = torch.randint(1, 64, (3, 10))
    mask = torch.ones(3, 10, dtype=torch.long)
    run = lambda: enc(input_ids=ids, attention_mask=mask).last_hidden_state
    enc.eval()
    e1, e2 = run(), run()
    assert torch.equal(e1, e2)
    enc.train()
    torch.manual_seed(7); t1 = run()
    torch.manual_seed(7); t2 = run()
    torch.manual_seed(8); t3 = run()
    assert torch.equal(t1, t2) and not torch.equal(t1, t3) and not torch.allclose(t1, e1, atol=1e-3)
    with tempfile.TemporaryDirectory() as d:
        PE.save_encoder(enc, d)
        back = PE.load_encoder(d)
        assert back.config.hidden_dropout_prob == 0.1 and back.embeddings.dropout.p == 0.1
    tr = rankpo_amd.RankPOTrainer(enc, None, reference_free=True)                     # disable_dropout defaults to True
    assert all(m.p == 0.0 for m in drops)
    assert torch.equal(run(), e1)                                                     # train mode, p = 0: the eval result
    del tr
    enc2 = PE.BertEncoder(PE.bert_config(**kw))
    rankpo_amd.RankPOTrainer(enc2, None, reference_free=True, disable_dropout=False)
    assert all(m.p == 0.1 for m in enc2.modules() if isinstance(m, torch.nn.Dropout))
    with pytest.raises(ValueError, match="attention_dropout"):
        PE.LlamaEncoder(PE.llama_config(vocab_size=16, hidden_size=8, intermediate_size=8, num_hidden_layers=1,
                                        num_attention_heads=2, num_key_value_heads=1, attention_dropout=0.1))


def test_compute_loss_defers_metrics_to_log():
    """compute_loss without return_outputs keeps the metric vector on the device; `log` resolves everything stored since the last
    log (mean over the steps) with one copy and reports what the reference's per-step `.item()` path reports
    (rankpo_trainer.py:570-587, 626-645)."""
    import rankpo_amd
    from rankpo_amd.rankpo_trainer import METRIC_KEYS
    tr = rankpo_amd.RankPOTrainer(None, None, reference_free=True, rankpo_weight=1.0, sft_weight=0.5)
    vecs = [torch.arange(len(METRIC_KEYS), dtype=torch.float32) + s for s in (0.0, 2.0, 4.0)]
    it = iter(vecs)
    tr.get_batch_loss_metrics = lambda model, batch, train_eval="train", sync_metrics=True: (
        torch.tensor(1.0), ("", next(it)) if not sync_metrics else None)
    for _ in vecs:
        assert float(tr.compute_loss(None, {})) == 1.0
    # the pending state is ONE running sum however many micro-steps pass between two logs (advisor, round 3: an unbounded list)
    assert tr._pending_metrics[2] == 3 and tr._pending_metrics[1].shape == vecs[0].shape and not tr._stored_metrics["train"]
    assert torch.equal(vecs[0], torch.arange(len(METRIC_KEYS), dtype=torch.float32))          # the caller's tensor is not the accumulator
    logs = tr.log({"loss": 1.0})
    assert tr._pending_metrics is None
    for i, k in enumerate(METRIC_KEYS):
        assert abs(logs[k] - (i + 2.0)) < 1e-6


def _cpu_optimizer_kernels(FlatAdamW):
    """torch stand-ins for the two HIP launches of FlatAdamW (rpo_sumsq_partial, rpo_adamw_step; include/rankpo_hip.h (4)) with
    the same contract, so that the partition / collective logic around them runs on gloo ranks without a GPU."""
    def sumsq(self, g):
        return g.float().pow(2).sum()

    def adamw(self, param, master, grad, m, v, lr, bc1, bc2, scale):
        b1, b2 = self.betas
        g = grad.float() * scale
        w = master if master is not None else param
        w.mul_(1.0 - lr * self.weight_decay)
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        w.sub_((lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + self.eps))
        if master is not None:
            param.copy_(master)
    FlatAdamW._sumsq, FlatAdamW._adamw = sumsq, adamw


def _partition_worker(rank, world, port, ret):
    """Optimizer-state partition (the reference's DeepSpeed ZeRO-1, configs/ds_zero1_config_llama.json:10-12, as reduce-scatter
    + per-rank AdamW shard + parameter all-gather) against the replicated all-reduce path: same parameters after 3 steps with
    gradient accumulation, replicas identical, state 1 / W the size."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rankpo_amd.train_step import FlatAdamW, TrainStep
        _cpu_optimizer_kernels(FlatAdamW)
        ok, why = True, []
        for dtype in (torch.bfloat16, torch.float32):
            runs = {}
            for part in (False, True):
                torch.manual_seed(0)
                net = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.Tanh(), torch.nn.Linear(40, 13),
                                          torch.nn.Tanh(), torch.nn.Linear(13, 5)).to(dtype)
                ts = TrainStep(net.parameters(), lambda b: net(b).float().pow(2).mean(), lr=1e-2, weight_decay=0.01, max_grad_norm=0.5,
                               gradient_accumulation_steps=2, total_steps=10, warmup_ratio=0.1, bucket_mb=1e-3,
                               partition_optimizer=part)
                opt = ts.opt
                assert opt.partition == part and len(opt.reducer.buckets) >= 2
                g = torch.Generator().manual_seed(100 + rank)
                for step in range(3):
                    ts.step([torch.randn(6, 24, generator=g).to(dtype) for _ in range(2)])
                runs[part] = (opt, opt.flat_param.clone())
            (rep, p_rep), (par, p_par) = runs[False], runs[True]
            # the parameters of the partitioned run live in a layout padded per bucket: compare parameter by parameter
            for (pa, oa), (pb, ob) in zip(zip(rep.reducer.order, rep.reducer.offsets), zip(par.reducer.order, par.reducer.offsets)):
                a, b = p_rep[oa:oa + pa.numel()].float(), p_par[ob:ob + pb.numel()].float()
                tol = 2e-2 if dtype == torch.bfloat16 else 1e-6             # bf16: a 1-ulp flip of a rounded parameter at most
                if not bool((a - b).abs().max() <= tol * a.abs().max().clamp_min(1e-3)):
                    why.append(("diff", str(dtype), float((a - b).abs().max()), float(a.abs().max())))
                # ... and only on a few elements: a ring all-reduce sums each chunk of the buffer in its own rank order, the
                # partitioned layout pads the buckets, so at W = 4 a bf16 gradient sum can round differently (W = 2: never)
                if dtype == torch.bfloat16 and not float((a != b).float().mean()) < (0.001 if world == 2 else 0.1):
                    why.append(("flips", float((a != b).float().mean()), a.numel()))
            # identical replicas after the all-gather
            chk = p_par.float().clone()
            dist.all_reduce(chk, op=dist.ReduceOp.MAX)
            ok = ok and torch.equal(chk, p_par.float())
            # state is 1 / W of the (padded) flat space; shards tile every bucket
            ok = ok and par.state_numel * world == par.reducer.numel and par.exp_avg.numel() == par.state_numel
            ok = ok and (par.master is None) == (dtype == torch.float32)
            ok = ok and all((e - s) % (8 * world) == 0 for s, e, _ in par.reducer.buckets)
            ok = ok and rep.state_numel == rep.reducer.numel
        ret[rank] = bool(ok) and not why
        if why:
            ret[f'why{rank}'] = why
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_optimizer_state_partition(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_partition_worker, args=(world, 29741 + world, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_bench_watchdog_names_the_phase_when_a_rank_stalls():
    """First contact with N GPUs must not end as a silent kill at the driver's limit: every rank logs its phases to stderr, the
    process group carries a timeout, and a per-rank watchdog ends the job NON-ZERO, naming the phase, when a phase outlives its
    budget.  Rehearsal on gloo: rank 1 stops answering in the second step; rank 0 then sits in the batch re-deal's all-reduce.
    The job must end well inside the collective timeout (60 s here), with the stuck phase in the output."""
    import subprocess
    import sys
    import time
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--rehearse-launch", "--rehearse-stall", "1", "--watchdog-scale", "0.25",
                        "--pg-timeout", "60"], capture_output=True, text=True, timeout=240, env=env)
    took = time.time() - t0
    assert r.returncode != 0, r.stderr[-2000:]
    assert "WATCHDOG: phase 'rehearsal steps'" in r.stderr and "exiting with code 3" in r.stderr, r.stderr[-3000:]
    assert "[bench r0 " in r.stderr and "[bench r1 " in r.stderr and "phase: process group init (gloo)" in r.stderr     # every rank logs
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]                # no result line from a job that hung
    assert took < 120, took
    # the watchdog itself: a phase inside its budget is left alone, `done()` disarms it
    import importlib
    B = importlib.import_module("bench")
    fired = []
    wd = B.Watchdog(0, scale=1.0, exit_fn=fired.append)
    wd.phase("quick", 5.0)
    wd.phase("slow", 30.0)
    time.sleep(1.2)
    assert not fired
    wd.phase("stuck", 0.3)
    time.sleep(1.5)
    assert fired == [3]
    wd2 = B.Watchdog(0, scale=1.0, exit_fn=fired.append)
    wd2.phase("ends", 0.3)
    wd2.done()
    time.sleep(1.2)
    assert fired == [3] and [n for n, _ in wd2.history] == ["start", "ends"]


def test_bench_lib_switch_really_switches(tmp_path):
    """`bench.py --lib other.so` is the in-step A/B of kernel builds.  Round 3's version set `_lib.LIB_PATH` after the package import
    had loaded and cached the in-tree build: every call still went to the in-tree library and the "A/B" compared it with itself
    (found in round 4 when a build that adds 2.3 ms to the backward measured no difference).  `select_library` must return, and make
    `_lib.load()` return, the OTHER file."""
    import importlib
    import shutil
    from rankpo_amd import _lib
    B = importlib.import_module("bench")
    in_tree, cached = _lib.LIB_PATH, _lib.load()
    other = str(tmp_path / "librankpo_hip_other.so")
    shutil.copy(in_tree, other)
    try:
        lib = B.select_library(other)
        assert os.path.samefile(lib._name, other) and _lib.load() is lib and lib is not cached
        assert lib.rpo_version() >= 100
        with pytest.raises(SystemExit):
            B.select_library(str(tmp_path / "missing.so"))
    finally:
        _lib.LIB_PATH, _lib._lib = in_tree, cached


def test_step_flops_count_matches_a_spelled_out_count():
    """`bench.llama_step_flops` (the numerator of `step_roofline`): against a count written out term by term for a small Llama
    shape -- GEMMs 2 FLOP per weight and token forward, x 3 with backward; causal attention 4 hd FLOP per (query, key <= query) pair
    and head forward, x 3.5 with backward; `required` runs the last block's q / o / MLP on the pooled rows only."""
    import importlib
    from types import SimpleNamespace
    B = importlib.import_module("bench")
    cfg = SimpleNamespace(hidden_size=64, num_attention_heads=4, num_key_value_heads=2, head_dim=16, intermediate_size=96,
                          num_hidden_layers=3)
    lens = [[5, 9, 2], [7]]
    d, nh, nkv, hd, ff, nl = 64, 4, 2, 16, 96, 3
    qo = 2 * d * nh * hd                      # q_proj + o_proj weights
    kv = 2 * d * nkv * hd
    mlp = 3 * d * ff
    T = sum(map(sum, lens))
    rows = sum(map(len, lens))
    pairs = sum(n * (n + 1) // 2 for b in lens for n in b)
    want_model_gemm = 3 * 2 * T * (qo + kv + mlp) * nl
    want_req_gemm = 3 * 2 * (T * (qo + kv + mlp) * (nl - 1) + T * kv + rows * (qo + mlp))
    attn_block = int(3.5 * 4 * hd * nh * sum(n * (n + 1) // 2 for n in lens[0])) + int(3.5 * 4 * hd * nh * sum(n * (n + 1) // 2 for n in lens[1]))
    got = B.llama_step_flops(cfg, lens)
    assert got["gemm_model"] == want_model_gemm and got["gemm_required"] == want_req_gemm
    assert got["attn_model"] == attn_block * nl
    assert got["attn_required"] == attn_block * (nl - 1) + sum(int(3.5 * 4 * hd * nh * sum(b)) for b in lens)
    assert pairs > 0 and got["gemm_required"] < got["gemm_model"] and got["attn_required"] < got["attn_model"]
    # the headline shape: ~948 TFLOP per cfg-2 step, from the mean lengths of the synthetic batches
    from rankpo_amd import encoder as PE
    c2 = PE.llama_3_2_1b_config()
    one = [[960] * 8 + [3072] * 48]
    f = B.llama_step_flops(c2, one)
    assert 0.8e15 < f["gemm_required"] + f["attn_required"] < 1.1e15


def test_generated_dkdv128_bodies_are_in_sync():
    """rankpo_amd/csrc/attention_dkdv128_gen.inc (the hand-placed slice bodies of fa_bwd_dkdv128_kernel and its literal-register
    statements) is generated text: it must be what tools/gen/gen_dkdv128_body.py emits today (edit the generator, not the file)."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    env = {k: v for k, v in os.environ.items() if not k.startswith("GEN_")}
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "gen", "gen_dkdv128_body.py")], capture_output=True,
                         text=True, check=True, env=env).stdout
    assert gen == open(os.path.join(root, "rankpo_amd", "csrc", "attention_dkdv128_gen.inc")).read()


def test_generated_fwd128w_statements_are_in_sync():
    """rankpo_amd/csrc/attention_fwd128w_gen.inc / attention_fwd64w_gen.inc (the hand-placed statements of fa_fwd128w_kernel /
    fa_fwd64w_kernel, the one-wave-per-SIMD forwards) are generated text: they must be what tools/gen/gen_fwd128w_body.py emits
    today (edit the generator, not the files)."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    env = {k: v for k, v in os.environ.items() if not k.startswith("GEN_")}
    for hd in (128, 64):
        gen = subprocess.run([sys.executable, os.path.join(root, "tools", "gen", "gen_fwd128w_body.py"), str(hd)], capture_output=True,
                             text=True, check=True, env=env).stdout
        assert gen == open(os.path.join(root, "rankpo_amd", "csrc", f"attention_fwd{hd}w_gen.inc")).read(), hd
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "gen", "gen_dq64w_body.py")], capture_output=True, text=True,
                         check=True, env=env).stdout                      # the head_dim-64 dQ kernel of the same make
    assert gen == open(os.path.join(root, "rankpo_amd", "csrc", "attention_dq64w_gen.inc")).read()


def test_no_valu_reads_a_transcendental_result_in_the_next_slot(tmp_path):
    """gfx950: a VALU instruction must not read the result of the v_exp_f32 / v_rcp_f32 / ... issued right in front of it (one wait
    state).  hipcc pads what it can see; the hand-placed asm streams and the one-instruction asm helpers it cannot.  Round 3's
    forward attention kernel packed un-exponentiated scores at head_dim 128 exactly this way (asm v_cvt_pk_bf16_f32 scheduled right
    behind its v_exp_f32) until the conversion became a compiler-visible vector conversion: this scans the ISA of every kernel of the
    library, built with the flags of the real build (no GPU needed: hipcc cross-compiles)."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    root = os.path.join(os.path.dirname(__file__), "..")
    # ONEWAVE64=1: the ISA of the optional head_dim-64 one-wave kernels is checked too (the default library leaves them out)
    subprocess.run(["make", "-s", "-C", os.path.join(root, "rankpo_amd", "csrc"), "isa", f"ISA_DIR={tmp_path}", "ONEWAVE64=1"],
                   check=True, capture_output=True, text=True)
    files = sorted(str(p) for p in tmp_path.glob("*.s"))
    assert len(files) == 8, files
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_trans_hazard.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    _check_dkdv_prefetch_registers(os.path.join(str(tmp_path), "attention.s"), root)
    _check_fwd128w_register_ownership(os.path.join(str(tmp_path), "attention.s"), root)
    _check_counted_waits(os.path.join(str(tmp_path), "attention.s"), os.path.join(str(tmp_path), "infonce.s"), root)


def _check_counted_waits(attention_s, infonce_s, root):
    """Advisor finding of round 5 (medium): the counted `s_waitcnt vmcnt(N)` waits of the one-wave-per-SIMD attention forwards and of
    the persistent 256 x 256 similarity kernel assume an exact number and order of vector-memory instructions in flight.
    tools/check_vmcnt_isa.py (run by `make` after every compile, a failure fails the build): the prologue rule on the code hipcc
    generated here, the recorded signatures (tools/isa_signatures.json) match that code, and both have teeth."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("check_vmcnt_isa", os.path.join(root, "tools", "check_vmcnt_isa.py"))
    C = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(C)
    assert C.main([attention_s, infonce_s]) == 0
    isa = open(attention_s).read()
    for kern, qp, waits in (("fa_fwd128w_kernel", 16, [12, 28]), ("fa_fwd64w_kernel", 8, [6, 14])):
        rep = C.prologue_report(isa, kern, qp)
        assert rep["ok"] and rep["counted_waits"] == waits, rep
        name, body = C.kernels(isa, kern)[0]
        first_dma = re.search(r"^\s*global_load_lds_dwordx4[^\n]*\n", body, flags=re.M)
        plain = re.search(r"^\s*global_load_dwordx4 v\[[^\n]*\n", body, flags=re.M)
        assert first_dma and plain and plain.start() > first_dma.start()
        # teeth (a): a rotary-table load hoisted in front of the Q pieces
        hoisted = body[:first_dma.start()] + plain.group(0) + body[first_dma.start():plain.start()] + body[plain.end():]
        bad = C.prologue_report(isa.replace(body, hoisted), kern, qp)
        assert not bad["ok"] and any("in front of or between" in p_ for p_ in bad["problems"]), bad
        # teeth (b): one K / V piece gone: no path carries exactly N instructions behind the Q pieces any more, and the signature moves
        dmas = list(re.finditer(r"^\s*global_load_lds_dwordx4[^\n]*\n", body, flags=re.M))
        gone = body[:dmas[qp].start()] + body[dmas[qp].end():]
        bad = C.prologue_report(isa.replace(body, gone), kern, qp)
        assert not bad["ok"] and any("no longer matches" in p_ for p_ in bad["problems"]), bad
        assert C.signature(isa.replace(body, gone), kern)["sha256"] != C.signature(isa, kern)["sha256"]


def _check_fwd128w_register_ownership(attention_s, root):
    """fa_fwd128w_kernel's asm statements own v[64:227] and the accumulator file BETWEEN statements (scores, fragments, the softmax
    scale, O^T / l / Q^T live there across hipcc's loop code): tools/check_fwd128w_isa.py reads the kernel's ISA -- nothing of hipcc's
    behind RPO_FW_INIT names them, no spill, no scratch -- and the check has teeth (a planted use is found)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_fwd128w_isa", os.path.join(root, "tools", "check_fwd128w_isa.py"))
    C = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(C)
    isa = open(attention_s).read()
    for kern in ("fa_fwd128w_kernel", "fa_fwd64w_kernel", "fa_bwd_dq64w_kernel"):
        rep = C.check(isa, kern)
        assert rep["ok"], (kern, rep["problems"][:10])
        assert rep["checked"] >= 1000 and rep["statements"] >= 100, rep  # the loop's own code and the statements were really seen
        body = "\n".join(C.kernel_body(isa, kern)[0])
        at = body.rindex("s_barrier")                                     # hipcc's code at the top of a key-tile iteration
        for planted in ("v_mov_b32_e32 v100, v1", "v_accvgpr_read_b32 v1, a17"):
            bad = C.check(isa.replace(body, body[:at] + "s_barrier\n\t" + planted + "\n" + body[at + len("s_barrier"):]), kern)
            assert not bad["ok"] and planted in " ".join(bad["problems"]), (kern, bad)


def _check_dkdv_prefetch_registers(attention_s, root):
    """Advisor finding of round 3 (medium): the hand-placed dK/dV kernels keep the next slice's row fragments in LITERAL VGPRs
    between two asm statements (v[128:175] at head_dim 64, v[96:175] at 128); hipcc only knows them as clobbers and may use them
    for its own values in the loop code between the statements.  tools/check_dkdv_isa.py walks the control-flow graph of the ISA
    (same .s files as the hazard scan above): no instruction on a path from a prefetching body to a HOT body -- the masked branch
    and the epilogue excluded -- may name them, for all four dK/dV kernels; and the check has teeth (a planted use is found)."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("check_dkdv_isa", os.path.join(root, "tools", "check_dkdv_isa.py"))
    C = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(C)
    isa = open(attention_s).read()
    for kern, prot in (("fa_bwd_dkdv4_kernelILb0", (128, 175)), ("fa_bwd_dkdv4_kernelILb1", (128, 175)),
                       ("fa_bwd_dkdv128_kernelILb0", (96, 175)), ("fa_bwd_dkdv128_kernelILb1", (96, 175))):
        rep = C.check(isa, kern, 192, prot)
        assert rep["ok"], rep
        assert rep["bodies"] == 4 and rep["hot_bodies"] == 2 and rep["protected"] == prot and rep["masked_blocks"] >= 1, rep
        assert rep["instructions_between"] >= 50, rep                  # the walk really covers the loop's own code
        # teeth: plant a write to a prefetched register behind the slice loop's barrier (the kernel's last s_barrier)
        name, body, _ = C.parse_kernel(isa, kern)
        at = body.rindex("s_barrier")
        planted = body[:at] + "s_barrier\n\tv_mov_b32_e32 v%d, v1\n" % (prot[0] + 2) + body[at + len("s_barrier"):]
        bad = C.check(isa.replace(body, planted), kern, 192, prot)
        assert not bad["ok"] and "name the prefetched registers" in " ".join(bad["problems"]), bad


def test_packed_path_verdict_spares_the_padded_paths_sync():
    """`pooled_last_token_multi` looks at every mask once (its one host sync) and leaves, per batch, whether the mask is right-padded
    in `last_right_padded`; `_mask(right_padded=...)` then decides without a second look at the tensor (advisor finding of round 2:
    a batch that falls back synchronised twice).  Left-padded, holed and empty-row masks return None from the packed path."""
    import torch
    from rankpo_amd import encoder as PE
    cfg = PE.llama_config(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                          num_key_value_heads=1, pad_token_id=0)
    enc = PE.LlamaEncoder(cfg)
    ids = torch.ones(3, 6, dtype=torch.int64)
    left = torch.tensor([[0, 0, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1]])
    right_with_empty_row = torch.tensor([[1, 1, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1]])
    assert enc.pooled_last_token_multi([(ids, left)]) is None and enc.last_right_padded == [False]
    assert enc.pooled_last_token_multi([(ids, right_with_empty_row), (ids, left)]) is None
    assert enc.last_right_padded == [True, False]         # an empty row fails the packed path, but the mask IS right-padded

    class NoLook:                                           # a mask whose values must not be read
        def __getattr__(self, name):
            raise AssertionError(f"_mask touched the mask ({name}) although the verdict was given")
    assert enc._mask(NoLook(), 6, torch.float32, right_padded=True) is None
    m = enc._mask(left, 6, torch.float32, right_padded=False)
    assert m.shape == (3, 1, 6, 6) and m.dtype == torch.bool
    assert torch.equal(m, enc._mask(left, 6, torch.float32))                         # the verdict changes no result
    assert enc._mask(right_with_empty_row, 6, torch.float32) is None


def test_deal_balanced_is_an_equal_sized_partition_that_evens_out_the_cost():
    from rankpo_amd.distributed import deal_balanced
    rs = np.random.RandomState(5)
    for hands, per in ((2, 8), (4, 8), (8, 8), (8, 1), (3, 5), (1, 7)):
        c = rs.uniform(10.0, 20.0, size=hands * per).tolist()
        h = deal_balanced(c, hands)
        assert sorted(sum(h, [])) == list(range(hands * per)) and all(len(x) == per for x in h)
        tot = [sum(c[i] for i in x) for x in h]
        naive = [sum(c[k * per:(k + 1) * per]) for k in range(hands)]
        assert max(tot) <= max(naive) + 1e-9
        if per >= 5 and hands > 1:
            assert max(tot) / (sum(tot) / hands) < 1.01 < max(naive) / (sum(naive) / hands) + 0.05     # within 1 % of perfect balance
        assert h == deal_balanced(c, hands)                                        # deterministic: every rank computes the same deal
    with pytest.raises(ValueError):
        deal_balanced([1.0, 2.0, 3.0], 2)


def _rebalance_worker(rank, world, port, ret):
    """`rebalance_groups` under gloo: ranks hold batches of different padded widths; afterwards the union of the ranks' (query +
    its passages) groups is the same set, every group is intact and in the collators' layout, the most expensive rank got cheaper,
    and every rank reports the same deal."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rankpo_amd.distributed import rebalance_groups, sequence_cost
        B, G = 4, 3
        rs = np.random.RandomState(100 + rank)
        Lq, Lp = 6 + rank, 12 + 2 * rank                       # the collators pad to the longest row of the LOCAL batch
        def side(n, L, tag0):
            lens = rs.randint(1, L + 1, size=n)
            lens[0] = L
            m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
            ids = rs.randint(100, 200, size=(n, L)) * m
            return ids, m, lens
        q_ids, q_m, q_len = side(B, Lq, 0)
        p_ids, p_m, p_len = side(B * G, Lp, 0)
        # tag every sequence with its global group id in its first token (all sequences have >= 1 token)
        for b in range(B):
            q_ids[b, 0] = 1000 + rank * B + b
            p_ids[b * G:(b + 1) * G, 0] = 1000 + rank * B + b
        q = {"input_ids": torch.tensor(q_ids), "attention_mask": torch.tensor(q_m)}
        p = {"input_ids": torch.tensor(p_ids), "attention_mask": torch.tensor(p_m)}
        q2, p2, info = rebalance_groups(q, p, pad_token_id=0)
        ok = q2["input_ids"].shape[0] == B and p2["input_ids"].shape[0] == B * G
        ok &= q2["input_ids"].shape == q2["attention_mask"].shape and p2["input_ids"].shape == p2["attention_mask"].shape
        tags_q = q2["input_ids"][:, 0].tolist()
        tags_p = p2["input_ids"][:, 0].view(B, G).tolist()
        ok &= all(tp == [tq] * G for tq, tp in zip(tags_q, tags_p))              # groups intact, passages behind their query
        ok &= bool((q2["input_ids"] * (1 - q2["attention_mask"])).eq(0).all())      # pad id under the mask's zeros
        # right-padded still, token counts of my groups = what the deal says
        ok &= bool((q2["attention_mask"][:, 1:] <= q2["attention_mask"][:, :-1]).all())
        my_cost = float(sequence_cost(torch.cat([q2["attention_mask"].sum(-1, keepdim=True),
                                                 p2["attention_mask"].sum(-1).view(B, G)], 1)).sum())
        ok &= abs(my_cost - info["cost_after"][rank]) < 1e-6
        ok &= max(info["cost_after"]) <= max(info["cost_before"]) + 1e-9
        # the same set of groups overall, the same deal on every rank
        gathered = [None] * world
        dist.all_gather_object(gathered, (tags_q, info["cost_after"], [int(q2["attention_mask"].sum()), int(p2["attention_mask"].sum())]))
        ok &= sorted(sum((g[0] for g in gathered), [])) == [1000 + i for i in range(world * B)]
        ok &= all(g[1] == gathered[0][1] for g in gathered)
        tot = [None] * world
        dist.all_gather_object(tot, int(q_m.sum() + p_m.sum()))
        ok &= sum(tot) == sum(g[2][0] + g[2][1] for g in gathered)                    # no token lost or invented
        # unequal per-rank batches: EVERY rank raises (nobody is left waiting inside the gather)
        nb = B - 1 if rank == 1 else B
        try:
            rebalance_groups({k: v[:nb] for k, v in q.items()}, {k: v[:nb * G] for k, v in p.items()})
            ok = False
        except ValueError as e:
            ok &= "different batch shapes" in str(e)
        # a rank-LOCAL defect (rank 1: one passage short of a whole group; then an empty batch, which divided by zero in round 3):
        # the verdict rides in the widths all-reduce, so every rank raises the same error instead of rank 1 alone
        for cut_q, cut_p in ((B, B * G - 1), (0, 0)):
            nq, np_ = (cut_q, cut_p) if rank == 1 else (B, B * G)
            try:
                rebalance_groups({k: v[:nq] for k, v in q.items()}, {k: v[:np_] for k, v in p.items()})
                ok = False
            except ValueError as e:
                ok &= "not a whole number of groups" in str(e) and (("this rank" in str(e)) == (rank == 1))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_rebalance_groups(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rebalance_worker, args=(world, 29761 + world, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


# ------------------------------------------------------------------------------------------------ bench.py's memory guard
_GIB = 2 ** 30
_TOK_CONTRASTIVE = 8 * 1280 + 48 * 4096          # cfg 2 / cfg 5: every row at full length (the pre-size step's batch)
_LLAMA_1B = (1_235_828_736 + 7 * 2048, 2, 2048, 8192, 16)      # parameters, bytes per element, d, ff, blocks
_LLAMA_8B = (7_504_953_344, 2, 4096, 14336, 32)
from rankpo_amd import memory as M      # noqa: E402  (the plan lives in the package since round 6; bench.py drives it)


def test_recomputation_context_is_seen_only_by_the_recomputation():
    """encoder._checkpoint_contexts hands torch.utils.checkpoint a recomputation context (ops.recomputing) under which a block's last
    computation may be skipped: it must be active during the recomputation inside backward -- on whichever thread autograd runs it --
    and at no other time (first forward, un-checkpointed calls, after backward)."""
    import torch
    from torch.utils.checkpoint import checkpoint
    from rankpo_amd import encoder as PE, ops
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            seen.append(ops.recomputing.active())
            ctx.save_for_backward(x)
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            (x_,) = ctx.saved_tensors                                # (unpacking is what triggers the recomputation)
            return g * 2.0 + 0.0 * x_
    x = torch.ones(3, requires_grad=True)
    y = checkpoint(lambda t: Probe.apply(t).sum(), x, use_reentrant=False, context_fn=PE._checkpoint_contexts)
    assert seen == [False] and not ops.recomputing.active()
    y.backward()
    assert seen == [False, True] and not ops.recomputing.active() and torch.equal(x.grad, torch.full((3,), 2.0))
    Probe.apply(x).sum().backward()                                  # no checkpoint: never a recomputation
    assert seen == [False, True, False]
    with ops.recomputing():
        with ops.recomputing():
            assert ops.recomputing.active()
        assert ops.recomputing.active()                              # nests
    assert not ops.recomputing.active()


def test_memory_guard_plans_cfg2_and_cfg5_without_a_gpu():
    """The checkpointing plan is a pure function of (usable bytes, model shape, world): bench.py's round-4 cfg-5 run died of a
    memory decision that only a GPU run could exercise (gpurun_out/r5h).  Pinned here against what the GPU measured:
    cfg 2 runs all 16 blocks un-checkpointed (measured worst-case peak 198 GiB, modelled 187); cfg 5 on one GPU checkpoints all
    32 (measured 253.4 GiB at 287.2 usable, modelled 253.7); with the optimizer state partitioned over 8 ranks 4 blocks run free."""
    import bench
    assert bench.plan_checkpointing is M.plan_checkpointing and bench.checkpoint_fewer is M.checkpoint_fewer      # ONE plan
    usable = int(287.2 * _GIB)
    assert M.plan_free_blocks(usable, *_LLAMA_1B, _TOK_CONTRASTIVE) == 16
    assert M.plan_checkpointing(usable, *_LLAMA_1B, _TOK_CONTRASTIVE) == (0, False)
    assert abs(M.modelled_peak_bytes(16, *_LLAMA_1B, _TOK_CONTRASTIVE) / _GIB - 198) < 15
    assert M.plan_free_blocks(usable, *_LLAMA_8B, _TOK_CONTRASTIVE) == 0
    assert M.plan_checkpointing(usable, *_LLAMA_8B, _TOK_CONTRASTIVE) == (32, False)
    # the calibration point: state + two kept input tensors per block + the block in flight (profiles/r05_cfg5_memory_summary.txt)
    assert abs(M.modelled_peak_bytes(0, *_LLAMA_8B, _TOK_CONTRASTIVE) / _GIB - 253.4) < 2.0
    assert abs(M.optimizer_state_bytes(_LLAMA_8B[0], 2, 1, False) / _GIB - 111.8) < 0.2
    # round 5: x + delta is formed in front of a checkpointed block, which then keeps ONE input tensor: 50.5 GiB less at cfg 5,
    # spent on two un-checkpointed blocks (each 17 GiB of activations instead of 1.6 GiB of input)
    assert abs(M.modelled_peak_bytes(0, *_LLAMA_8B, _TOK_CONTRASTIVE, block_inputs=1) / _GIB - (253.7 - 50.5)) < 0.5
    assert M.plan_free_blocks(usable, *_LLAMA_8B, _TOK_CONTRASTIVE, block_inputs=1) == 2
    assert M.plan_checkpointing(usable, *_LLAMA_8B, _TOK_CONTRASTIVE, block_inputs=1) == (30, False)
    assert M.plan_checkpointing(usable, *_LLAMA_1B, _TOK_CONTRASTIVE, block_inputs=1) == (0, False)
    # ... and what the measured worst-case step leaves under the plan's own budget goes back, once: cfg 5 measured 204.4 GiB with 30
    # blocks checkpointed (modelled 234); with 2 x 11 GiB set aside for the transposed d(gate|up) buffer one more block runs free
    shape8 = _LLAMA_8B + (_TOK_CONTRASTIVE,)
    assert M.checkpoint_fewer(int(204.4 * _GIB), usable, 30, *shape8, block_inputs=1, reserve=2 * int(11.0 * _GIB)) == 29
    assert M.checkpoint_fewer(int(204.4 * _GIB), usable, 30, *shape8, block_inputs=1) == 28
    assert M.checkpoint_fewer(int(243.0 * _GIB), usable, 30, *shape8, block_inputs=1) == 30          # no room: nothing changes
    assert M.checkpoint_fewer(int(60.0 * _GIB), usable, 2, *shape8, block_inputs=1) == 0             # never below zero
    assert M.checkpoint_fewer(int(60.0 * _GIB), usable, 0, *shape8, block_inputs=1) == 0
    # 8 ranks: `auto` partitions the optimizer state exactly because the replicated state forces checkpointing, and gives the
    # freed 70 GiB back to activations
    ckpt, part = M.plan_checkpointing(usable, *_LLAMA_8B, _TOK_CONTRASTIVE, world=8, multi=True, partition_mode="auto")
    assert part is True and ckpt == 28
    assert M.plan_checkpointing(usable, *_LLAMA_8B, _TOK_CONTRASTIVE, world=8, multi=True, partition_mode="off") == (32, False)
    assert M.plan_checkpointing(usable, *_LLAMA_1B, _TOK_CONTRASTIVE, world=8, multi=True, partition_mode="auto") == (0, False)
    # a card with less room (another tenant, RCCL buffers): fewer free blocks, never a negative count; monotone in the room
    last = -1
    for room in (40, 120, 180, 230, 287.2, 400):
        f = M.plan_free_blocks(int(room * _GIB), *_LLAMA_1B, _TOK_CONTRASTIVE)
        assert 0 <= f <= 16 and f >= last
        last = f
    assert M.plan_free_blocks(int(40 * _GIB), *_LLAMA_1B, _TOK_CONTRASTIVE) == 0
    # encoders without per-block control: all or nothing
    assert M.plan_checkpointing(int(100 * _GIB), *_LLAMA_1B, _TOK_CONTRASTIVE, per_block_control=False) == (-1, False)
    assert M.plan_checkpointing(usable, *_LLAMA_1B, _TOK_CONTRASTIVE, per_block_control=False) == (0, False)
    # a --share-gpu rehearsal splits the card between its ranks
    assert M.usable_hbm(200 * _GIB, 10 * _GIB, 288 * _GIB, 4) == (210 * _GIB) // 4
    assert M.usable_hbm(280 * _GIB, 20 * _GIB, 288 * _GIB) == 288 * _GIB


def test_memory_guard_retry_and_transposed_buffer_decisions():
    """The corrections that follow the MEASURED worst-case step: "tight" means out of memory or above 94 % of the usable HBM, the
    answer is a quarter more of the blocks until none is left; the 11 GiB transposed d(gate|up) buffer of the Llama-3-8B shape is
    refused on one GPU (253.4 GiB measured: the round-4 OOM) and admitted where twice its size still leaves 10 % free; an
    out-of-memory error may be answered by a retry only when the process is alone."""
    import bench
    usable = int(287.2 * _GIB)
    assert M.presize_is_tight(int(253.4 * _GIB), usable) is False          # cfg 5, N = 1: 88 %
    assert M.presize_is_tight(int(271 * _GIB), usable) is True
    assert M.presize_is_tight(0, usable, oom=True) is True
    assert [M.checkpoint_more(c, 32) for c in (0, 8, 22, 28, 31, 32)] == [8, 16, 30, 32, 32, None]
    assert M.checkpoint_more(0, 2) == 1 and M.checkpoint_more(16, 16) is None
    need8 = M.transposed_dgu_bytes(14336, _TOK_CONTRASTIVE, 2)
    need1 = M.transposed_dgu_bytes(8192, _TOK_CONTRASTIVE, 2)
    lim = 6 * _GIB
    assert abs(need8 / _GIB - 11.05) < 0.05 and abs(need1 / _GIB - 6.31) < 0.05
    assert M.admit_transposed_dgu(int(253.4 * _GIB), need8, usable, lim) is False      # cfg 5 on one GPU
    assert M.admit_transposed_dgu(int(198 * _GIB), need1, usable, lim) is True         # cfg 2's full-length batch
    assert M.admit_transposed_dgu(int(100 * _GIB), 5 * _GIB, usable, lim) is None      # within ops' static default: no decision
    # 8 ranks, state partitioned: what the plan's own peak model says for the free-block count it chose, and for one block fewer
    # (the measured peak decides at run time; on every rank the same way: the verdict is all-reduced)
    for free, expect in ((4, False), (3, True)):
        peak = M.modelled_peak_bytes(free, *_LLAMA_8B, _TOK_CONTRASTIVE, world=8, partitioned=True)
        assert M.admit_transposed_dgu(int(peak), need8, usable, lim) is expect
    assert M.may_retry_after_oom(1) is True and M.may_retry_after_oom(2) is False


def test_train_step_abort_leaves_a_clean_reducer(monkeypatch):
    """bench.py answers an out-of-memory error of the single-rank pre-size step by checkpointing more blocks and stepping again;
    the failed step must leave no armed reducer, no pending bucket and no stale gradient (advisor, round 4)."""
    from rankpo_amd.train_step import FlatAdamW, TrainStep
    monkeypatch.setattr(FlatAdamW, "_sumsq", FlatAdamW._sumsq)        # restored after the test: the stand-ins below are class-wide
    monkeypatch.setattr(FlatAdamW, "_adamw", FlatAdamW._adamw)
    _cpu_optimizer_kernels(FlatAdamW)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    boom = {"on": True}

    def loss_fn(b):
        y = net(b).pow(2).mean()
        if boom["on"]:
            y.backward()                                   # half a step's gradients are in the flat buffer ...
            raise torch.OutOfMemoryError("synthetic")       # ... when the step dies
        return y
    ts = TrainStep(net.parameters(), loss_fn, lr=1e-2, total_steps=4, warmup_ratio=0.0)
    x = torch.randn(4, 6)
    with pytest.raises(torch.OutOfMemoryError):
        ts.step([x])
    r = ts.opt.reducer
    assert r.flat.abs().sum() > 0
    ts.abort_step()
    assert r.flat.abs().sum() == 0 and not r._armed and not r._works and sum(r._pending) == 0
    lo = r.flat.data_ptr()
    assert all(lo <= p.grad.data_ptr() < lo + r.flat.numel() * r.flat.element_size() for p in net.parameters())
    boom["on"] = False
    before = [p.detach().clone() for p in net.parameters()]
    ts.step([x])
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    for p, b in zip(ref.parameters(), before):
        p.data.copy_(b)
    ts2 = TrainStep(ref.parameters(), lambda b: ref(b).pow(2).mean(), lr=1e-2, total_steps=4, warmup_ratio=0.0)
    ts2.step([x])
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.equal(a, b)                           # the retried step = a first step: nothing of the failed one leaked
    # an exception out of the OPTIMIZER update is not retryable (moments may be half-updated): abort_step refuses (advisor, round 5)
    real_step = ts.opt.step

    def failing_step(*a, **kw):
        raise torch.OutOfMemoryError("synthetic, inside opt.step")
    ts.opt.step = failing_step
    with pytest.raises(torch.OutOfMemoryError):
        ts.step([x])
    with pytest.raises(RuntimeError, match="inside the optimizer update"):
        ts.abort_step()
    ts.opt.step = real_step


def test_infonce_backward_dispatch_never_hands_the_hip_product_what_it_refuses():
    """Advisor finding of round 5 (medium): `rpo_sim_gemm_nt` returns RPO_ERR_UNSUPPORTED for operands of 4 GiB or more (32-bit
    piece offsets) and `check()` raises; the dispatch in `_bwd_product` now restates the entry point's conditions
    (`ops.sim_gemm_nt_takes`) so that such a dS goes to `ds @ x_all` as it did before the hand-written arm existed."""
    from rankpo_amd import ops
    ok = ops.sim_gemm_nt_takes
    assert ok(16384, 16384, 2048, 16384)                            # the sweep's top shape
    assert not ok(32768, 65536, 2048, 65536)                        # dS = 4 GiB
    assert ok(32768, 65536 - 64, 2048, 65536 - 64)                  # just below
    assert not ok(1024, 2 ** 21, 1024, 2 ** 21)                     # the transposed embeddings [d, K] = 4 GiB
    assert not ok(4096, 16400, 2048, 16400)                         # reduction not a multiple of the 64-element K-step
    assert not ok(4096, 16384, 2044, 16384)                         # output rows not 16-byte pieces
    assert not ok(4096, 16384, 2048, 16388)                         # a dS view whose row stride is not a multiple of 8
    assert not ok(4096, 16384, 2048, 16384, ds_ptr=8)               # unaligned base
    assert not ok(4096, 16384, 2048, 16384, bf16=False)             # f32 / fp16 storage: the library GEMM


def test_gradient_checkpointing_enable_resolves_to_the_memory_plan():
    """The reference's `--gradient_checkpointing` (scripts/train/run_contrastive.sh:39) checkpoints every block.  Here the bare call
    resolves to rankpo_amd.memory's plan (as few blocks as fit); "all" / None / an int keep their explicit meaning; on a CPU encoder
    there is no HBM to measure and every block is checkpointed, as HF does."""
    import torch
    from rankpo_amd import encoder as PE, memory as M
    cfg = PE.llama_config(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2,
                          num_key_value_heads=1, pad_token_id=0)
    enc = PE.LlamaEncoder(cfg).train()
    enc.gradient_checkpointing_enable()
    assert enc.checkpoint_layers == "auto" and enc._checkpointed_blocks(1000) == 3 and enc.memory_plan.tokens == 1024
    enc.gradient_checkpointing_enable(layers="all")
    assert enc.checkpoint_layers is None and enc._checkpointed_blocks(10) == 3
    enc.gradient_checkpointing_enable(layers=1)
    assert enc._checkpointed_blocks(10) == 1
    # the plan itself, from byte counts (what a GPU encoder measures): Llama-3.2-1B keeps every block at the BASELINE batch,
    # Llama-3-8B on one GPU checkpoints 30 of 32, on 8 GPUs with the optimizer state partitioned far fewer; a longer batch only
    # ever moves towards more checkpointing
    usable = int(287.5 * _GIB)
    p1 = M.plan_for_shape(usable, *_LLAMA_1B, _TOK_CONTRASTIVE)
    assert p1.checkpoint_blocks == 0 and p1.tokens == _TOK_CONTRASTIVE and p1.modelled_peak < 0.85 * usable
    p8 = M.plan_for_shape(usable, *_LLAMA_8B, _TOK_CONTRASTIVE)
    need8 = M.transposed_dgu_bytes(14336, _TOK_CONTRASTIVE, 2)
    assert p8.checkpoint_blocks == 30 and p8.transposed_dgu_limit == need8      # 11 GiB buffer: modelled peak 234 GiB + 2 x 11 < 90 %
    assert M.plan_for_shape(int(250 * _GIB), *_LLAMA_8B, _TOK_CONTRASTIVE).transposed_dgu_limit == 6 * 2 ** 30    # less HBM: refused
    p8w = M.plan_for_shape(usable, *_LLAMA_8B, _TOK_CONTRASTIVE, world=8, partitioned=True)
    assert p8w.checkpoint_blocks == 25 and p8w.transposed_dgu_limit == 6 * 2 ** 30    # the room goes to 5 more free blocks, not to the buffer
    #   (a recomputed block costs ~56 ms per step, the buffer saves ~1.2 ms per block: bench.py makes the same choice from measurements)
    assert M.plan_for_shape(usable, *_LLAMA_1B, 2 * _TOK_CONTRASTIVE).checkpoint_blocks > 0
    assert M.pad_tokens(1) == 256 and M.pad_tokens(256) == 256 and M.pad_tokens(257) == 512


def test_encode_bench_counts_the_same_forward_flops_as_the_training_bench():
    """bench_inference.forward_flops (the `encode` block's frac of the MFMA peak) is the forward third of bench.llama_step_flops's
    `required` count: GEMMs 2 of 6 FLOP per weight and token, causal attention 4 of 14 hd nh per (query, key) pair."""
    import bench
    import bench_inference as BI
    from rankpo_amd import encoder as PE
    lens = [4096, 2048, 3001, 17, 1]
    for cfg in (PE.llama_3_2_1b_config(), PE.llama_3_8b_config()):
        f = bench.llama_step_flops(cfg, [lens])
        assert BI.forward_flops(cfg, lens) == f["gemm_required"] // 3 + round(f["attn_required"] / 3.5)
    texts = BI.synthetic_texts(BI.load_bench_tokenizer(), 5, 20, 40, seed=1)
    assert len(texts) == 5 and len(texts[0].split()) == 40 and all(20 <= len(t.split()) <= 40 for t in texts)


def test_fused_search_chunk_schedule():
    """retrieval.FlatIPIndex.chunk_schedule (host arithmetic + one shape query of the C library): the fused search makes its first chunk
    -- the only one that still pays a score matrix and a selection pass -- as small as the 256 x 256 scoring kernel admits, then grows with the rows seen;
    chunks tile the corpus exactly; shapes the kernel does not take keep the plain schedule."""
    import torch
    from rankpo_amd.retrieval import FlatIPIndex

    def index(n, d, chunk_rows=262144, dtype=torch.bfloat16):
        ix = FlatIPIndex.__new__(FlatIPIndex)
        ix.emb, ix.ntotal, ix.chunk_rows, ix.split, ix.fused = torch.empty((n, d), dtype=dtype, device="meta"), n, chunk_rows, None, True
        return ix

    sizes = lambda sch: [b - a for a, b in sch]
    big = index(1_000_000, 2048)
    assert sizes(big.chunk_schedule(1024, 100)) == [12288, 31232, 111360, 262144, 262144, 262144, 58688]   # 2.56 x the rows seen, whole tiles
    assert sizes(big.chunk_schedule(256, 100)) == [49152, 125696, 262144, 262144, 300864]                  # (a tail below the first chunk's size joins)
    assert sizes(big.chunk_schedule(1024, 1024)) == [12288, 12288, 18432, 32256, 56320, 98560, 172544, 262144, 262144, 73024]
    assert sizes(big.chunk_schedule(1024, 100, fused=False)) == [262144, 262144, 262144, 213568]
    assert sizes(big.chunk_schedule(64, 100)) == [262144, 262144, 262144, 213568]               # <= 64 query rows: another kernel scores them
    assert sizes(index(1_000_000, 2048, dtype=torch.float32).chunk_schedule(1024, 100)) == [262144, 262144, 262144, 213568]
    assert sizes(index(1_000_000, 100).chunk_schedule(1024, 100)) == [262144, 262144, 262144, 213568]     # d % 64
    assert sizes(index(200_000, 192, 100_000).chunk_schedule(70, 1024)) == [49152, 49152, 101696]
    assert sizes(index(131_072, 256, 65_536).chunk_schedule(1024, 100)) == [12288, 31232, 65536, 22016]
    assert sizes(index(80_000, 64).chunk_schedule(1024, 100)) == [12288, 31232, 36480]
    assert sizes(index(149_999, 64, 50_000).chunk_schedule(65, 1)) == [50000, 50000, 49999]               # chunk_rows < 2 first chunks: plain
    assert sizes(index(30_000, 64).chunk_schedule(1024, 1024)) == [12288, 17712]
    assert sizes(index(5_000, 64).chunk_schedule(1024, 100)) == [5000]
    for nq, k in ((1024, 100), (300, 1000), (65, 7)):
        sch = big.chunk_schedule(nq, k)
        assert sch[0][0] == 0 and sch[-1][1] == 1_000_000 and all(a[1] == b[0] for a, b in zip(sch, sch[1:])) and sch[0][1] >= k
