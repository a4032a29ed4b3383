"""GPU rehearsal of the N > 1 code path with the RCCL backend at world_size 1 (one GPU box): process-group init as
bench.py does it, asynchronous embedding all-gather, own-row gradient window, gradient-bucket hooks and the flat AdamW
step.  With one rank the results must equal the non-distributed run exactly."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def nccl_world1():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29633")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield
    dist.destroy_process_group()


def _batch(rs, N, L, vocab):
    lens = rs.randint(L // 2, L + 1, size=N)
    lens[0] = L
    m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    return {"input_ids": torch.tensor(rs.randint(1, vocab, size=(N, L)) * m).to(DEV), "attention_mask": torch.tensor(m).to(DEV)}


def test_cross_device_path_world1_equals_local(nccl_world1):
    import rankpo_amd
    from rankpo_amd import encoder as PE
    from rankpo_amd.train_step import TrainStep
    cfg = PE.llama_config(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                          num_attention_heads=4, num_key_value_heads=2, pad_token_id=0)
    rs = np.random.RandomState(0)
    batches = [{"query": _batch(rs, 4, 24, 256), "passage": _batch(rs, 12, 48, 256)} for _ in range(3)]
    results = []
    # local; cross-device path with replicated optimizer state (bucketed all-reduce); the same with the optimizer state
    # partitioned (RCCL reduce-scatter per bucket, AdamW per shard, in-place parameter all-gather): at one rank all three agree
    for xdev, part in ((False, False), (True, False), (True, True)):
        torch.manual_seed(0)
        enc = PE.LlamaEncoder(cfg).to(DEV).to(torch.bfloat16)
        model = rankpo_amd.ModelForTraining(encoder=enc, temperature=0.02, negatives_cross_device=xdev).train()
        ts = TrainStep(model.parameters(), lambda b: model(**b)["loss"], lr=1e-3, total_steps=10, warmup_ratio=0.0,
                       bucket_mb=0.05, force_collectives=xdev, partition_optimizer=part)
        assert len(ts.opt.reducer.buckets) > 1
        assert ts.opt.reducer._reduce == xdev            # the bucketed async all-reduce hooks are live in the xdev run
        assert ts.opt.partition == part and ts.opt.reducer.shard == part
        losses = [ts.step(b).item() for b in batches]
        out = model(**batches[0])
        assert out.q_reps.shape[0] == 4 and out.p_reps.shape[0] == 12
        results.append((losses, enc.layers[1].mlp.down_proj.weight.detach().float().clone()))
    assert results[0][0] == results[1][0] == results[2][0]
    assert torch.equal(results[0][1], results[1][1]) and torch.equal(results[0][1], results[2][1])
    assert results[0][0][0] != results[0][0][2]      # parameters really moved


def test_distributed_gather_api_and_metrics_allreduce(nccl_world1):
    import rankpo_amd
    from rankpo_amd import encoder as PE
    cfg = PE.llama_config(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=1,
                          num_attention_heads=2, num_key_value_heads=1, pad_token_id=0)
    model = rankpo_amd.ModelForTraining(config=cfg, temperature=0.02, negatives_cross_device=True).to(DEV)
    assert model.process_rank == 0 and model.world_size == 1
    x = torch.randn(5, 64, device=DEV, requires_grad=True)
    for method in (1, 2, 3):
        y = model.distributed_gather(x, use_method=method)
        assert torch.equal(y, x)
        (y * 2).sum().backward()
    assert torch.allclose(x.grad, torch.full_like(x, 6.0))
    tr = rankpo_amd.RankPOTrainer(model.model, None, beta=2.0, temperature=0.1, reference_free=True)
    rs = np.random.RandomState(1)
    loss, metrics = tr.compute_loss(model.model, {"query": _batch(rs, 3, 10, 64), "passage": _batch(rs, 6, 12, 64)},
                                    return_outputs=True)
    assert np.isfinite(loss.item()) and "rewards/accuracies" in metrics and "sft_loss" not in metrics


def test_two_ranks_end_to_end_on_one_gpu():
    """The WHOLE N > 1 bench path with two real rank processes on the HIP kernels: `bench.py --gpus 2 --share-gpu` (both ranks on this
    box's one GPU, process group on gloo because RCCL refuses two ranks on one device; reference: torchrun --nproc-per-node N,
    scripts/train/run_contrastive.sh:27-30, with negatives_cross_device, modeling.py:287-290).  Checks what the `comm` block is
    for: the backend saw both ranks, one q||p all-gather per step, the redundantly computed global loss is the SAME on both ranks,
    the replicas' parameters are bit-identical after the steps -- with the replicated and with the partitioned optimizer state."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    for extra in ([], ["--partition-optimizer", "on", "--gas", "2"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--workload", "tiny",
                            "--steps", "3", "--warmup", "1", "--no-sweep", "--no-cpu-baseline"] + extra,
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        out = json.loads(lines[0])
        comm = out["comm"]
        assert out["n_gpus"] == 2 and comm["ranks_seen"] == 2 and comm["world_size"] == 2
        assert comm["allgather_calls_per_step"] == 1.0 * (2 if extra else 1)          # one per micro-step
        assert comm["loss_equal_over_ranks"] is True and comm["params_in_sync"] is True and comm["late_buckets"] == 0
        assert comm["optimizer_state_partitioned"] == bool(extra)
        # the global batch's groups were re-dealt to the ranks by packed-token cost before every micro-step (the loss equality
        # above holds WITH the re-deal: the global batch is unchanged)
        # `--balance auto` with two ranks: the first half of the timed steps (2 of 3) runs with the re-deal, the rest as sampled,
        # and the line carries both halves' step times (bracketed by barriers inside the one run)
        rb = comm["rebalance"]
        assert rb["micro_steps"] == 2 * (2 if extra else 1) and 1.0 <= rb["max_over_mean_as_run"] <= rb["max_over_mean_as_sampled"]
        on, off = rb["segments"]
        assert (on["redeal"], on["steps"], off["redeal"], off["steps"]) == (True, 2, False, 1)
        assert on["ms_per_step"] > 0 and off["ms_per_step"] > 0 and on["max_over_mean_cost_as_sampled"] >= 1.0
        assert out["config"]["memory_guard"]["retries"] == 0 and out["step_roofline"]["frac"] > 0
        assert np.isfinite(out["loss_last"]) and out["loss_first"] != out["loss_last"]


def test_rebalanced_global_batch_gives_the_same_loss():
    """`bench.py --balance on` re-deals the (query + passages) groups of the global batch to the ranks by packed-token cost
    (distributed.rebalance_groups); the global batch is the same set of groups, so the cross-device InfoNCE loss of the first step
    -- before any parameter moved -- must equal the un-balanced run's up to the summation order of a bf16 forward
    (reference: modeling.py:287-314, the loss is a mean over the gathered batch)."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    first = {}
    for mode in ("on", "off"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--workload", "tiny",
                            "--steps", "1", "--warmup", "0", "--no-sweep", "--no-cpu-baseline", "--balance", mode],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert (out["comm"]["rebalance"] is not None) == (mode == "on")
        assert out["comm"]["loss_equal_over_ranks"] is True
        first[mode] = out["loss_first"]
    assert abs(first["on"] - first["off"]) <= 2e-3 * abs(first["off"]), first
