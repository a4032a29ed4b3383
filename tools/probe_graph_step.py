#!/usr/bin/env python3
"""Feasibility probe: the cfg-1 training step (BGE-small architecture, f32, fixed padded shapes) captured into a HIP graph with
torch.cuda.graph vs run eagerly.  The probe bakes the learning rate of the captured step into the graph (timing only)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import rankpo_amd
from rankpo_amd.encoder import build_encoder
from rankpo_amd.train_step import TrainStep

dev = torch.device("cuda", 0)
arch, B, K, Lq, Lp, temperature, dtn = bench.WORKLOADS[os.environ.get("WL", "cfg1")]
cfg = bench.build_config(arch)
torch.manual_seed(0)
with torch.device(dev):
    enc = build_encoder(cfg)
enc = enc.to(torch.float32 if dtn == "f32" else torch.bfloat16)
model = rankpo_amd.ModelForTraining(encoder=enc, temperature=temperature, use_inbatch_neg=True, unpad=False).train()
ts = TrainStep(model.parameters(), lambda b: model(**b)["loss"], lr=1e-5, total_steps=1000, warmup_ratio=0.0)
batches = [bench.synth_batch(cfg, B, K, Lq, Lp, 1234 + i, dev) for i in range(8)]
static = {k: {kk: vv.clone() for kk, vv in v.items()} for k, v in batches[0].items()}

def load(b):
    for k in static:
        for kk in static[k]:
            static[k][kk].copy_(b[k][kk])

def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for i in range(3):
    ts.step(batches[i])
print(f"eager: {timeit(lambda i: ts.step(batches[i % 8])):.2f} ms/step")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(3):
        load(batches[i]); ts.step(static)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = ts.step(static)
torch.cuda.synchronize()
def replay(i):
    load(batches[i % 8]); g.replay()
print(f"graph: {timeit(replay):.2f} ms/step, loss {loss.item():.4f}")
