#!/usr/bin/env python3
"""Runs only the similarity+InfoNCE forward at sweep shapes (for rocprofv3 counter passes / timing)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rankpo_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
shapes = [(int(a), int(b), int(c)) for a, b, c in (s.split("x") for s in os.environ.get("SHAPES", "16384x16384x2048").split(","))]
for Q, P, d in shapes:
    q = torch.nn.functional.normalize(torch.randn(Q, d, device=dev), dim=-1).to(torch.bfloat16)
    p = torch.nn.functional.normalize(torch.randn(P, d, device=dev), dim=-1).to(torch.bfloat16)
    variants = [""]                # the library reads no environment switches any more: one variant, the shipped one
    n = int(os.environ.get("REPS", "5"))
    for rnd in range(int(os.environ.get("ROUNDS", "1"))):
        for var in variants:
            for _ in range(2):
                ops.infonce_loss(q, p, 0.02)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.infonce_loss(q, p, 0.02)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print(f"{Q}x{P}x{d} [{var}]: {ms*1e3:.1f} us  {2.0*Q*P*d/ms/1e9:.0f} TFLOP/s", flush=True)
