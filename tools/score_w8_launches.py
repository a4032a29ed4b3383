"""Scoring + InfoNCE at the W = 8 shape (64 gathered queries x 384 gathered passages, d = 2048, bf16; modeling.py:287-314):
how many launches and how long.  Run under `rocprofv3 --kernel-trace --stats -- python3 tools/score_w8_launches.py`; prints the
HIP-event time per forward call (100 calls back to back)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rankpo_amd import ops  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
for Q, P, d in ((64, 384, 2048), (64, 384, 4096)):
    q = torch.nn.functional.normalize(torch.randn(Q, d, device=dev), dim=-1).bfloat16()
    p = torch.nn.functional.normalize(torch.randn(P, d, device=dev), dim=-1).bfloat16()
    for _ in range(10):
        ops.infonce_loss(q, p, 0.02)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        loss, _ = ops.infonce_loss(q, p, 0.02)
    e1.record()
    torch.cuda.synchronize()
    print(f"Q={Q} P={P} d={d}: {e0.elapsed_time(e1) * 10:.2f} us per forward call, loss {loss.item():.6f}")
