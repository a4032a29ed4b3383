#!/usr/bin/env python3
"""Scoring entry points at the shapes the reference produces (warm-up 20, median of 200 calls, each call
bracketed by HIP events on the launch stream): rpo_infonce_fwd / rpo_infonce_bwd at 8 x 48 (W = 1) and 64 x 384 (W = 8),
d = 2048 / 4096 bf16 and the cfg-1 f32 shape; rpo_rankpo_fwd / _bwd at B = 8."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rankpo_amd import _lib

dev = "cuda:0"
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def timed(call, n=200, warm=20):
    for _ in range(warm):
        assert call() == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    # back to back: what a call costs inside a busy stream (its kernels + the gaps between them), launch latency hidden
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return ts[len(ts) // 2], e0.elapsed_time(e1) * 1e3 / n


for Q, P, d, dt in ((8, 48, 2048, 1), (64, 384, 2048, 1), (64, 384, 4096, 1), (8, 48, 384, 0), (16, 96, 2048, 1), (8, 16, 2048, 1), (8, 16, 256, 1), (8, 64, 2048, 1), (8, 48, 1024, 1)):
    tdt = torch.bfloat16 if dt else torch.float32
    torch.manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn(Q, d, device=dev), dim=-1).to(tdt)
    p = torch.nn.functional.normalize(torch.randn(P, d, device=dev), dim=-1).to(tdt)
    scores = torch.empty(Q, P, device=dev, dtype=tdt)
    lse = torch.empty(Q, device=dev); loss = torch.empty((), device=dev); gl = torch.ones((), device=dev)
    nws = lib.rpo_infonce_workspace_bytes(Q, P, d, dt)
    ws = torch.zeros(max(nws, 256), dtype=torch.uint8, device=dev)
    dq = torch.empty_like(q); dp = torch.empty_like(p)
    fwd = lambda: lib.rpo_infonce_fwd(q.data_ptr(), p.data_ptr(), Q, P, d, dt, 0.02, 0, scores.data_ptr(), lse.data_ptr(),
                                      loss.data_ptr(), ws.data_ptr(), nws, st)
    bwd = lambda: lib.rpo_infonce_bwd(q.data_ptr(), p.data_ptr(), scores.data_ptr(), lse.data_ptr(), gl.data_ptr(), Q, P, d, dt,
                                      0.02, 0, 0, Q, 0, P, dq.data_ptr(), dp.data_ptr(), None, 0, st)
    f, fm = timed(fwd)
    b, bm = timed(bwd)
    print(f"infonce {Q}x{P}x{d} {'bf16' if dt else 'f32'}: fwd {f:.1f} us isolated / {fm:.1f} us back-to-back, bwd {b:.1f} / {bm:.1f} us, "
          f"loss {loss.item():.5f}", flush=True)
