import sys, time; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from rankpo_amd import ops
T, H, hd = 151552, 40, 64
x = torch.randn(T, 3072, device='cuda').to(torch.bfloat16)
pos = torch.arange(T, device='cuda') % 4096
inv = 1.0 / (500000.0 ** (torch.arange(0, hd, 2, device='cuda', dtype=torch.float32) / hd))
fr = torch.outer(pos.float(), inv); cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
for _ in range(3): ops.rope_(x, cos, sin, H, hd)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): ops.rope_(x, cos, sin, H, hd)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print(f"rope [{T}, {H} x {hd}] in place: {dt*1e6:.0f} us = {(2*T*H*hd*2 + 2*T*32*4)/dt/1e12:.2f} TB/s")
