#!/usr/bin/env python3
"""bench.py's `roofline_sweep` leg alone (similarity + InfoNCE forward through the C ABI, preallocated buffers, SURVEY 8d shapes)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
for r in bench.sweep(None, torch.device("cuda", 0)):
    print(json.dumps(r))
