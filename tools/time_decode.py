"""Quick decode timing on the GPU box (HIP events on the launch stream)."""
import numpy as np
import torch
from multiposenet_amd.inference.utils import KeypointDecoder

for B in (1, 32, 256):
    dec = KeypointDecoder(B)
    lg = torch.randn(B, 128, 128, 17, device="cuda") * 1.5 - 4.6
    hm = torch.sigmoid(lg)
    box = torch.tensor([[512.0, 512.0]] * B, dtype=torch.float64, device="cuda")
    for _ in range(20):
        dec(hm, box, 0.2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        dec(hm, box, 0.2)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    byt = hm.numel() * 4
    print(f"decode B={B}: {us:.2f} us/call  {us / B:.3f} us/image  {byt / us / 1e3:.1f} GB/s")
