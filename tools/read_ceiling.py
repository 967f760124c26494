"""Read-only bandwidth ceilings on this box (what a reduction over one / two 134 MB tensors can reach): python tools/read_ceiling.py"""
import os
import sys
import torch
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
sys.path.insert(0, os.path.join(_root, 'tools'))
from time_misc_util import timeit
for n in (32 * 256 * 256 * 32, 32 * 128 * 128 * 64, 32 * 64 * 64 * 128):
    a = torch.randn(n, device='cuda').to(torch.bfloat16)
    b = torch.randn(n, device='cuda').to(torch.bfloat16)
    byt = n * 2
    t1 = timeit(lambda: torch.sum(a, dtype=torch.float32))
    t2 = timeit(lambda: torch.dot(a, b))
    af, bf = a.view(torch.int16), b.view(torch.int16)
    t3 = timeit(lambda: torch.max(af))
    print(f"{byt / 1e6:7.1f} MB: sum(a) {t1:7.1f} us {byt / t1 / 1e3:6.0f} GB/s | dot(a, b) {t2:7.1f} us {2 * byt / t2 / 1e3:6.0f} GB/s | max(a) {t3:7.1f} us {byt / t3 / 1e3:6.0f} GB/s")
