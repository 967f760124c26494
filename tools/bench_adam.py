"""mpn_adam_step on an arena of the PRN's size (about 52 M floats) and of the keypoint network's (5.5 M), cold caches."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

for n in (52_000_000, 5_500_000):
    nbuf = 3 if n > 10_000_000 else 12
    bufs = [[torch.randn(n, device="cuda") * s for s in (1.0, 0.01, 0.01, 1e-4)] for _ in range(nbuf)]
    for b in bufs:
        b[3].abs_()
    hyper = torch.tensor([1e-3, 1e-3, 0, 0], device="cuda")
    def run(i):
        p, g, m, v = bufs[i % nbuf]
        ops.adam_step(p, g, m, v, hyper)
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 12
    a.record()
    for i in range(reps):
        run(i)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    print(f"n={n}: {us:7.1f} us  {7 * 4 * n / us / 1e6:5.2f} TB/s", flush=True)
    del bufs
