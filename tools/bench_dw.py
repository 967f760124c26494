"""Depthwise 3x3 layers of the bench shape (bs32 @ 512x512, bf16): forward with the producer's affine + ReLU6 + statistics,
forward without affine, and a plain device copy of the same tensors (the read + write ceiling): python tools/bench_dw.py"""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dt = torch.bfloat16
st = torch.cuda.current_stream()


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


B = 32
for (H, C, s) in [(256, 32, 1), (256, 64, 2), (128, 128, 1), (128, 128, 2), (64, 256, 1), (64, 256, 2), (32, 512, 1), (32, 512, 2), (16, 1024, 1)]:
    x = torch.randn(B, H, H, C, device="cuda").to(dt)
    w = torch.randn(3, 3, C, device="cuda") * 0.2
    aff = ops.Affine(torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, 2)
    OH = H // s
    y = torch.empty(B, OH, OH, C, device="cuda", dtype=dt)
    part = torch.empty(ops.dwconv_num_parts(B, H, H, C, s, dt) * 2 * C, device="cuda")
    byt = (x.numel() + y.numel()) * 2
    t1 = timed(lambda: ops.dwconv_fwd(x, w, s, aff, out=y, stats_part=part))
    t2 = timed(lambda: ops.dwconv_fwd(x, w, s, None, out=y))
    xs = x[:, ::s, ::s, :] if s == 2 else x
    ycp = torch.empty_like(x)
    t3 = timed(lambda: ycp.copy_(x))
    print(f"{C:5d}ch @{H:3d}x{H:<3d} s{s}: fwd {t1 * 1e6:6.1f} us {byt / t1 / 1e9:5.0f} GB/s ({byt / t1 / 8e12:.3f}) | no affine/stats {t2 * 1e6:6.1f} us "
          f"({byt / t2 / 8e12:.3f}) | torch copy of x {t3 * 1e6:6.1f} us ({2 * x.numel() * 2 / t3 / 8e12:.3f} of 8 TB/s)", flush=True)
