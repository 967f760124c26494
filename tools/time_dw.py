"""Time the depthwise kernels at the bench shapes: python tools/time_dw.py"""
import sys
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
from multiposenet_amd import ops
from time_misc_util import timeit

dt = torch.bfloat16
N = 32
for (H, C, s) in [(256, 32, 1), (256, 64, 2), (128, 128, 1), (128, 128, 2), (64, 256, 1), (64, 256, 2), (32, 512, 1), (32, 512, 2), (16, 1024, 1)]:
    x = torch.randn(N, H, H, C, device='cuda').to(dt)
    w = torch.randn(3, 3, C, device='cuda') * 0.2
    sc = torch.rand(C, device='cuda') + 0.5
    sh = torch.randn(C, device='cuda') * 0.1
    aff = ops.Affine(sc, sh, 2)
    OH = H // s
    y = torch.empty(N, OH, OH, C, device='cuda', dtype=dt)
    part = torch.empty(ops.dwconv_num_parts(N, H, H, C, s, dt) * 2 * C, device='cuda')
    us = timeit(lambda: ops.dwconv_fwd(x, w, s, aff, out=y, stats_part=part))
    byt = (x.numel() + y.numel()) * 2
    dy = torch.randn(N, OH, OH, C, device='cuda').to(dt)
    dw = torch.empty(3, 3, C, device='cuda')
    us2 = timeit(lambda: ops.dwconv_bwd_weight(x, dy, s, aff, dw))
    dx = torch.empty_like(x)
    us3 = timeit(lambda: ops.dwconv_bwd_data(dy, w, (H, H), s, out=dx))
    print(f"dw {C:4d}ch @{H} s{s}: dgrad {us3:6.1f} us {byt / us3 / 1e3:6.0f} GB/s | fwd {us:6.1f} us {byt / us / 1e3:6.0f} GB/s | wgrad(+reduce) {us2:6.1f} us {byt / us2 / 1e3:6.0f} GB/s")
