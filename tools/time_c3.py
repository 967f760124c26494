"""Time the 3x3 layers of the bench shape (forward with affine + statistics, data gradient), HIP events: python tools/time_c3.py"""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


dt = torch.bfloat16
N = 32
for (H, Cin, Cout) in [(128, 128, 128), (64, 128, 128), (32, 128, 128), (16, 128, 128), (128, 512, 64), (128, 64, 512), (112, 256, 256), (112, 64, 64), (56, 64, 64)]:
    x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
    pc = ops.PackedConv(torch.randn(3, 3, Cin, Cout, device='cuda') * 0.05, dt)
    aff = ops.Affine(torch.rand(Cin, device='cuda') + 0.5, torch.randn(Cin, device='cuda') * 0.1, 1)
    y = torch.empty(N, H, H, Cout, device='cuda', dtype=dt)
    dy = torch.randn(N, H, H, Cout, device='cuda').to(dt)
    dx = torch.empty_like(x)
    part = torch.empty(ops.conv_num_parts(N, H, H, 3) * 2 * Cout, device='cuda')
    fl = 2.0 * N * H * H * Cin * Cout * 9
    uf = t(lambda: ops.conv_fwd(x, pc.fwd, Cout, 3, aff, out=y, stats_part=part))
    ub = t(lambda: ops.conv_fwd(dy, pc.bwd, Cin, 3, None, out=dx))
    print(f"3x3 {Cin:4d}->{Cout:4d} @{H:3d}: fwd {uf:7.1f} us {fl / uf / 1e6:7.1f} TF ({fl / uf / 2.5e9:.3f}) | dgrad {ub:7.1f} us {fl / ub / 1e6:7.1f} TF ({fl / ub / 2.5e9:.3f})", flush=True)
