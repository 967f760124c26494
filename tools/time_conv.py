"""Time the dominant conv shapes (HIP events). Usage: python tools/time_conv.py"""
import sys
import torch
sys.path.insert(0, '.')
from multiposenet_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

dt = torch.bfloat16
N = 32
for (H, Cin, Cout, k) in [(128, 128, 128, 3), (128, 512, 64, 3), (128, 64, 512, 3), (64, 128, 128, 3), (256, 32, 64, 1), (128, 128, 128, 1), (64, 256, 256, 1), (32, 256, 512, 1), (32, 512, 512, 1), (16, 512, 1024, 1), (16, 1024, 1024, 1), (16, 1024, 128, 1)]:
    x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
    w = torch.randn(k, k, Cin, Cout, device='cuda') * 0.05
    pc = ops.PackedConv(w, dt)
    sc = torch.rand(Cin, device='cuda') + 0.5; sh = torch.randn(Cin, device='cuda') * 0.1
    y = torch.empty(N, H, H, Cout, device='cuda', dtype=dt)
    part = torch.empty(ops.conv_num_parts(N, H, H, k) * 2 * Cout, device='cuda')
    us = t(lambda: ops.conv_fwd(x, pc.fwd, Cout, k, ops.Affine(sc, sh, 1), out=y, stats_part=part))
    fl = 2.0 * N * H * H * Cin * Cout * k * k
    byt = (x.numel() + y.numel()) * 2
    print(f"fwd  k{k} {Cin:4d}->{Cout:4d} @{H}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {byt/us/1e3:7.1f} GB/s(alg)")
    dy = torch.randn(N, H, H, Cout, device='cuda').to(dt)
    dw = torch.empty(k, k, Cin, Cout, device='cuda')
    npart = ops.conv_wgrad_num_parts(N, H, H, Cin, Cout, k, dt)
    wp = torch.empty(npart * dw.numel(), device='cuda')
    us = t(lambda: ops.conv_bwd_weight(x, dy, k, ops.Affine(sc, sh, 1), dw, wp))
    print(f"wgrd k{k} {Cin:4d}->{Cout:4d} @{H}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  (nsplit {npart})")
