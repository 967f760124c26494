"""The 3x3 weight gradients of the step's shapes, cold (rotating tensor sets), HIP events: python tools/time_wgrad3x3.py"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd import ops

dt = torch.bfloat16
N = 32
for (H, Cin, Cout) in [(128, 128, 128), (64, 128, 128), (128, 512, 64)]:
    sets = 3
    xs = [torch.randn(N, H, H, Cin, device="cuda").to(dt) for _ in range(sets)]
    dys = [torch.randn(N, H, H, Cout, device="cuda").to(dt) for _ in range(sets)]
    sc = torch.rand(Cin, device="cuda") + 0.5; sh = torch.randn(Cin, device="cuda") * 0.1
    dw = torch.empty(3, 3, Cin, Cout, device="cuda")
    npart = ops.conv_wgrad_num_parts(N, H, H, Cin, Cout, 3, dt)
    wp = torch.empty(npart * dw.numel(), device="cuda")
    fn = lambda i: ops.conv_bwd_weight(xs[i % sets], dys[i % sets], 3, ops.Affine(sc, sh, 1), dw, wp, reduce=False)
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 12
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 2.0 * N * H * H * Cin * Cout * 9
    print(f"3x3 wgrad {Cin:4d}->{Cout:4d} @{H:3d}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s ({fl / us / 2.5e9:.3f})  slab {npart * dw.numel() * 4 / 1e6:.0f} MB", flush=True)
