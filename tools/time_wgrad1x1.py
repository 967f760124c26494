"""1x1 weight gradients of the bench shape (bs32 @ 512x512, bf16), cold (rotating tensor sets): python tools/time_wgrad1x1.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd import ops
dt = torch.bfloat16
B = 32
st = torch.cuda.current_stream()
LAYERS = [("pw1", 256, 32, 64), ("pw2", 128, 64, 128), ("pw3", 128, 128, 128), ("pw4", 64, 128, 256), ("pw5", 64, 256, 256),
          ("pw6", 32, 256, 512), ("pw7-11", 32, 512, 512), ("pw12", 16, 512, 1024), ("pw13", 16, 1024, 1024),
          ("lat5", 16, 1024, 128), ("lat4", 32, 512, 128), ("lat3", 64, 256, 128), ("lat2", 128, 128, 128)]
for name, H, Cin, Cout in LAYERS:
    byt = B * H * H * (Cin + Cout) * 2
    nset = max(2, min(12, int(1.2e9 // byt)))
    xs = [torch.randn(B, H, H, Cin, device="cuda").to(dt) for _ in range(nset)]
    dys = [torch.randn(B, H, H, Cout, device="cuda").to(dt) for _ in range(nset)]
    aff = ops.Affine(torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1, 2)
    dw = torch.empty(1, 1, Cin, Cout, device="cuda")
    nparts = ops.conv_wgrad_num_parts(B, H, H, Cin, Cout, 1, dt)
    slab = torch.empty(nparts * Cin * Cout, device="cuda")
    fn = lambda i: ops.conv_bwd_weight(xs[i % nset], dys[i % nset], 1, aff, dw, slab, reduce=False)
    for i in range(nset):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 3 * nset
    e0.record(st)
    for i in range(iters):
        fn(i)
    e1.record(st)
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    fl = 2.0 * B * H * H * Cin * Cout
    print(f"{name:7s} {Cin:5d}->{Cout:<5d} @{H:3d}: {t * 1e6:6.1f} us  {fl / t / 1e12:6.1f} TFLOP/s  {byt / t / 1e9:6.0f} GB/s  slabs {nparts}", flush=True)
    del xs, dys
    torch.cuda.empty_cache()
