"""Pointwise 1x1 layers of the bench shape (bs32 @ 512x512, bf16), forward (affine + ReLU6 + statistics) and data gradient,
each launch timed alone with HIP events: python tools/bench_pw.py [iters]"""
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


for (H, Cin, Cout) in [(128, 128, 128), (64, 128, 256), (64, 256, 256), (32, 256, 512), (32, 512, 512), (16, 512, 1024), (16, 1024, 1024)]:
    B = 32
    x = torch.randn(B, H, H, Cin, device="cuda").to(dt)
    pc = ops.PackedConv(torch.randn(1, 1, Cin, Cout, device="cuda") * 0.05, dt)
    aff = ops.Affine(torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1, 2)
    y = torch.empty(B, H, H, Cout, device="cuda", dtype=dt)
    dy = torch.randn(B, H, H, Cout, device="cuda").to(dt)
    dx = torch.empty_like(x)
    part = torch.empty(ops.conv_num_parts(B, H, H, 1) * 2 * Cout, device="cuda")
    fl = 2.0 * B * H * H * Cin * Cout
    byt = (x.numel() + y.numel()) * 2
    tf = timed(lambda: ops.conv_fwd(x, pc.fwd, Cout, 1, aff, out=y, stats_part=part))
    tb = timed(lambda: ops.conv_fwd(dy, pc.bwd, Cin, 1, None, out=dx))
    print(f"{Cin:5d}->{Cout:5d} @{H:3d}x{H:<3d} fwd {tf * 1e6:7.1f} us {fl / tf / 1e12:7.1f} TF ({fl / tf / 2.5e15:.3f}) {byt / tf / 1e9:6.0f} GB/s | "
          f"dgrad {tb * 1e6:7.1f} us {fl / tb / 1e12:7.1f} TF ({fl / tb / 2.5e15:.3f})", flush=True)
