"""Same-box wall time of 3x3 forward launches (bf16): python tools/time_c3.py [N H W Cin Cout]... (default: the bench layer
[32,128,128,128] -> 128), plain and with affine + ReLU on load + statistics."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops
dt = torch.bfloat16
a = [int(v) for v in sys.argv[1:]]
shapes = [tuple(a[i:i + 5]) for i in range(0, len(a), 5)] or [(32, 128, 128, 128, 128)]
for N, H, W, Cin, Cout in shapes:
    x = torch.randn(N, H, W, Cin, device='cuda').to(dt)
    pc = ops.PackedConv(torch.randn(3, 3, Cin, Cout, device='cuda') * 0.05, dt)
    aff = ops.Affine(torch.rand(Cin, device='cuda') + 0.5, torch.randn(Cin, device='cuda') * 0.1, 1)
    y = torch.empty(N, H, W, Cout, device='cuda', dtype=dt)
    part = torch.empty(ops.conv_num_parts(N, H, W, 3) * 2 * Cout, device='cuda')
    flop = 2.0 * N * H * W * Cin * 9 * Cout
    for name, fn in (("plain", lambda: ops.conv_fwd(x, pc.fwd, Cout, 3, None, out=y)),
                     ("affine + ReLU on load, statistics", lambda: ops.conv_fwd(x, pc.fwd, Cout, 3, aff, out=y, stats_part=part))):
        for _ in range(600):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 50)
        ts.sort()
        print(f"conv3x3 {Cin} -> {Cout} @ {N}x{H}x{W}, {name}: {ts[4]:.1f} us per launch = {flop / ts[4] / 1e6:.1f} TFLOP/s", flush=True)
