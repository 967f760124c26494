"""Same-box wall time of the shipped 3x3 forward at the bench layer ([32,128,128,128] -> 128, bf16), plain and with affine + statistics:
python tools/time_c3.py"""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops
dt, N, H, C = torch.bfloat16, 32, 128, 128
x = torch.randn(N, H, H, C, device='cuda').to(dt)
pc = ops.PackedConv(torch.randn(3, 3, C, C, device='cuda') * 0.05, dt)
aff = ops.Affine(torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1, 1)
y = torch.empty(N, H, H, C, device='cuda', dtype=dt)
part = torch.empty(ops.conv_num_parts(N, H, H, 3) * 2 * C, device='cuda')
flop = 2.0 * N * H * H * C * 9 * C
for name, fn in (("plain", lambda: ops.conv_fwd(x, pc.fwd, C, 3, None, out=y)),
                 ("affine + ReLU on load, statistics", lambda: ops.conv_fwd(x, pc.fwd, C, 3, aff, out=y, stats_part=part))):
    for _ in range(2000):
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
    print(f"shipped conv3x3_cs 128 -> 128 @128^2 x 32, {name}: {ts[4]:.1f} us per launch = {flop / ts[4] / 1e6:.1f} TFLOP/s", flush=True)
