"""Run ONE 3x3 layer back to back for a few seconds (for tools/power_probe.sh): python tools/loop_c3.py H Cin Cout [affine+stats 0/1] [seconds]"""
import sys
import time
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops
H, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
full = len(sys.argv) < 5 or sys.argv[4] == "1"
secs = float(sys.argv[5]) if len(sys.argv) > 5 else 6.0
dt, N = torch.bfloat16, 32
x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
pc = ops.PackedConv(torch.randn(3, 3, Cin, Cout, device='cuda') * 0.05, dt)
aff = ops.Affine(torch.rand(Cin, device='cuda') + 0.5, torch.randn(Cin, device='cuda') * 0.1, 1) if full else None
y = torch.empty(N, H, H, Cout, device='cuda', dtype=dt)
part = torch.empty(ops.conv_num_parts(N, H, H, 3) * 2 * Cout, device='cuda') if full else None
t0 = time.time()
us = []
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500):
        ops.conv_fwd(x, pc.fwd, Cout, 3, aff, out=y, stats_part=part)
    e1.record()
    torch.cuda.synchronize()
    us.append(e0.elapsed_time(e1) * 2.0)
print("3x3 %d -> %d @%d, %s: %d x 500 launches back to back, us per launch: first %.1f, last %.1f, min %.1f" % (
    Cin, Cout, H, "affine + statistics" if full else "plain", len(us), us[0], us[-1], min(us)))
