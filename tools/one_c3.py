"""Run ONE 3x3 layer a few times (for rocprofv3 --pmc): python tools/one_c3.py H Cin Cout [affine+stats 0/1]"""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops
H, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
full = len(sys.argv) < 5 or sys.argv[4] == "1"
dt, N = torch.bfloat16, 32
x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
pc = ops.PackedConv(torch.randn(3, 3, Cin, Cout, device='cuda') * 0.05, dt)
aff = ops.Affine(torch.rand(Cin, device='cuda') + 0.5, torch.randn(Cin, device='cuda') * 0.1, 1) if full else None
y = torch.empty(N, H, H, Cout, device='cuda', dtype=dt)
part = torch.empty(ops.conv_num_parts(N, H, H, 3) * 2 * Cout, device='cuda') if full else None
for _ in range(5):
    ops.conv_fwd(x, pc.fwd, Cout, 3, aff, out=y, stats_part=part)
torch.cuda.synchronize()
