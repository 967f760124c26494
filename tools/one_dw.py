"""Run ONE depthwise forward shape a few times (for rocprofv3 --pmc). usage: python tools/one_dw.py H C stride"""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

H, C, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt, N = torch.bfloat16, 32
x = torch.randn(N, H, H, C, device='cuda').to(dt)
w = torch.randn(3, 3, C, device='cuda') * 0.2
aff = ops.Affine(torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1, 2)
y = torch.empty(N, H // s, H // s, C, device='cuda', dtype=dt)
part = torch.empty(ops.dwconv_num_parts(N, H, H, C, s, dt) * 2 * C, device='cuda')
for _ in range(6):
    ops.dwconv_fwd(x, w, s, aff, out=y, stats_part=part)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", (x.numel() + y.numel()) * 2)
