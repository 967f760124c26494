"""Run ONE conv shape a few times (for rocprofv3 --pmc). usage: python tools/one_conv.py [fwd|wgrad] H Cin Cout k"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd import ops
mode, H, Cin, Cout, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dt = torch.bfloat16
N = 32
x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
w = torch.randn(k, k, Cin, Cout, device='cuda') * 0.05
pc = ops.PackedConv(w, dt)
sc = torch.rand(Cin, device='cuda') + 0.5; sh = torch.randn(Cin, device='cuda') * 0.1
y = torch.empty(N, H, H, Cout, device='cuda', dtype=dt)
part = torch.empty(ops.conv_num_parts(N, H, H, k) * 2 * Cout, device='cuda')
dy = torch.randn(N, H, H, Cout, device='cuda').to(dt)
dw = torch.empty(k, k, Cin, Cout, device='cuda')
npart = ops.conv_wgrad_num_parts(N, H, H, Cin, Cout, k, dt)
wp = torch.empty(npart * dw.numel(), device='cuda')
for _ in range(5):
    if mode == 'fwd':
        ops.conv_fwd(x, pc.fwd, Cout, k, ops.Affine(sc, sh, 1), out=y, stats_part=part)
    else:
        ops.conv_bwd_weight(x, dy, k, ops.Affine(sc, sh, 1), dw, wp)
torch.cuda.synchronize()
