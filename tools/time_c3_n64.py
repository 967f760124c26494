"""Same-box A/B of the 64-channel-tile variant of the persistent 3x3 kernel against the tiled kernel (MPN_LIB selects the library)."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops


def t(fn, n=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


dt = torch.bfloat16
for (N, H, W, Cin, Cout) in [(32, 128, 128, 512, 64), (32, 128, 128, 64, 512), (16, 112, 176, 64, 64), (16, 56, 88, 64, 64), (16, 28, 44, 64, 64)]:
    x = torch.randn(N, H, W, Cin, device='cuda').to(dt)
    pc = ops.PackedConv(torch.randn(3, 3, Cin, Cout, device='cuda') * 0.05, dt)
    aff = ops.Affine(torch.rand(Cin, device='cuda') + 0.5, torch.randn(Cin, device='cuda') * 0.1, 1)
    y = torch.empty(N, H, W, Cout, device='cuda', dtype=dt)
    dy = torch.randn(N, H, W, Cout, device='cuda').to(dt)
    dx = torch.empty_like(x)
    part = torch.empty(ops.conv_num_parts(N, H, W, 3) * 2 * Cout, device='cuda')
    fl = 2.0 * N * H * W * Cin * Cout * 9
    uf = t(lambda: ops.conv_fwd(x, pc.fwd, Cout, 3, aff, out=y, stats_part=part))
    ub = t(lambda: ops.conv_fwd(dy, pc.bwd, Cin, 3, None, out=dx))
    print(f"3x3 {Cin:4d}->{Cout:4d} @{H:3d}x{W:3d}: fwd {uf:7.1f} us ({fl / uf / 2.5e9:.3f}) | dgrad {ub:7.1f} us ({fl / ub / 2.5e9:.3f})", flush=True)
