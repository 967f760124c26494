"""Knock-out timing of the persistent 3x3 kernel on the dominant shape (128 -> 128 @128x128, bs32): run once per library variant
(tools/ko_c3.sh builds libmpn_hip_ko{1,2,3}.so with -DMPN_KO=1|2|3: no epilogue / no halo staging in the tile loop / neither)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd import ops

dt = torch.bfloat16
N, H, C = 32, 128, 128
x = torch.randn(N, H, H, C, device='cuda').to(dt)
pc = ops.PackedConv(torch.randn(3, 3, C, C, device='cuda') * 0.05, dt)
aff = ops.Affine(torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1, 1)
y = torch.empty(N, H, H, C, device='cuda', dtype=dt)
part = torch.empty(ops.conv_num_parts(N, H, H, 3) * 2 * C, device='cuda')


def t(fn, n=60):
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


for _ in range(2):
    uf = t(lambda: ops.conv_fwd(x, pc.fwd, C, 3, aff, out=y, stats_part=part))
    up = t(lambda: ops.conv_fwd(x, pc.fwd, C, 3, None, out=y))
fl = 2.0 * N * H * H * C * C * 9
print(f"{os.environ.get('MPN_LIB', 'shipped'):45s} affine+stats {uf:6.1f} us ({fl / uf / 2.5e9:.3f})   plain {up:6.1f} us ({fl / up / 2.5e9:.3f})", flush=True)
