"""Same-box timing of the 3x3 data gradient alone, + the separate batch-norm reduction, against the fused launch
(mpn_conv_bwd_data_bn_grouped) at the bench shape: python tools/time_c3_bnr.py"""
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


dt, N, C = torch.bfloat16, 32, 128
sizes = [(128, 128), (64, 64), (32, 32), (16, 16)]
pc = ops.PackedConv(torch.randn(3, 3, C, C, device='cuda') * 0.05, dt)
dys = [torch.randn(N, h, w, C, device='cuda').to(dt) for h, w in sizes]
xs = [torch.randn(N, h, w, C, device='cuda').to(dt) for h, w in sizes]
outs = [torch.empty_like(x) for x in xs]
bns = []
for _ in sizes:
    bn = ops.BNState(torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1, torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), 1)
    bn.scale.copy_(bn.gamma); bn.shift.copy_(bn.beta); bn.invstd.fill_(1.0)
    bns.append(bn)
parts = [torch.empty(max(ops.conv_num_parts(N, h, w, 3), ops._lib.lib().mpn_bn_stats_num_parts(N * h * w)) * 2 * C, device='cuda') for h, w in sizes]
none4 = [None] * 4
a = t(lambda: ops.conv_fwd_grouped(dys, [pc.bwd] * 4, C, 3, none4, outs, none4))
b = t(lambda: ops.bn_bwd_reduce_grouped(bns, outs, xs, parts))
c = t(lambda: ops.conv_bwd_data_bn_grouped(dys, [pc.bwd] * 4, C, bns, xs, outs, parts))
print(f"grouped 4-level 3x3 data gradient {a:.1f} us + separate reduction {b:.1f} us = {a + b:.1f} us; fused {c:.1f} us", flush=True)
