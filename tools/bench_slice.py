"""bn_bwd_reduce / bn_bwd_apply of one 128-channel layer at [32,128,128]: the gradient as a channel slice of the 512-channel
concat gradient (256-byte pieces at a 1 KB pitch) against a dense tensor; cold caches."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

N, H, C = 32, 128, 128
reps = 12


def timed(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


nb = 3
wide = [torch.randn(N, H, H, 4 * C, device="cuda").bfloat16() for _ in range(nb)]
dense = [torch.randn(N, H, H, C, device="cuda").bfloat16() for _ in range(2 * nb)]
xs = [torch.randn(N, H, H, C, device="cuda").bfloat16() for _ in range(2 * nb)]
bn = ops.BNState(torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), 1)
bn.mean.normal_(); bn.invstd.fill_(1.0); bn.scale.copy_(bn.gamma); bn.shift.copy_(bn.beta)
bn.k1 = torch.randn(C, device="cuda") * 0.01; bn.k2 = torch.randn(C, device="cuda") * 0.01
part = torch.empty(ops._lib.lib().mpn_bn_stats_num_parts(N * H * H) * 2 * C, device="cuda")
for name, src in (("slice of 512", lambda i: wide[i % nb][..., :C]), ("dense", lambda i: dense[i % (2 * nb)])):
    tr = timed(lambda i: ops.bn_bwd_reduce_grouped([bn], [src(i)], [xs[i % (2 * nb)]], [part]))
    ta = timed(lambda i: ops.bn_bwd_apply_grouped([bn], [src(i)], [xs[i % (2 * nb)]]))
    print(f"{name:14s}: reduce {tr:6.1f} us   apply {ta:6.1f} us", flush=True)
