"""Does the relative placement of the two operand tensors matter? bn_bwd_apply / torch add with x at base + offset bytes."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

shp = (32, 128, 128, 128)
C = shp[-1]
n = shp[0] * shp[1] * shp[2] * C
nbuf = 5
reps = 20


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


bn = ops.BNState(torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, torch.zeros(C, device="cuda"),
                 torch.ones(C, device="cuda"), 2)
bn.mean.normal_(); bn.invstd.fill_(1.0); bn.scale.copy_(bn.gamma); bn.shift.copy_(bn.beta)
bn.k1 = torch.randn(C, device="cuda") * 0.01; bn.k2 = torch.randn(C, device="cuda") * 0.01
pad = 1 << 20   # elements of slack per buffer
big_d = [torch.randn(n + pad, device="cuda").bfloat16() for _ in range(nbuf)]
big_x = [torch.randn(n + pad, device="cuda").bfloat16() for _ in range(nbuf)]
print("bases", [hex(t.data_ptr()) for t in big_d[:2]], [hex(t.data_ptr()) for t in big_x[:2]])
for off in (0, 128, 256, 1024, 2048, 4096 + 256, 16384, 65536 + 1024, 1 << 19, (1 << 19) + 4096 + 256):   # bytes
    e = off // 2
    ds = [t[:n].view(shp) for t in big_d]
    xs = [t[e:e + n].view(shp) for t in big_x]
    ta = timed(lambda i: ops.bn_bwd_apply(bn, ds[i % nbuf], xs[i % nbuf]))
    tt = timed(lambda i: torch.add(ds[i % nbuf], xs[i % nbuf], out=ds[i % nbuf]))
    print(f"x offset {off:8d} B: apply {ta:6.1f} us  torch add in place {tt:6.1f} us", flush=True)
