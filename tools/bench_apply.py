"""bn_bwd_apply against plain streams of the same bytes (two 16-bit reads + one write per element), per layer shape.
Rotates through enough tensors that nothing is served from the 256 MB Infinity Cache; HIP events around a run of launches.
    python tools/bench_apply.py [reps]"""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [(32, 256, 256, 64), (32, 128, 128, 128), (32, 64, 64, 256), (32, 32, 32, 512), (32, 128, 128, 64)]


def timed(fn, n):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for shp in SHAPES:
    C = shp[-1]
    nbytes = 2 * shp[0] * shp[1] * shp[2] * C
    nbuf = max(3, int(1.2e9 // (2 * nbytes)))
    ds = [torch.randn(shp, device="cuda").bfloat16() for _ in range(nbuf)]
    xs = [torch.randn(shp, device="cuda").bfloat16() for _ in range(nbuf)]
    bn = ops.BNState(torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, torch.zeros(C, device="cuda"),
                     torch.ones(C, device="cuda"), 2)
    bn.mean.normal_(); bn.invstd.fill_(1.0); bn.scale.copy_(bn.gamma); bn.shift.copy_(bn.beta)
    bn.k1 = torch.randn(C, device="cuda") * 0.01; bn.k2 = torch.randn(C, device="cuda") * 0.01
    t_apply = timed(lambda i: ops.bn_bwd_apply(bn, ds[i % nbuf], xs[i % nbuf]), reps)
    t_add = timed(lambda i: torch.add(ds[i % nbuf], xs[i % nbuf], out=ds[i % nbuf]), reps)
    t_add2 = timed(lambda i: torch.add(ds[i % nbuf], xs[i % nbuf], out=ds[(i + 1) % nbuf]), reps)
    t_copy = timed(lambda i: ds[i % nbuf].copy_(xs[i % nbuf]), reps)
    tb = lambda us, k: k * nbytes / us / 1e6
    print(f"{shp}: apply {t_apply:7.1f} us {tb(t_apply, 3):5.2f} TB/s | torch add in place {t_add:7.1f} us {tb(t_add, 3):5.2f} | "
          f"add to a third tensor {t_add2:7.1f} us {tb(t_add2, 3):5.2f} | copy {t_copy:7.1f} us {tb(t_copy, 2):5.2f} TB/s", flush=True)
    del ds, xs
