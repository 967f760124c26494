"""Time the memory-bound helper kernels at the bench shape: python tools/time_misc.py"""
import sys
import torch
sys.path.insert(0, '.')
from multiposenet_amd import ops

dt = torch.bfloat16
N = 32


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay(); g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * n)


cat = torch.empty(N, 128, 128, 512, device='cuda', dtype=dt)
dcat = torch.randn(N, 128, 128, 512, device='cuda').to(dt)
for lvl, (s, u) in enumerate([(128, 1), (64, 2), (32, 4), (16, 8)]):
    x = torch.randn(N, s, s, 128, device='cuda').to(dt)
    dx = torch.empty_like(x)
    sc = torch.rand(128, device='cuda') + 0.5
    sh = torch.randn(128, device='cuda') * 0.1
    us = timeit(lambda: ops.bilinear_up_fwd(x, u, cat, lvl * 128, ops.Affine(sc, sh, 1)))
    byt = x.numel() * 2 + N * 128 * 128 * 128 * 2
    print(f"bilinear fwd u={u}: {us:7.1f} us  {byt / us / 1e3:7.1f} GB/s")
    us = timeit(lambda: ops.bilinear_up_bwd(dcat, u, lvl * 128, 128, out=dx))
    print(f"bilinear bwd u={u}: {us:7.1f} us  {byt / us / 1e3:7.1f} GB/s")
x = torch.randn(N, 128, 128, 128, device='cuda').to(dt)
print("fwd u=1 no affine:", timeit(lambda: ops.bilinear_up_fwd(x, 1, cat, 0, None)))
aff = ops.Affine(sc, sh, 1)
print("fwd u=1 affine (prebuilt):", timeit(lambda: ops.bilinear_up_fwd(x, 1, cat, 0, aff)))
y2 = torch.empty(N, 128, 128, 128, device='cuda', dtype=dt)
print("fwd u=1 affine, dense out:", timeit(lambda: ops.bilinear_up_fwd(x, 1, y2, 0, aff)))
M = N * 128 * 128
xh = torch.randn(M, 64, device='cuda').to(dt)
dl = torch.randn(M, 18, device='cuda')
wh = torch.randn(64, 18, device='cuda') * 0.01
sc64 = torch.rand(64, device='cuda') + 0.5
sh64 = torch.randn(64, device='cuda') * 0.1
xh4 = xh.view(N, 128, 128, 64)
dA = torch.empty_like(xh4)
dwdb = torch.empty(64 * 18 + 18, device='cuda')
aff64 = ops.Affine(sc64, sh64, 1)
us = timeit(lambda: ops.heatmap_head_bwd(xh4, dl, wh, aff64, dA, dwdb))
print(f"head_bwd (+reduce): {us:7.1f} us")
img = torch.rand(N, 512, 512, 3, device='cuda')
w0 = torch.randn(3, 3, 3, 32, device='cuda') * 0.1
y0 = torch.empty(N, 256, 256, 32, device='cuda', dtype=dt)
us = timeit(lambda: ops.stem_conv_fwd(img, w0, 32, dt, out=y0))
print(f"stem fwd: {us:7.1f} us")
dy0 = torch.randn(N, 256, 256, 32, device='cuda').to(dt)
dw0 = torch.empty(3, 3, 3, 32, device='cuda')
import inspect
us = timeit(lambda: ops.stem_conv_bwd_weight(img, dy0, dw0))
print(f"stem wgrad (+reduce): {us:7.1f} us")
