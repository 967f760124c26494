"""Time the batch-norm backward passes at the bench shapes (bs 32 @ 512x512): python tools/time_bn.py"""
import sys
import torch
import os
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
sys.path.insert(0, os.path.join(_root, 'tools'))
from multiposenet_amd import ops, _lib
from multiposenet_amd.ops import call, ptr, stream_ptr
from time_misc_util import timeit

dt = torch.bfloat16
N = 32
shapes = [(256, 32), (256, 64), (128, 64), (128, 128), (64, 128), (64, 256), (32, 256), (32, 512), (16, 512), (16, 1024),
          (128, 512)]
tot = {"reduce": 0.0, "apply": 0.0}
for (H, C) in shapes:
    M = N * H * H
    x = torch.randn(M, C, device='cuda').to(dt)
    dA = torch.randn(M, C, device='cuda').to(dt)
    one = lambda: torch.rand(C, device='cuda') + 0.5
    bn = ops.BNState(one(), one(), one(), one(), 1)
    for t in (bn.scale, bn.invstd):
        t.copy_(one())
    for t in (bn.shift, bn.mean, bn.k1, bn.k2):
        t.copy_(torch.randn(C, device='cuda') * 0.01)
    bn.dgamma, bn.dbeta = one(), one()
    nparts = _lib.lib().mpn_bn_stats_num_parts(M)
    part = torch.empty(nparts * 2 * C, device='cuda')
    dc = _lib.dtype_code(dt)
    red = lambda: call("mpn_bn_bwd_reduce", ptr(dA), ptr(x), M, C, dc, ptr(bn.scale), ptr(bn.shift), ptr(bn.mean),
                       ptr(bn.invstd), 1, ptr(part), stream_ptr())
    app = lambda: call("mpn_bn_bwd_apply", ptr(dA), ptr(x), M, C, dc, ptr(bn.scale), ptr(bn.shift), ptr(bn.mean),
                       ptr(bn.invstd), ptr(bn.k1), ptr(bn.k2), 1, None, stream_ptr())
    ur, ua = timeit(red), timeit(app)
    byt = M * C * 2
    # same-box ceilings of the same traffic shapes: torch.add(a, b, out=a) = 2 reads + 1 write, torch copy = 1 read + 1 write
    y = torch.empty_like(x)
    ut = timeit(lambda: torch.add(dA, x, out=dA))
    uc = timeit(lambda: y.copy_(x))
    print(f"{H:4d}x{H:<4d} C={C:5d}  reduce {ur:7.1f} us {2 * byt / ur / 1e3:7.0f} GB/s   apply {ua:7.1f} us {3 * byt / ua / 1e3:7.0f} GB/s"
          f"   | torch add (2r+1w) {ut:7.1f} us {3 * byt / ut / 1e3:7.0f} GB/s   copy {uc:7.1f} us {2 * byt / uc / 1e3:7.0f} GB/s")
