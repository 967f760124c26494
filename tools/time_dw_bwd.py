"""Stride-1 depthwise backward, cold (rotating tensor sets): the fused walk (mpn_dwconv_bwd_fused) against the separate weight-gradient
and data-gradient (+ reduction) launches: python tools/time_dw_bwd.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd import ops
dt, B = torch.bfloat16, 32
st = torch.cuda.current_stream()
for name, H, C in [("dw1", 256, 32), ("dw3", 128, 128), ("dw5", 64, 256), ("dw7-11", 32, 512), ("dw13", 16, 1024)]:
    byt = B * H * H * C * 2
    nset = max(2, min(10, int(1.6e9 // (3 * byt))))
    xs = [torch.randn(B, H, H, C, device="cuda").to(dt) for _ in range(nset)]
    dys = [torch.randn(B, H, H, C, device="cuda").to(dt) for _ in range(nset)]
    outs = [torch.empty(B, H, H, C, device="cuda", dtype=dt) for _ in range(nset)]
    w = torch.randn(3, 3, C, device="cuda") / 3
    one = lambda: torch.rand(C, device="cuda") + 0.5
    bn = ops.BNState(one(), one(), one(), one(), 2)
    bn.scale.copy_(one()); bn.invstd.copy_(one()); bn.shift.copy_(torch.randn(C, device="cuda") * 0.5); bn.mean.copy_(torch.randn(C, device="cuda") * 0.3)
    rows = ops.dwconv_wgrad_num_parts(B, H, H, C, 1, dt)
    wpart = torch.empty(rows * 9 * C, device="cuda"); sp = torch.empty(max(rows, ops.dwconv_bwd_data_bn_num_parts(B, H, H, C, 1, dt)) * 2 * C, device="cuda")
    dw = torch.zeros(3, 3, C, device="cuda")

    def timed(fn):
        for i in range(nset):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 3 * nset
        e0.record(st)
        for i in range(it):
            fn(i)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / it
    fused = timed(lambda i: ops.dwconv_bwd_fused(xs[i % nset], dys[i % nset], w, bn, dw, out=outs[i % nset], wpart=wpart, bn_part=sp, reduce=False))
    wg = timed(lambda i: ops.dwconv_bwd_weight(xs[i % nset], dys[i % nset], 1, bn.affine, dw, wpart, reduce=False))
    dg = timed(lambda i: ops.dwconv_bwd_data(dys[i % nset], w, (H, H), 1, out=outs[i % nset], bn=bn, x_bn=xs[i % nset], part=sp))
    print(f"{name:7s} {C:5d} ch @{H:3d}: fused {fused:6.1f} us = {3 * byt / fused / 1e6:5.2f} TB/s of its 3 passes ({3 * byt / fused / 8e6:.3f} of 8 TB/s)   "
          f"separate: weight {wg:6.1f} + data {dg:6.1f} = {wg + dg:6.1f} us", flush=True)
    del xs, dys, outs
    torch.cuda.empty_cache()
