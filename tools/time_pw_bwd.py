"""Thin pointwise layers' backward, cold (rotating tensor sets): the separate launches (batch-norm apply pass, weight gradient, data
gradient + reduction) against mpn_conv1x1_bwd_fused (after the apply pass) and mpn_conv1x1_bwd_fused_apply: python tools/time_pw_bwd.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd import ops
dt, B = torch.bfloat16, 32
st = torch.cuda.current_stream()
for name, H, Cin, Cout in [("pw1", 256, 32, 64), ("pw2", 128, 64, 128), ("pw3", 128, 128, 128)]:
    M = B * H * H
    byt = M * (Cin + Cout) * 2
    nset = max(2, min(8, int(1.6e9 // (2 * byt))))
    xs = [torch.randn(B, H, H, Cin, device="cuda").to(dt) for _ in range(nset)]
    gs = [torch.randn(B, H, H, Cout, device="cuda").to(dt) for _ in range(nset)]
    ys = [torch.randn(B, H, H, Cout, device="cuda").to(dt) for _ in range(nset)]
    outs = [torch.empty(B, H, H, Cin, device="cuda", dtype=dt) for _ in range(nset)]
    w = torch.randn(1, 1, Cin, Cout, device="cuda") / Cout ** 0.5
    pc = ops.PackedConv(w, dt)

    def mkbn(C):
        one = lambda: torch.rand(C, device="cuda") + 0.5
        bn = ops.BNState(one(), one(), one(), one(), 2)
        bn.scale.copy_(one()); bn.invstd.copy_(one()); bn.shift.copy_(torch.randn(C, device="cuda") * 0.5); bn.mean.copy_(torch.randn(C, device="cuda") * 0.3)
        bn.k1.copy_(torch.randn(C, device="cuda") * 0.05); bn.k2.copy_(torch.randn(C, device="cuda") * 0.05)
        return bn
    below, own = mkbn(Cin), mkbn(Cout)
    rows = ops.conv_wgrad_num_parts(B, H, H, Cin, Cout, 1, dt)
    wpart = torch.empty(rows * Cin * Cout, device="cuda")
    sp = torch.empty(max(rows, ops.conv_num_parts(B, H, H, 1)) * 2 * Cin, device="cuda")
    dw = torch.zeros(1, 1, Cin, Cout, device="cuda")

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
    ap = timed(lambda i: ops.call("mpn_bn_bwd_apply", ops.ptr(gs[i % nset]), ops.ptr(ys[i % nset]), M, Cout, ops._lib.dtype_code(dt), ops.ptr(own.scale),
                                  ops.ptr(own.shift), ops.ptr(own.mean), ops.ptr(own.invstd), ops.ptr(own.k1), ops.ptr(own.k2), 2, None, ops.stream_ptr()))
    wg = timed(lambda i: ops.conv_bwd_weight(xs[i % nset], gs[i % nset], 1, below.affine, dw, wpart, reduce=False))
    dg = timed(lambda i: ops.conv_bwd_data_bn(gs[i % nset], pc.bwd, Cin, 1, below, xs[i % nset], outs[i % nset], sp)) \
        if ops.conv_bwd_data_bn_supported(Cout, Cin, 1, dt) else float("nan")
    line = f"{name} {Cin:4d}->{Cout:<4d} @{H:3d}: apply {ap:6.1f} + weight {wg:6.1f} + data {dg:6.1f} = {ap + wg + dg:6.1f} us"
    if ops.conv1x1_bwd_fused_supported(Cin, Cout, dt):
        fu = timed(lambda i: ops.conv1x1_bwd_fused(xs[i % nset], gs[i % nset], w, below, outs[i % nset], wpart, sp))
        line += f" | apply + fused {ap + fu:6.1f} (fused {fu:6.1f})"
    if ops.conv1x1_bwd_fused_apply_supported(Cin, Cout, dt):
        fa = timed(lambda i: ops.conv1x1_bwd_fused(xs[i % nset], gs[i % nset], w, below, outs[i % nset], wpart, sp, apply_bn=own, y_raw=ys[i % nset]))
        line += f" | apply folded in {fa:6.1f} us = {(M * (2 * Cin + 2 * Cout) * 2) / fa / 1e6:5.2f} TB/s of its four passes"
    print(line, flush=True)
    del xs, gs, ys, outs
    torch.cuda.empty_cache()
