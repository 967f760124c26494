"""Depthwise 3x3 layers of the bench shape (bs32 @ 512x512, bf16) timed COLD: every launch works on another set of tensors,
the sets together far larger than the 256 MB memory-side cache (what the kernels see inside the train step, where each runs
behind producers that have just pushed hundreds of MB through that cache) - next to the same launches re-run on ONE set (what
round 3's bench leg printed) and a torch copy of the same bytes under the same rotation.   python tools/bench_dw_cold.py [iters]"""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dt = torch.bfloat16
st = torch.cuda.current_stream()


def timed(fn, n):
    for i in range(n):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(iters):
        fn(i)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


B = 32
print("layer | pass: cold us (frac of 8 TB/s) / same-buffers us (frac) | torch copy of the same bytes cold (frac)")
for (H, C, s) in [(256, 32, 1), (256, 64, 2), (128, 128, 1), (128, 128, 2), (64, 256, 1), (64, 256, 2), (32, 512, 1), (32, 512, 2), (16, 1024, 1)]:
    OH = H // s
    per_set = (B * H * H * C + B * OH * OH * C) * 2
    nset = max(2, min(12, int(1.5e9 // per_set)))
    xs = [torch.randn(B, H, H, C, device="cuda").to(dt) for _ in range(nset)]
    ys = [torch.empty(B, OH, OH, C, device="cuda", dtype=dt) for _ in range(nset)]
    dys = [torch.randn(B, OH, OH, C, device="cuda").to(dt) for _ in range(nset)]
    dxs = [torch.empty(B, H, H, C, device="cuda", dtype=dt) for _ in range(nset)]
    w = torch.randn(3, 3, C, device="cuda") * 0.2
    aff = ops.Affine(torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, 2)
    dw = torch.empty(3, 3, C, device="cuda")
    part = torch.empty(ops.dwconv_num_parts(B, H, H, C, s, dt) * 2 * C, device="cuda")
    slab = torch.empty(ops.dwconv_wgrad_num_parts(B, H, H, C, s, dt) * 9 * C, device="cuda")
    byt = per_set
    line = f"{C:5d}ch @{H:3d}x{H:<3d} s{s} ({nset} sets of {per_set / 1e6:.0f} MB)"
    for name, fn in (("fwd", lambda i: ops.dwconv_fwd(xs[i % nset], w, s, aff, out=ys[i % nset], stats_part=part)),
                     ("dgrad", lambda i: ops.dwconv_bwd_data(dys[i % nset], w, (H, H), s, out=dxs[i % nset])),
                     ("wgrad", lambda i: ops.dwconv_bwd_weight(xs[i % nset], dys[i % nset], s, aff, dw, slab, reduce=False))):
        tc = timed(fn, nset)
        tw = timed(lambda i: fn(0), 3)
        line += f" | {name}: {tc * 1e6:6.1f} us ({byt / tc / 8e12:.3f}) / {tw * 1e6:6.1f} ({byt / tw / 8e12:.3f})"
    # the copy moves the same number of bytes as the forward pass: read x, write a tensor of y's size (stride-2: a quarter)
    tcp = timed(lambda i: ys[i % nset].copy_(xs[i % nset][:, ::s, ::s, :]) if s == 2 else ys[i % nset].copy_(xs[i % nset]), nset)
    rb = (B * OH * OH * C * 2) * 2 if s == 2 else byt
    line += f" | copy {tcp * 1e6:6.1f} us ({rb / tcp / 8e12:.3f})"
    print(line, flush=True)
    del xs, ys, dys, dxs
    torch.cuda.empty_cache()
