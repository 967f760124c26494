"""Conv2d_0 (3x3 stride 2, 3 -> 32) forward (+ batch-norm statistics) and weight gradient at [32,512,512,3] f32 images, bf16
activations, cold caches (rotating buffers)."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

N, H, C0 = 32, 512, 32
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


imgs = [torch.rand(N, H, H, 3, device="cuda") for _ in range(4)]
w = torch.randn(3, 3, 3, C0, device="cuda") * 0.2
outs = [torch.empty(N, H // 2, H // 2, C0, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
dys = [torch.randn(N, H // 2, H // 2, C0, device="cuda").bfloat16() for _ in range(4)]
rows = ops.stem_conv_fwd_num_parts(N, H, H, C0, torch.bfloat16)
part = torch.empty(rows * 2 * C0, device="cuda")
nparts = ops._lib.lib().mpn_stem_conv_wgrad_num_parts(N, H, H)
wpart = torch.empty(nparts * 27 * C0, device="cuda")
dw = torch.empty_like(w)
tf = timed(lambda i: ops.stem_conv_fwd(imgs[i % 4], w, C0, torch.bfloat16, out=outs[i % 4], stats_part=part))
tw = timed(lambda i: ops.stem_conv_bwd_weight(imgs[i % 4], dys[i % 4], dw, part=wpart, reduce=False))
bf = (N * H * H * 3 * 4 + N * (H // 2) ** 2 * C0 * 2) / 1e6
print(f"stem forward + statistics {tf:6.1f} us ({bf / tf:5.2f} TB/s of {bf:.0f} MB)   weight gradient {tw:6.1f} us ({bf / tw:5.2f} TB/s), {nparts} slab rows", flush=True)
