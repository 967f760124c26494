"""heatmap head forward / backward at [32,128,128,64] bf16 (final_bn + ReLU on load), cold caches."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

N, H, C = 32, 128, 64
M = N * H * H
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


nb = 6
xs = [torch.randn(N, H, H, C, device="cuda").bfloat16() for _ in range(nb)]
dls = [torch.randn(N, H, H, 18, device="cuda") for _ in range(nb)]
dAs = [torch.empty(N, H, H, C, device="cuda", dtype=torch.bfloat16) for _ in range(nb)]
outs = [torch.empty(N, H, H, 18, device="cuda") for _ in range(nb)]
w = torch.randn(1, 1, C, 18, device="cuda") * 0.1
bias = torch.zeros(18, device="cuda")
aff = ops.Affine(torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.3, 1)
nparts = ops._lib.lib().mpn_heatmap_head_bwd_num_parts(M)
part = torch.empty(nparts * (C * 18 + 18), device="cuda")
bnp = torch.empty(nparts * 2 * C, device="cuda")
dw = torch.empty(C * 18 + 18, device="cuda")
tf = timed(lambda i: ops.heatmap_head_fwd(xs[i % nb], w, bias, aff, out=outs[i % nb]))
tb = timed(lambda i: ops.heatmap_head_bwd(xs[i % nb], dls[i % nb], w, aff, dAs[i % nb], dw, part=part, reduce=False))
line = f"head forward {tf:6.1f} us   backward {tb:6.1f} us"
if ops.heatmap_head_bwd_bn_supported(C, torch.bfloat16):
    tbb = timed(lambda i: ops.heatmap_head_bwd(xs[i % nb], dls[i % nb], w, aff, dAs[i % nb], dw, part=part, reduce=False, bn_part=bnp))
    line += f"   backward + batch-norm reduction {tbb:6.1f} us"
print(line, flush=True)
