import sys, os, torch
sys.path.insert(0, "/root/repo")
from multiposenet_amd import ops
dt = torch.bfloat16; B = 32; st = torch.cuda.current_stream()
for (H, C, s) in [(256, 32, 1), (128, 128, 1), (64, 256, 1)]:
    per = 2 * B * H * H * C * 2
    nset = max(2, min(12, int(1.5e9 // per)))
    xs = [torch.randn(B, H, H, C, device="cuda").to(dt) for _ in range(nset)]
    ys = [torch.empty(B, H, H, C, device="cuda", dtype=dt) for _ in range(nset)]
    w = torch.randn(3, 3, C, device="cuda") * 0.2
    aff = ops.Affine(torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, 2)
    part = torch.empty(ops.dwconv_num_parts(B, H, H, C, s, dt) * 2 * C, device="cuda")
    out = []
    for rep in range(3):
        for i in range(nset): ops.dwconv_fwd(xs[i % nset], w, s, aff, out=ys[i % nset], stats_part=part)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(3 * nset): ops.dwconv_fwd(xs[i % nset], w, s, aff, out=ys[i % nset], stats_part=part)
        e1.record(st); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / (3 * nset))
    print(f"{C}ch@{H}: cold fwd us", [round(v, 1) for v in out])
