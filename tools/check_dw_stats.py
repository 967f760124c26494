import sys, torch, numpy as np
sys.path.insert(0, '.')
from multiposenet_amd import ops
torch.manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
  for (N, H, C, s) in [(2, 64, 32, 1), (2, 64, 64, 2), (2, 32, 128, 1), (2, 32, 128, 2), (2, 16, 256, 1), (2, 16, 256, 2), (2, 8, 512, 1), (2, 8, 512, 2), (2, 4, 1024, 1), (3, 20, 40, 1), (1, 9, 24, 2)]:
    x = torch.randn(N, H, H, C, device='cuda').to(dt)
    w = torch.randn(3, 3, C, device='cuda') * 0.3
    sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda') * 0.2
    OH = (H + s - 1) // s
    y = torch.empty(N, OH, OH, C, device='cuda', dtype=dt)
    nparts = ops.dwconv_num_parts(N, H, H, C, s, dt)
    part = torch.full((nparts * 2 * C,), float('nan'), device='cuda')
    ops.dwconv_fwd(x, w, s, ops.Affine(sc, sh, 2), out=y, stats_part=part)
    p = part.view(nparts, 2, C)
    ssum, ssq = p[:, 0].double().sum(0), p[:, 1].double().sum(0)
    # reference from the f32 accumulators is not available: compare with sums of the stored y (rounded) loosely, and
    # with an f32 torch conv exactly for f32
    a = torch.clamp(x.float() * sc + sh, 0, 6).permute(0, 3, 1, 2)
    if s == 2:
        a = torch.nn.functional.pad(a, (0, 1 if H % 2 == 0 else 1, 0, 1 if H % 2 == 0 else 1)) if H % 2 == 0 else torch.nn.functional.pad(a, (1, 1, 1, 1))
        ref = torch.nn.functional.conv2d(a.double(), w.permute(2, 0, 1).unsqueeze(1).double(), stride=2, groups=C)
    else:
        ref = torch.nn.functional.conv2d(a.double(), w.permute(2, 0, 1).unsqueeze(1).double(), padding=1, groups=C)
    ref = ref.permute(0, 2, 3, 1)
    ey = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    es = (ssum - ref.sum((0, 1, 2))).abs().max().item() / ref.abs().sum((0,1,2)).max().item()
    eq = (ssq - (ref ** 2).sum((0, 1, 2))).abs().max().item() / (ref ** 2).sum((0,1,2)).max().item()
    print(dt, (N, H, C, s), "nparts", nparts, "y %.2e sum %.2e sq %.2e" % (ey, es, eq), "nan" if torch.isnan(part).any() else "")
