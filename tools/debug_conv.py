import numpy as np, torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from multiposenet_amd import ops
from oracle import network as onet
torch.manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    for (k, Cin, Cout, H, W) in [(1, 32, 64, 16, 16), (1, 128, 128, 24, 16), (3, 128, 128, 16, 16), (3, 128, 128, 12, 16)]:
        x = torch.randn(1, H, W, Cin).to(dtype).float()
        w = (torch.randn(k, k, Cin, Cout) / (k * Cin ** 0.5)).to(dtype).float()
        want = onet.conv2d_same(x.permute(0, 3, 1, 2), w).permute(0, 2, 3, 1)
        pc = ops.PackedConv(w.cuda(), dtype)
        got = ops.conv_fwd(x.to(dtype).cuda(), pc.fwd, Cout, k).float().cpu()
        err = (got - want).abs()
        print(dtype, k, Cin, Cout, "maxerr", float(err.max()), "nan", int(torch.isnan(got).sum()))
        bad = (err > 1e-2) | torch.isnan(got)
        rows = bad.reshape(-1, Cout).any(1).nonzero().flatten().tolist()
        cols = bad.reshape(-1, Cout).any(0).nonzero().flatten().tolist()
        print("  bad rows", rows[:40], len(rows), "bad cols", cols[:40], len(cols))
        print("  got[0,0,0,:8]", got[0, 0, 0, :8].tolist())
        print("  want[0,0,0,:8]", want[0, 0, 0, :8].tolist())
        # ratio test
        print("  got/want row0", (got[0,0,0,:8] / want[0,0,0,:8]).tolist())
