import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from multiposenet_amd import ops, _lib
H, Cin, Cout, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dt = torch.bfloat16; N = 32
x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
dy = torch.randn(N, H, H, Cout, device='cuda').to(dt)
sc = torch.rand(Cin, device='cuda') + 0.5; sh = torch.randn(Cin, device='cuda') * 0.1
dw = torch.empty(k, k, Cin, Cout, device='cuda')
npart = ops.conv_wgrad_num_parts(N, H, H, Cin, Cout, k, dt)
wp = torch.empty(npart * dw.numel(), device='cuda')
dbg = torch.zeros(4096 * 4, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.mpn_debug_set_wgrad_stamps.argtypes = [ctypes.c_void_p]
for _ in range(2):
    ops.conv_bwd_weight(x, dy, k, ops.Affine(sc, sh, 1), dw, wp)
lib.mpn_debug_set_wgrad_stamps(ctypes.c_void_p(dbg.data_ptr()))
ops.conv_bwd_weight(x, dy, k, ops.Affine(sc, sh, 1), dw, wp)
torch.cuda.synchronize()
lib.mpn_debug_set_wgrad_stamps(None)
d = dbg.cpu().numpy().reshape(-1, 4).astype(np.float64)
d = d[d[:, 0] > 0]
print("blocks", len(d), "nsplit", npart, "mean stage ticks %.0f  mma ticks %.0f" % (d[:, 0].mean(), d[:, 1].mean()))
