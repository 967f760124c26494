"""Phase times inside conv_wgrad_bf16_kernel (s_memtime, 100 MHz): python tools/stamp_wgrad.py H Cin Cout k"""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops, _lib
H, Cin, Cout, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dt = torch.bfloat16; N = 32
x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
dy = torch.randn(N, H, H, Cout, device='cuda').to(dt)
sc = torch.rand(Cin, device='cuda') + 0.5; sh = torch.randn(Cin, device='cuda') * 0.1
dw = torch.empty(k, k, Cin, Cout, device='cuda')
npart = ops.conv_wgrad_num_parts(N, H, H, Cin, Cout, k, dt)
wp = torch.empty(npart * dw.numel(), device='cuda')
nblk = 4096
dbg = torch.zeros(nblk * 8 * 4, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.mpn_diag_set_wgrad_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.conv_bwd_weight(x, dy, k, ops.Affine(sc, sh, 1), dw, wp)
lib.mpn_diag_set_wgrad_stamps(ctypes.c_void_p(dbg.data_ptr()))
ops.conv_bwd_weight(x, dy, k, ops.Affine(sc, sh, 1), dw, wp)
torch.cuda.synchronize()
lib.mpn_diag_set_wgrad_stamps(None)
d = dbg.cpu().numpy().reshape(nblk, 8, 4).astype(np.float64)
used = d.sum(axis=(1, 2)) > 0
d = d[used] / 100.0   # us
print("blocks", int(used.sum()), "nsplit", npart)
print("mean per-wave us: commit %.1f  load-issue %.1f  barrier %.1f  mfma %.1f  | total %.1f" % (*d.mean((0, 1)), d.sum(2).mean()))
print("per-wave (block 0):", np.round(d[0], 1).tolist())
