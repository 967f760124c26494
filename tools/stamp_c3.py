"""Phase stamps of the persistent 3x3 kernel (diagnostic build: tools/build_variant.sh diag conv_mfma.hip,conv3x3.hip "-DMPN_DIAG",
MPN_LIB=multiposenet_amd/libmpn_hip_diag.so): python tools/stamp_c3.py H Cin Cout [affine+stats 0/1]"""
import ctypes
import sys
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops, _lib
H, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
full = len(sys.argv) < 5 or sys.argv[4] == "1"
dt, N = torch.bfloat16, 32
x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
pc = ops.PackedConv(torch.randn(3, 3, Cin, Cout, device='cuda') * 0.05, dt)
aff = ops.Affine(torch.rand(Cin, device='cuda') + 0.5, torch.randn(Cin, device='cuda') * 0.1, 1) if full else None
y = torch.empty(N, H, H, Cout, device='cuda', dtype=dt)
part = torch.empty(ops.conv_num_parts(N, H, H, 3) * 2 * Cout, device='cuda') if full else None
dbg = torch.zeros(256 * 8, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.mpn_diag_set_conv_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.conv_fwd(x, pc.fwd, Cout, 3, aff, out=y, stats_part=part)
lib.mpn_diag_set_conv_stamps(ctypes.c_void_p(dbg.data_ptr()))
ops.conv_fwd(x, pc.fwd, Cout, 3, aff, out=y, stats_part=part)
torch.cuda.synchronize()
lib.mpn_diag_set_conv_stamps(None)
d = dbg.cpu().numpy().reshape(256, 8).astype(np.float64)
d = d[d[:, 0] > 0]
nch = Cin // 64
cols = [0] + [1 + min(c, 5) for c in range(nch)] + [7]
ph = np.diff(d[:, cols], axis=1)
rt0, rt1 = d[:, 4], d[:, 5]
print("blocks with a third tile:", len(d), " block lifetimes (us, 100 MHz clock): mean %.1f min %.1f max %.1f; first start -> last end %.1f us; start spread %.1f us"
      % ((rt1 - rt0).mean() / 100, (rt1 - rt0).min() / 100, (rt1 - rt0).max() / 100, (rt1.max() - rt0.min()) / 100, (rt0.max() - rt0.min()) / 100))
print("mean cycles per phase (chunks..., epilogue):", np.round(ph.mean(0)), " total", round((d[:, 7] - d[:, 0]).mean()),
      " MFMA cycles per SIMD and tile:", 2 * 16 * 16 * 9 * Cin // 32)
