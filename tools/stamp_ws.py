"""Phase times of conv_ws_kernel (s_memtime cycles): MPN_CONV_WS=1 python tools/stamp_ws.py H Cin Cout [affine]"""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from multiposenet_amd import ops, _lib
H, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
use_aff = len(sys.argv) > 4 and sys.argv[4] == "1"
dt = torch.bfloat16; N = 32; k = 3
x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
w = torch.randn(k, k, Cin, Cout, device='cuda') * 0.05
pc = ops.PackedConv(w, dt)
sc = torch.rand(Cin, device='cuda') + 0.5; sh = torch.randn(Cin, device='cuda') * 0.1
y = torch.empty(N, H, H, Cout, device='cuda', dtype=dt)
part = torch.empty(ops.conv_num_parts(N, H, H, k) * 2 * Cout, device='cuda')
aff = ops.Affine(sc, sh, 1) if use_aff else None
st = part if use_aff else None
dbg = torch.zeros(256 * 8, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.mpn_debug_set_conv_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.conv_fwd(x, pc.fwd, Cout, k, aff, out=y, stats_part=st)
lib.mpn_debug_set_conv_stamps(ctypes.c_void_p(dbg.data_ptr()))
ops.conv_fwd(x, pc.fwd, Cout, k, aff, out=y, stats_part=st)
torch.cuda.synchronize()
lib.mpn_debug_set_conv_stamps(None)
d = dbg.cpu().numpy().reshape(256, 8).astype(np.float64)
d = d[d[:, 0] > 0]
tiles = N * (H // 8) * (H // 16) * max(1, Cout // 128) / len(d)
nst = tiles * (Cin // 64) * 9
print(f"blocks {len(d)} units/block {tiles:.1f} stages/block {nst:.0f}")
m = d.mean(0)
print("consumer per stage: mfma-work %.0f  barrier-wait %.0f  | epilogue per unit %.0f" % (m[0] / nst, m[1] / nst, m[2] / tiles))
print("producer per stage: work %.0f  barrier-wait %.0f  vmcnt-wait %.0f" % (m[3] / nst, m[4] / nst, m[5] / nst))
