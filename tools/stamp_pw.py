"""Phase times of pw_gemm_kernel from a -DMPN_DIAG build: tools/build_variant.sh diag pointwise.hip -DMPN_DIAG;
MPN_LIB=multiposenet_amd/libmpn_hip_diag.so python tools/stamp_pw.py H Cin Cout affine(0/1)"""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops, _lib
H, Cin, Cout, aff = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dt = torch.bfloat16; N = 32
x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
pc = ops.PackedConv(torch.randn(1, 1, Cin, Cout, device='cuda') * 0.05, dt)
a = ops.Affine(torch.rand(Cin, device='cuda') + 0.5, torch.randn(Cin, device='cuda') * 0.1, 2) if aff else None
y = torch.empty(N, H, H, Cout, device='cuda', dtype=dt)
part = torch.empty(ops.conv_num_parts(N, H, H, 1) * 2 * Cout, device='cuda')
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.mpn_diag_set_pw_stamps.argtypes = [ctypes.c_void_p]
for _ in range(20):
    ops.conv_fwd(x, pc.fwd, Cout, 1, a, out=y, stats_part=part)
lib.mpn_diag_set_pw_stamps(ctypes.c_void_p(dbg.data_ptr()))
ops.conv_fwd(x, pc.fwd, Cout, 1, a, out=y, stats_part=part)
torch.cuda.synchronize()
lib.mpn_diag_set_pw_stamps(None)
d = dbg.cpu().numpy().reshape(-1, 8)
d = d[d[:, 0] != 0]
ph = np.diff(d[:, :6].astype(np.int64), axis=1)
rt = (d[:, 7] - d[:, 6]).astype(np.float64) * 10.0   # ns (100 MHz)
names = ["prologue", "main loop", "barrier+O write", "stats", "barrier+copy-out"]
print(f"{len(d)} blocks; per-block cycles (median / max):")
for i, n in enumerate(names):
    print(f"  {n:16s} {np.median(ph[:, i]):9.0f} {ph[:, i].max():9.0f}")
tot = (d[:, 5] - d[:, 0]).astype(np.float64)
print(f"  total      {np.median(tot):9.0f} cycles = {np.median(rt):7.0f} ns -> clock {np.median(tot / rt):.2f} GHz")
print(f"  first start .. last end: {(d[:, 7].max() - d[:, 6].min()) * 10.0:.0f} ns; starts spread {(d[:, 6].max() - d[:, 6].min()) * 10.0:.0f} ns")
