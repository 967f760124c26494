import sys, ctypes, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops, _lib
H, Cin, Cout, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dt = torch.bfloat16; N = 32
x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
w = torch.randn(k, k, Cin, Cout, device='cuda') * 0.05
pc = ops.PackedConv(w, dt)
sc = torch.rand(Cin, device='cuda') + 0.5; sh = torch.randn(Cin, device='cuda') * 0.1
y = torch.empty(N, H, H, Cout, device='cuda', dtype=dt)
nparts = ops.conv_num_parts(N, H, H, k)
part = torch.empty(nparts * 2 * Cout, device='cuda')
nblk = nparts * max(1, Cout // 128)   # (upper bound: the 256-pixel 3x3 kernel launches half of it)
dbg = torch.zeros(nblk * 8, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.mpn_diag_set_conv_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.conv_fwd(x, pc.fwd, Cout, k, ops.Affine(sc, sh, 1), out=y, stats_part=part)
lib.mpn_diag_set_conv_stamps(ctypes.c_void_p(dbg.data_ptr()))
ops.conv_fwd(x, pc.fwd, Cout, k, ops.Affine(sc, sh, 1), out=y, stats_part=part)
torch.cuda.synchronize()
lib.mpn_diag_set_conv_stamps(None)
d = dbg.cpu().numpy().reshape(nblk, 8).astype(np.float64)
d = d[d[:, 0] > 0]   # (the 256-pixel variant launches half the blocks)
nblk = len(d)
t0 = d[:, 0].min()
ph = np.diff(d[:, :5], axis=1)   # stage A, main loop, epilogue, stats
print("blocks", nblk, "kernel span (ticks of 100MHz?)", (d[:, 4].max() - t0))
print("mean phase ticks: stageA %.0f  main %.0f  epilogue %.0f  stats %.0f  | total %.0f" % (*ph.mean(0), (d[:, 4] - d[:, 0]).mean()))
rt = (d[:, 6] - d[:, 5])
ok = rt > 0
clk = np.median((d[ok, 4] - d[ok, 0]) / rt[ok]) * 100e6
print("in-kernel clock %.2f GHz (s_memtime / s_memrealtime x 100 MHz, median over blocks); block time %.1f us" % (clk / 1e9, np.median(rt[ok]) / 100.0))
print("start-time spread of blocks (first/last start):", d[:, 0].min() - t0, d[:, 0].max() - t0)
