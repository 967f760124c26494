"""Per-wave stage stamps of the persistent 3x3 kernel's third tile (diagnostic build: tools/build_variant.sh diag conv_mfma.hip,conv3x3.hip
"-DMPN_DIAG"; MPN_LIB=multiposenet_amd/libmpn_hip_diag.so python tools/stamp_c3_stages.py [affine+stats 0/1]): where a stage's time goes -
its body (fragment reads, 48 MFMAs, staging work), the counted wait for this wave's weight pieces, the block barrier."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd import ops, _lib
full = len(sys.argv) < 2 or sys.argv[1] == "1"
dt, N, H, C = torch.bfloat16, 32, 128, 128
x = torch.randn(N, H, H, C, device='cuda').to(dt)
pc = ops.PackedConv(torch.randn(3, 3, C, C, device='cuda') * 0.05, dt)
aff = ops.Affine(torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1, 1) if full else None
y = torch.empty(N, H, H, C, device='cuda', dtype=dt)
part = torch.empty(ops.conv_num_parts(N, H, H, 3) * 2 * C, device='cuda') if full else None
NB = 256
dbg = torch.zeros(NB * 8 + NB * 8 * 13 * 3, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.mpn_diag_set_conv_stamps.argtypes = [ctypes.c_void_p]
for _ in range(5):
    ops.conv_fwd(x, pc.fwd, C, 3, aff, out=y, stats_part=part)
lib.mpn_diag_set_conv_stamps(ctypes.c_void_p(dbg.data_ptr()))
ops.conv_fwd(x, pc.fwd, C, 3, aff, out=y, stats_part=part)
torch.cuda.synchronize()
lib.mpn_diag_set_conv_stamps(None)
d = dbg.cpu().numpy()
w = d[NB * 8:].reshape(NB, 8, 13, 3).astype(np.float64)
ok = w[:, :, 0, 0] > 0
w = w[ok.all(1)]
print("blocks:", len(w), "(affine + statistics)" if full else "(plain)")
body, wait, bar = [], [], []
prev_end = None
print("stage   body   wait  barrier   (mean cycles over blocks and waves; body = from the previous barrier's end to this stage's wait)")
for st in range(12):
    b = (w[:, :, st, 0] - (w[:, :, st - 1, 2] if st > 0 else w[:, :, st, 0]))
    wt = w[:, :, st, 1] - w[:, :, st, 0]
    br = w[:, :, st, 2] - w[:, :, st, 1]
    print(f"{st:5d} {b.mean():6.0f} {wt.mean():6.0f} {br.mean():8.0f}     wait by wave: " + " ".join(f"{v:5.0f}" for v in wt.mean(0)) + "   barrier by wave: " + " ".join(f"{v:5.0f}" for v in br.mean(0)))
    body.append(b.mean()); wait.append(wt.mean()); bar.append(br.mean())
ep = w[:, :, 12, 1] - w[:, :, 12, 0]
print(f"epilogue {ep.mean():.0f} cycles (by wave: " + " ".join(f"{v:5.0f}" for v in ep.mean(0)) + ")")
tile = w[:, :, 12, 1] - w[:, :, 0, 0]
print(f"sum over stages 1..11: body {sum(body[1:]):.0f}  wait {sum(wait[1:]):.0f}  barrier {sum(bar[1:]):.0f};  first stage's wait + barrier {wait[0] + bar[0]:.0f};  tile (stage 0's wait -> end of epilogue) {tile.mean():.0f}")
