"""Per-wave phase stamps of the channel-split 3x3 kernel (diagnostic build: tools/build_variant.sh diag conv_mfma.hip,conv3x3.hip,conv3x3_cs.hip
"-DMPN_DIAG", MPN_LIB=multiposenet_amd/libmpn_hip_diag.so): python tools/stamp_c3cs.py H Cin Cout [affine+stats 0/1]"""
import ctypes
import os
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
dbg = torch.zeros(256 * 256, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.mpn_diag_set_conv_stamps.argtypes = [ctypes.c_void_p]
if os.environ.get("MPN_DIAG_C3_BLOCKS"):      # fewer persistent blocks than compute units: what the clock does when part of the chip multiplies
    lib.mpn_diag_set_c3_blocks(int(os.environ["MPN_DIAG_C3_BLOCKS"]))
for _ in range(3):
    ops.conv_fwd(x, pc.fwd, Cout, 3, aff, out=y, stats_part=part)
lib.mpn_diag_set_conv_stamps(ctypes.c_void_p(dbg.data_ptr()))
ops.conv_fwd(x, pc.fwd, Cout, 3, aff, out=y, stats_part=part)
torch.cuda.synchronize()
lib.mpn_diag_set_conv_stamps(None)
d = dbg.cpu().numpy().reshape(256, 8, 32).astype(np.float64)
d = d[d[:, 0, 0] > 0]
nch = Cin // 64
# stamps (conv3x3_cs.hip): 0 loop top; 16..21 ends of the first chunk's stages; 1 + c behind chunk c's barrier; 22..27 ends of the last
# chunk's stages (nch > 1); 9 behind the image writes of the tile (in front of the last chunk's barrier)
if nch == 1:
    cols, names = [0, 21, 9, 1], ["stages+epi", "image", "barrier"]
else:
    cols = [0, 21, 1] + ([nch - 1] if nch > 2 else []) + [27, 9, nch]
    names = ["chunk0 stages+epi", "barrier"] + (["middle chunks"] if nch > 2 else []) + ["last chunk", "image", "barrier"]
ph = np.diff(d[:, :, cols], axis=2)          # [block][wave][phase]
print("blocks with a third tile:", len(d), " block lifetime us: mean %.1f" % ((d[:, 0, 13] - d[:, 0, 12]).mean() / 100))
print("phase cycles, mean over blocks, per wave (rows = waves 0..7):")
print("   " + "  ".join("%18s" % n for n in names))
for wv in range(8):
    print("w%d " % wv + "  ".join("%18.0f" % v for v in ph[:, wv, :].mean(0)))
print("tile total (wave 0): %.0f cycles; MFMA cycles per SIMD and tile: %d" % ((d[:, 0, nch] - d[:, 0, 0]).mean(), 2 * 16 * 16 * 9 * Cin // 32))
tiles_per_block = N * ((H + 15) // 16) ** 2 * (Cout // 128) / len(d)
print("blocks %d, tiles per block %.1f: clock estimate %.3f GHz (tile cycles x tiles per block / block lifetime)" % (
    len(d), tiles_per_block, (d[:, 0, nch] - d[:, 0, 0]).mean() * tiles_per_block / ((d[:, 0, 13] - d[:, 0, 12]).mean() * 10)))

st = [0] + [16 + i for i in range(6)]
print("first chunk, per stage (loop top -> end of stage 0, ... stage 5), waves 0 and 4:")
for wv in (0, 4):
    print("w%d " % wv + "  ".join("%8.0f" % v for v in np.diff(d[:, wv, st], axis=1).mean(0)))
if nch > 1:
    st = [nch - 1] + [22 + i for i in range(6)]
    print("last chunk, per stage:")
    for wv in (0, 4):
        print("w%d " % wv + "  ".join("%8.0f" % v for v in np.diff(d[:, wv, st], axis=1).mean(0)))
