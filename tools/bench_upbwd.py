"""mpn_bilinear_up_bwd on the subnet's three upsampled levels: reading a 128-channel slice of the 512-channel concat gradient
against the same kernel on a dense 128-channel tensor; cold caches (rotating buffers)."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

N, H, C = 32, 128, 128
reps = 12


def timed(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


wide = [torch.randn(N, H, H, 4 * C, device="cuda").bfloat16() for _ in range(3)]
dense = [torch.randn(N, H, H, C, device="cuda").bfloat16() for _ in range(8)]
for lvl, u in ((1, 2), (2, 4), (3, 8)):
    out = torch.empty(N, H // u, H // u, C, device="cuda", dtype=torch.bfloat16)
    ts = timed(lambda i: ops.bilinear_up_bwd(wide[i % 3], u, lvl * C, C, out=out))
    td = timed(lambda i: ops.bilinear_up_bwd(dense[i % 8], u, 0, C, out=out))
    mb = N * H * H * C * 2 / 1e6
    print(f"upsample {u}: slice of 512 channels {ts:6.1f} us ({mb / ts:5.2f} TB/s)   dense {td:6.1f} us ({mb / td:5.2f} TB/s)", flush=True)
