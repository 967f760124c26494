"""How fast is the main loop of the 1x1 GEMM kernel (pointwise.hip) when the problem is large enough that prologue, epilogue and
the single-wave grid stop mattering?  M = 32*H*H pixels, K = Cin, N = Cout:  python tools/bench_gemm_core.py"""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd import ops

dt = torch.bfloat16
st = torch.cuda.current_stream()


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


for (H, Cin, Cout) in [(32, 512, 512), (32, 2048, 512), (64, 2048, 512), (64, 2048, 2048), (64, 1024, 1024), (128, 512, 512)]:
    B = 32
    x = (torch.rand(B, H, H, Cin, device="cuda") * 2 - 1).to(dt)
    pc = ops.PackedConv((torch.rand(1, 1, Cin, Cout, device="cuda") * 2 - 1) * 0.05, dt)
    y = torch.empty(B, H, H, Cout, device="cuda", dtype=dt)
    fl = 2.0 * B * H * H * Cin * Cout
    t = timed(lambda: ops.conv_fwd(x, pc.fwd, Cout, 1, None, out=y))
    print(f"M={B * H * H:7d} K={Cin:5d} N={Cout:5d}  {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF ({fl / t / 2.5e15:.3f})", flush=True)
