"""Prototype 4-wave 3x3 kernel (tools/proto/c3w4.hip) against the shipped 8-wave kernel: parity on a small case, then time
and in-kernel cycles per stage (ideal: 96 MFMAs x 16 cycles = 1536) on the dominant shapes.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/proto/c3w4.hip -o tools/proto/libc3w4.so"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiposenet_amd import ops

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libc3w4.so"))
lib.c3w4_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.c3w4_launch.restype = ctypes.c_int


def pack(w):
    """[3(ky),3(kx),Cin,128] f32 -> [chunk][kx][ky][co][32 ci] bf16"""
    ky, kx, cin, co = w.shape
    return w.view(3, 3, cin // 32, 32, co).permute(2, 1, 0, 4, 3).contiguous().to(torch.bfloat16)


def run(x, wpk, y, blocks, stamps=None, mode=0):
    N, H, W, cin = x.shape
    rc = lib.c3w4_launch(x.data_ptr(), wpk.data_ptr(), y.data_ptr(), N, H, W, cin, blocks,
                         stamps.data_ptr() if stamps is not None else None, torch.cuda.current_stream().cuda_stream, mode)
    assert rc == 0, rc


def timed(fn, reps=20):
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


torch.manual_seed(0)
# ---- parity: [2,32,48,64] -> 128 against the shipped kernel and an f32 reference
N, H, W, C = 2, 32, 48, 64
x = torch.randn(N, H, W, C, device="cuda").bfloat16()
w = torch.randn(3, 3, C, 128, device="cuda") / (9 * C) ** 0.5
y = torch.full((N, H, W, 128), float("nan"), device="cuda", dtype=torch.bfloat16)
run(x, pack(w), y, 7)
want = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.bfloat16().float().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
ship = ops.conv_fwd(x, ops.PackedConv(w, torch.bfloat16).fwd, 128, 3)
print("parity: max |proto - f32 ref| %.4f (ref max %.2f), max |proto - shipped| %.4f" %
      (float((y.float() - want).abs().max()), float(want.abs().max()), float((y.float() - ship.float()).abs().max())), flush=True)
assert float((y.float() - want).abs().max()) < 0.03 * float(want.abs().max())

for (N, H, W, C) in ((32, 128, 128, 128), (32, 128, 128, 512), (32, 64, 64, 128)):
    nb = 3
    xs = [torch.randn(N, H, W, C, device="cuda").bfloat16() for _ in range(nb)]
    w = torch.randn(3, 3, C, 128, device="cuda") / (9 * C) ** 0.5
    wpk, pk = pack(w), ops.PackedConv(w, torch.bfloat16)
    y = torch.empty(N, H, W, 128, device="cuda", dtype=torch.bfloat16)
    flops = 2.0 * N * H * W * 9 * C * 128
    t_ship = timed(lambda i: ops.conv_fwd(xs[i % nb], pk.fwd, 128, 3, out=y))
    line = f"[{N},{H},{W},{C}] -> 128: shipped {t_ship:7.1f} us ({flops / t_ship / 1e6 / 2500:.3f} of 2.5 PF)"
    for blocks in (256,):
        stamps = torch.zeros(2 * blocks, dtype=torch.int64, device="cuda")
        t = timed(lambda i: run(xs[i % nb], wpk, y, blocks))
        run(xs[0], wpk, y, blocks, stamps)
        torch.cuda.synchronize()
        st = stamps.view(blocks, 2).double()
        cps = float((st[:, 0] / st[:, 1]).mean())
        line += f"   prototype ({blocks} blocks) {t:7.1f} us ({flops / t / 1e6 / 2500:.3f}), {cps:.0f} counter ticks per stage"
    print(line, flush=True)
    for mode, what in ((1, "no global loads / LDS commits"), (3, "+ no fragment reads"), (7, "+ no barrier"), (4, "all work, no barrier (wrong results)")):
        stamps = torch.zeros(2 * 256, dtype=torch.int64, device="cuda")
        t = timed(lambda i: run(xs[i % nb], wpk, y, 256, mode=mode))
        run(xs[0], wpk, y, 256, stamps, mode)
        torch.cuda.synchronize()
        st = stamps.view(256, 2).double()
        print(f"      knock-out {mode} ({what}): {t:7.1f} us, {float((st[:, 0] / st[:, 1]).mean()):.0f} ticks per stage", flush=True)
