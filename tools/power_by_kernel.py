"""Board power and shader clock while ONE kernel of the step runs back to back (rocm-smi sampled from a thread, ~4 samples per second,
3 s per kernel after 1 s of settling): python tools/power_by_kernel.py
Which families of the step sit at the board's power cap (1400 W) and which do not."""
import json
import os
import re
import subprocess
import sys
import threading
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd import ops

dt, N = torch.bfloat16, 32
dev = "cuda"
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            p = c = None
            for card, v in d.items():
                for k, x in v.items():
                    if 'ower' in k and 'W' in k:
                        try:
                            p = float(x)
                        except Exception:
                            pass
                    if k.startswith('sclk'):
                        m = re.search(r'(\d+)Mhz', str(x))
                        if m:
                            c = int(m.group(1))
            samples.append((time.time(), p, c))
        except Exception:
            pass
        time.sleep(0.15)


def run(name, fn, flops=None, nbytes=None, secs=4.0, batch=200):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    us = []
    while time.time() - t0 < secs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(batch):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3 / batch)
    t1 = time.time()
    s = [(p, c) for (t, p, c) in samples if t0 + 1.0 <= t <= t1 and p is not None]
    ps = sorted(p for p, c in s)
    cs = sorted(c for p, c in s if c)
    rate = ""
    if flops:
        rate = "%7.1f TFLOP/s (%.3f)" % (flops / us[-1] / 1e6, flops / us[-1] / 2.5e9)
    if nbytes:
        rate = "%7.2f TB/s   (%.3f)" % (nbytes / us[-1] / 1e6, nbytes / us[-1] / 8e6)
    print("%-44s %7.1f us %s  power median %4.0f W (n=%d)  sclk median %4d MHz" % (
        name, us[-1], rate, ps[len(ps) // 2] if ps else -1, len(ps), cs[len(cs) // 2] if cs else -1), flush=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()

# --- dense 3x3 forward (the dominant kernel) and its weight gradient
H, Cin, Cout = 128, 128, 128
x = torch.randn(N, H, H, Cin, device=dev).to(dt)
pc = ops.PackedConv(torch.randn(3, 3, Cin, Cout, device=dev) * 0.05, dt)
aff = ops.Affine(torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1, 1)
y = torch.empty(N, H, H, Cout, device=dev, dtype=dt)
part = torch.empty(ops.conv_num_parts(N, H, H, 3) * 2 * Cout, device=dev)
fl = 2.0 * N * H * H * Cin * Cout * 9
run("3x3 128->128 @128 fwd, affine + statistics", lambda: ops.conv_fwd(x, pc.fwd, Cout, 3, aff, out=y, stats_part=part), flops=fl)
xz = torch.zeros_like(x)
run("  the same on an all-zero input", lambda: ops.conv_fwd(xz, pc.fwd, Cout, 3, aff, out=y, stats_part=part), flops=fl)
dy = torch.randn(N, H, H, Cout, device=dev).to(dt)
dw = torch.empty(3, 3, Cin, Cout, device=dev)
npart = ops.conv_wgrad_num_parts(N, H, H, Cin, Cout, 3, dt)
wp = torch.empty(npart * dw.numel(), device=dev)
run("3x3 128->128 @128 weight gradient", lambda: ops.conv_bwd_weight(x, dy, 3, aff, dw, wp, reduce=False), flops=fl)
del x, y, dy, xz

# --- deep pointwise (512 -> 512 @32^2) and the thin one (64 -> 128 @128^2)
for (H, Cin, Cout) in [(32, 512, 512), (128, 64, 128)]:
    x = torch.randn(N, H, H, Cin, device=dev).to(dt)
    pc1 = ops.PackedConv(torch.randn(1, 1, Cin, Cout, device=dev) * 0.05, dt)
    aff1 = ops.Affine(torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1, 2)
    y = torch.empty(N, H, H, Cout, device=dev, dtype=dt)
    part1 = torch.empty(ops.conv_num_parts(N, H, H, 1) * 2 * Cout, device=dev)
    run("1x1 %d->%d @%d fwd, affine + statistics" % (Cin, Cout, H), lambda: ops.conv_fwd(x, pc1.fwd, Cout, 1, aff1, out=y, stats_part=part1),
        flops=2.0 * N * H * H * Cin * Cout)
    del x, y

# --- depthwise forward (128 channels @128^2, stride 1) and the batch-norm backward apply pass on the same tensor
H, C = 128, 128
x = torch.randn(N, H, H, C, device=dev).to(dt)
w = torch.randn(3, 3, C, device=dev)
affd = ops.Affine(torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, 2)
y = torch.empty(N, H, H, C, device=dev, dtype=dt)
partd = torch.empty(ops.dwconv_num_parts(N, H, H, C, 1, dt) * 2 * C, device=dev)
run("depthwise 128 @128 s1 fwd, affine + statistics", lambda: ops.dwconv_fwd(x, w, 1, affd, out=y, stats_part=partd), nbytes=2.0 * x.numel() * 2)
a = torch.randn(N, H, H, C, device=dev).to(dt)
b = torch.randn(N, H, H, C, device=dev).to(dt)
run("torch.add(a, b, out=a) on the same tensors", lambda: torch.add(a, b, out=a), nbytes=3.0 * a.numel() * 2)
stop = True
