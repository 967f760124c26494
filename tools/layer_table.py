"""Per-layer efficiency map of the dense convs (fwd / dgrad / wgrad+reduce) at the bench shape (bs32, 512x512, bf16).
usage: python tools/layer_table.py [filter]   -> time, floor = max(FLOP/2.5PF, alg bytes/6TB/s), excess * count"""
import sys
import torch
sys.path.insert(0, '.')
from multiposenet_amd import ops

dt = torch.bfloat16
N = 32
LAYERS = [  # (name, H, Cin, Cout, k, count)
    ("pw1", 256, 32, 64, 1, 1), ("pw2", 128, 64, 128, 1, 1), ("pw3", 128, 128, 128, 1, 1), ("pw4", 64, 128, 256, 1, 1),
    ("pw5", 64, 256, 256, 1, 1), ("pw6", 32, 256, 512, 1, 1), ("pw7-11", 32, 512, 512, 1, 5), ("pw12", 16, 512, 1024, 1, 1),
    ("pw13", 16, 1024, 1024, 1, 1), ("lat5", 16, 1024, 128, 1, 1), ("lat4", 32, 512, 128, 1, 1), ("lat3", 64, 256, 128, 1, 1),
    ("lat2", 128, 128, 128, 1, 1), ("c3@16", 16, 128, 128, 3, 3), ("c3@32", 32, 128, 128, 3, 3), ("c3@64", 64, 128, 128, 3, 3),
    ("c3@128", 128, 128, 128, 3, 3), ("final", 128, 512, 64, 3, 1),
]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay(); g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * n)


flt = sys.argv[1] if len(sys.argv) > 1 else ""
tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
print(f"{'layer':8s} {'pass':6s} {'us':>8s} {'floor':>7s} {'TF/s':>7s} {'GB/s':>7s} {'excess*cnt':>10s}")
for name, H, Cin, Cout, k, cnt in LAYERS:
    if flt and flt not in name:
        continue
    x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
    dy = torch.randn(N, H, H, Cout, device='cuda').to(dt)
    w = torch.randn(k, k, Cin, Cout, device='cuda') * 0.05
    pc = ops.PackedConv(w, dt)
    sc = torch.rand(Cin, device='cuda') + 0.5
    sh = torch.randn(Cin, device='cuda') * 0.1
    y = torch.empty(N, H, H, Cout, device='cuda', dtype=dt)
    dx = torch.empty(N, H, H, Cin, device='cuda', dtype=dt)
    part = torch.empty(ops.conv_num_parts(N, H, H, k) * 2 * Cout, device='cuda')
    dw = torch.empty(k, k, Cin, Cout, device='cuda')
    npart = ops.conv_wgrad_num_parts(N, H, H, Cin, Cout, k, dt)
    wp = torch.empty(npart * dw.numel(), device='cuda')
    fl = 2.0 * N * H * H * Cin * Cout * k * k
    byt = (x.numel() + y.numel()) * 2
    runs = [("fwd", lambda: ops.conv_fwd(x, pc.fwd, Cout, k, ops.Affine(sc, sh, 1), out=y, stats_part=part), byt),
            ("dgrad", lambda: ops.conv_fwd(dy, pc.bwd, Cin, k, None, out=dx), byt),
            ("wgrad", lambda: ops.conv_bwd_weight(x, dy, k, ops.Affine(sc, sh, 1), dw, wp), byt)]
    for pname, fn, b in runs:
        us = timeit(fn)
        floor = max(fl / 2.5e9, b / 6.0e6)
        tot[pname][0] += us * cnt
        tot[pname][1] += floor * cnt
        print(f"{name:8s} {pname:6s} {us:8.1f} {floor:7.1f} {fl / us / 1e6:7.1f} {b / us / 1e3:7.1f} {(us - floor) * cnt:10.1f}"
              + (f"   nsplit {npart}" if pname == "wgrad" else ""))
for k_, (a, b) in tot.items():
    print(f"TOTAL {k_:6s} {a:8.1f} us   floor {b:8.1f} us")
