"""Long replay of the keypoint train step (bs32 @ 512) from its hipGraph: finite losses and variables after thousands of steps, steady
step time (python tools/soak.py [steps])."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd.net import KeypointNet
from multiposenet_amd.train import Trainer
from multiposenet_amd.synthetic import synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
hp = {"initial_learning_rate": 3e-4, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
net = KeypointNet(dtype=torch.bfloat16, seed=0)
tr = Trainer(net, hp, use_graph=True)
feats, labels = tr.input_buffers(*synthetic_batch(32, 512, 512, rank=0, device="cuda:0"))
for i in range(10):
    tr.step(feats, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
marks = []
for i in range(steps):
    l = tr.step(feats, labels)
    if (i + 1) % (steps // 4) == 0:
        torch.cuda.synchronize()
        marks.append((i + 1, round(float(l[6]), 4), round((time.perf_counter() - t0) / (i + 1) * 1e3, 3)))
ok = all(bool(torch.isfinite(t).all()) for t in (net.theta, net.adam_m, net.adam_v, net.moving))
print("keypoints", "finite" if ok else "NON-FINITE", marks, flush=True)
assert ok
