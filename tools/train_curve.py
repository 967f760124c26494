"""Loss curve of N training steps on ONE fixed synthetic batch (overfitting it): bf16 throughput build next to the f32
build, same seeds - a long-run sanity check of the whole step (python tools/train_curve.py [steps] [batch] [size])."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd.net import KeypointNet
from multiposenet_amd.train import Trainer
from multiposenet_amd.synthetic import synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
hp = {"initial_learning_rate": 1e-3, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
for dt in (torch.float32, torch.bfloat16):
    net = KeypointNet(dtype=dt, seed=0)
    tr = Trainer(net, hp, use_graph=True)
    feats, labels = synthetic_batch(B, S, S, rank=0, device="cuda:0")
    feats, labels = tr.input_buffers(feats, labels)
    curve = []
    for i in range(steps):
        l = tr.step(feats, labels)
        if i % (steps // 10) == 0 or i == steps - 1:
            curve.append((i, round(float(l[6]), 4)))
    ok = all(torch.isfinite(t).all() for t in (net.theta, net.adam_m, net.adam_v, net.moving))
    print(str(dt).split(".")[1], "finite" if ok else "NON-FINITE", curve)
