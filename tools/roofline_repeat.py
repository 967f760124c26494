"""How the dominant-kernel figure of bench.py moves with the state of the chip: python tools/roofline_repeat.py
(25 training steps as in the default bench, then the roofline leg five times in a row)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd.net import KeypointNet
from multiposenet_amd.train import Trainer
from multiposenet_amd.synthetic import synthetic_batch
from bench_legs import dominant_kernel_roofline

net = KeypointNet(dtype=torch.bfloat16, device="cuda:0", seed=0)
tr = Trainer(net, {"initial_learning_rate": 3e-4, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0})
feats, labels = synthetic_batch(32, 512, 512, rank=0, device="cuda:0")
feats, labels = tr.input_buffers(feats, labels)
for _ in range(25):
    tr.step(feats, labels)
torch.cuda.synchronize()
for rep in range(5):
    r = dominant_kernel_roofline(net, 32, 512, torch.bfloat16)
    print(rep, r["launch_us"], r["frac"])
for iters in (50, 200):
    r = dominant_kernel_roofline(net, 32, 512, torch.bfloat16, iters=iters)
    print("iters", iters, r["launch_us"], r["frac"])
