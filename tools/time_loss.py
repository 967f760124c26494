import sys, torch
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
from multiposenet_amd import ops
from time_misc_util import timeit
B, h = 32, 128
logits = torch.randn(B, h, h, 18, device="cuda")
lab = {"heatmaps": torch.rand(B, h, h, 17, device="cuda") * 0.9, "loss_masks": (torch.rand(B, h, h, device="cuda") < 0.95).float(),
       "segmentation_masks": (torch.rand(B, h, h, device="cuda") < 0.3).float(), "num_boxes": torch.randint(1, 8, (B,), device="cuda", dtype=torch.int32)}
ps = [torch.randn(B, h >> l, h >> l, 128, device="cuda").to(torch.bfloat16) for l in range(4)]
dl = torch.empty_like(logits)
daux = [torch.empty(B, h >> l, h >> l, device="cuda") for l in range(4)]
out = torch.zeros(8, device="cuda")
print("loss us", timeit(lambda: ops.keypoint_loss(logits, lab, ps, dl, daux, None, out)))
