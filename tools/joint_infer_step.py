"""The joint inference graph (Detector) on one 640 x 640 image, a few calls, for a kernel trace:
rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o run -- python3 tools/joint_infer_step.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from multiposenet_amd.inference import Detector
from multiposenet_amd.prn import initial_values as prn_values
from multiposenet_amd.retinanet import initial_head_values
head = initial_head_values(0)
head["class_net/logits/kernel"] = (np.random.RandomState(8).randn(3, 3, 64, 6) * 0.4).astype(np.float32)
head["class_net/logits/bias"] = np.full(6, -2.0, np.float32)
det = Detector(None, dtype=torch.bfloat16, detector_path=head, prn_path=prn_values(seed=0))
image = np.random.RandomState(0).randint(0, 256, (640, 640, 3)).astype(np.uint8)
for _ in range(5):
    out = det(image, score_threshold=0.0)
torch.cuda.synchronize()
print("persons", int(out["num_boxes"]))
