"""One leg's step, replayed a few times, for a kernel trace: rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o run --
python3 tools/leg_step.py detector|prn [replays]. detector = BASELINE config 4 (RetinaNet head train step, batch 16 @ 896 x 1408,
bf16), prn = config 5 (PRN train step, 128 crops, fp16). The trace's last `replays` graph launches are the steps."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))

leg = sys.argv[1]
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if leg == "detector":
    from multiposenet_amd.retinanet import PersonDetectorNet
    batch, height, width = 16, 896, 1408
    net = PersonDetectorNet(dtype=torch.bfloat16, seed=0)
    g_ = torch.Generator(device="cuda"); g_.manual_seed(4321)
    images = torch.rand((batch, height, width, 3), generator=g_, device="cuda")
    rs = np.random.RandomState(7)
    maxn = 12
    boxes = np.zeros((batch, maxn, 4), np.float32)
    for b in range(batch):
        for n in range(maxn):
            cy, cx = rs.rand(2); h, w = 0.08 + 0.5 * rs.rand(2)
            boxes[b, n] = [max(cy - h / 2, 0), max(cx - w / 2, 0), min(cy + h / 2, 1), min(cx + w / 2, 1)]
    gt = {"boxes": torch.from_numpy(boxes).cuda(), "num_boxes": torch.from_numpy(rs.randint(1, maxn + 1, batch).astype(np.int32)).cuda()}
    hp = {"initial_learning_rate": 1e-3, "num_steps": 150000, "weight_decay": 5e-5, "localization_loss_weight": 1.0,
          "classification_loss_weight": 2.0, "gamma": 2.0, "alpha": 0.25}
    step = lambda: net.train_step(images, gt, hp)
elif leg == "prn":
    from multiposenet_amd.prn import PoseResidualNet
    B = 128
    net = PoseResidualNet(batch=B, dtype=torch.float16, seed=0)
    rs = np.random.RandomState(3)
    x = torch.tensor(rs.rand(B, 56, 36, 17).astype(np.float32)).cuda()
    y = torch.zeros(B, 56, 36, 17)
    for b in range(B):
        for k in range(17):
            y[b, rs.randint(56), rs.randint(36), k] = 1.0
    y = y.cuda()
    step = lambda: net.train_step(x, y, 1e-3, 200000)
else:
    raise SystemExit("usage: leg_step.py detector|prn [replays]")
for _ in range(2):
    step()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(replays):
    graph.replay()
e1.record()
torch.cuda.synchronize()
print(f"{leg}: {e0.elapsed_time(e1) / replays:.3f} ms per step over {replays} graph replays")
