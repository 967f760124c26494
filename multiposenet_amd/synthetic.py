"""Synthetic batches of the keypoint pipeline's output contract, generated on the device.

Shapes/dtypes follow detector/input_pipeline/keypoints_detector_pipeline.py:104-110:
features {'images': [b,H,W,3] f32 in [0,1]}, labels {'heatmaps': [b,H/4,W/4,17] f32 with exact 1.0
peaks, 'loss_masks', 'segmentation_masks': [b,H/4,W/4] f32 in {0,1}, 'num_boxes': [b] int32}.
Recipe: SURVEY.md 8(d) (seed = 1234 + rank; uniform heatmaps in [0,0.9) with 64 exact-1.0 peaks per
image; Bernoulli(0.95) loss mask; Bernoulli(0.3) segmentation mask; num_boxes in [1,8)).
This is data plumbing (torch RNG on the device), not part of the measured hot path.
"""
import torch


def synthetic_batch(batch_size, height=512, width=512, rank=0, device="cuda:0", peaks=64):
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    h, w = height // 4, width // 4
    images = torch.rand((batch_size, height, width, 3), generator=g, device=device, dtype=torch.float32)
    heat = torch.rand((batch_size, h, w, 17), generator=g, device=device, dtype=torch.float32) * 0.9
    idx = torch.randint(0, h * w * 17, (batch_size, peaks), generator=g, device=device)
    heat.view(batch_size, -1).scatter_(1, idx, 1.0)
    loss_masks = (torch.rand((batch_size, h, w), generator=g, device=device) < 0.95).float()
    seg = (torch.rand((batch_size, h, w), generator=g, device=device) < 0.3).float()
    num_boxes = torch.randint(1, 8, (batch_size,), generator=g, device=device, dtype=torch.int32)
    features = {"images": images}
    labels = {"heatmaps": heat, "loss_masks": loss_masks, "segmentation_masks": seg, "num_boxes": num_boxes}
    return features, labels
