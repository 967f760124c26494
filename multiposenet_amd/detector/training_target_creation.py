"""get_training_targets with the reference's signature (detector/training_target_creation.py:5-42) on the HIP kernel
mpn_retina_match (matching :45-123, create_targets :126-159, encode box_utils.py:78-110)."""
import numpy as np
import torch

from .. import _lib


def get_training_targets(anchors, groundtruth_boxes, positives_threshold=0.5, negatives_threshold=0.4):
    """
    Arguments:
        anchors: a float array / tensor with shape [num_anchors, 4].
        groundtruth_boxes: a float array / tensor with shape [N, 4].
    Returns:
        regression_targets: a float32 CUDA tensor [num_anchors, 4]; matches: an int32 CUDA tensor [num_anchors]
        (-1 background, -2 ignore, else the index of the matched box).
    """
    def dev(t):
        t = torch.as_tensor(np.asarray(t, dtype=np.float32)) if not torch.is_tensor(t) else t.float()
        return t.cuda().contiguous()
    an, gt = dev(anchors), dev(groundtruth_boxes).reshape(-1, 4)
    A, n = an.shape[0], gt.shape[0]
    maxn = max(n, 1)
    boxes = torch.zeros((1, maxn, 4), dtype=torch.float32, device=an.device)
    boxes[0, :n] = gt
    nb = torch.tensor([n], dtype=torch.int32, device=an.device)
    matches = torch.empty((1, A), dtype=torch.int32, device=an.device)
    targets = torch.empty((1, A, 4), dtype=torch.float32, device=an.device)
    nm = torch.zeros(1, dtype=torch.int32, device=an.device)
    ws = torch.empty(_lib.lib().mpn_retina_match_workspace_bytes(1, maxn), dtype=torch.uint8, device=an.device)
    _lib.call("mpn_retina_match", _lib.ptr(an), _lib.ptr(boxes), _lib.ptr(nb), 1, A, maxn, float(positives_threshold),
              float(negatives_threshold), _lib.ptr(matches), _lib.ptr(targets), _lib.ptr(nm), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    return targets[0], matches[0]
