"""Target-heatmap rendering - same surface as the reference's `detector/input_pipeline/heatmap_creation.py`.

`get_heatmaps(keypoints, boxes, width, height, downsample)` keeps the signature and the (bit-identical) result of
heatmap_creation.py:6-72; the blobs are rendered by the HIP kernel `mpn_heatmap_render`.  `HeatmapRenderer` /
`get_heatmaps_batch` are the batched device-resident form a training loop uses instead of one GIL-bound
`tf.py_func` call per image (keypoints_detector_pipeline.py:86-90): one launch per batch, labels never leave HBM.
"""
import math

import numpy as np

from ... import _lib

NUM_KEYPOINTS = 17  # detector/constants.py:10


@_lib.device_guarded("__call__")
class HeatmapRenderer:
    """Owns the output + scratch buffers for (batch, image size, downsample); persons per batch may vary."""

    def __init__(self, batch, width, height, downsample, device="cuda:0", max_persons=256):
        import torch
        self.B, self.width, self.height, self.downsample = int(batch), int(width), int(height), int(downsample)
        if self.width < 2 or self.height < 2 or self.downsample < 1:
            raise ValueError("width, height must be >= 2 and downsample >= 1")
        self.h = math.ceil(self.height / self.downsample)          # heatmap_creation.py:27-28
        self.w = math.ceil(self.width / self.downsample)
        self.device = torch.device(device)
        self.out = torch.empty((self.B, self.h, self.w, NUM_KEYPOINTS), dtype=torch.float32, device=self.device)
        self._reserve(max_persons)

    def _reserve(self, persons):
        import torch
        self.max_persons = int(persons)
        nbytes = _lib.lib().mpn_heatmap_render_workspace_bytes(self.max_persons)
        self.workspace = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)

    def __call__(self, keypoints, boxes, first_person, out=None):
        """keypoints int32 [P,17,3] (y,x,vis), boxes f32 [P,4], first_person int32 [B+1] - device tensors.
        Returns the float32 [B,h,w,17] heatmaps (the renderer's own buffer unless `out` is given)."""
        import torch
        P = int(keypoints.shape[0])
        if tuple(keypoints.shape) != (P, NUM_KEYPOINTS, 3) or keypoints.dtype != torch.int32:
            raise ValueError(f"keypoints must be int32 [P,17,3], got {keypoints.dtype} {tuple(keypoints.shape)}")
        if tuple(boxes.shape) != (P, 4) or boxes.dtype != torch.float32:
            raise ValueError(f"boxes must be float32 [P,4], got {boxes.dtype} {tuple(boxes.shape)}")
        if tuple(first_person.shape) != (self.B + 1,) or first_person.dtype != torch.int32:
            raise ValueError(f"first_person must be int32 [{self.B + 1}]")
        if not (keypoints.is_contiguous() and boxes.is_contiguous() and first_person.is_contiguous()):
            raise ValueError("inputs must be contiguous")
        out = self.out if out is None else out
        if tuple(out.shape) != tuple(self.out.shape) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError(f"out must be contiguous float32 {tuple(self.out.shape)}")
        if P > self.max_persons:
            self._reserve(max(P, 2 * self.max_persons))
        _lib.call("mpn_heatmap_render", _lib.ptr(keypoints), _lib.ptr(boxes), _lib.ptr(first_person), self.B, P,
                  self.width, self.height, self.downsample, _lib.ptr(out), _lib.ptr(self.workspace),
                  self.workspace.numel(), _lib.stream_ptr())
        return out


_renderers = {}


def _renderer(batch, width, height, downsample, device):
    key = (batch, width, height, downsample, str(device))
    if key not in _renderers:
        _renderers[key] = HeatmapRenderer(batch, width, height, downsample, device)
    return _renderers[key]


def _check_people(keypoints, boxes, width, height):
    keypoints, boxes = np.asarray(keypoints), np.asarray(boxes)
    if keypoints.ndim != 3 or keypoints.shape[1:] != (NUM_KEYPOINTS, 3):
        raise ValueError(f"keypoints must have shape [num_persons, 17, 3], got {keypoints.shape}")
    if boxes.shape != (keypoints.shape[0], 4):
        raise ValueError(f"boxes must have shape [num_persons, 4], got {boxes.shape}")
    if not np.issubdtype(keypoints.dtype, np.integer):
        raise ValueError("keypoints must be an integer array (heatmap_creation.py:9)")
    vis = keypoints[:, :, 2] > 0
    y, x = keypoints[:, :, 0][vis], keypoints[:, :, 1][vis]
    if y.size and (y.min() < 0 or y.max() > height - 1 or x.min() < 0 or x.max() > width - 1):
        raise ValueError("visible keypoints must lie in [0, height-1] x [0, width-1] (heatmap_creation.py:11-12)")
    return keypoints.astype(np.int32), boxes.astype(np.float32)


def get_heatmaps_batch(people, width, height, downsample, device="cuda:0"):
    """people: list of (keypoints [P_b,17,3] int, boxes [P_b,4] float32) numpy pairs, one per image (all images
    width x height).  Returns a float32 [B,h,w,17] device tensor."""
    import torch
    checked = [_check_people(k, b, width, height) for k, b in people]
    counts = np.array([0] + [k.shape[0] for k, _ in checked], np.int64)
    first = torch.from_numpy(np.cumsum(counts).astype(np.int32)).to(device)
    kp = torch.from_numpy(np.concatenate([k for k, _ in checked] + [np.zeros((0, NUM_KEYPOINTS, 3), np.int32)])).to(device)
    bx = torch.from_numpy(np.concatenate([b for _, b in checked] + [np.zeros((0, 4), np.float32)])).to(device)
    return _renderer(len(checked), int(width), int(height), int(downsample), device)(kp, bx, first)


def get_heatmaps(keypoints, boxes, width, height, downsample):
    """
    Drop-in for detector/input_pipeline/heatmap_creation.py:6 `get_heatmaps`.

    Arguments:
        keypoints: a numpy int array with shape [num_persons, 17, 3], in format (y, x, visibility).
        boxes: a numpy float array with shape [num_persons, 4], absolute (ymin, xmin, ymax, xmax).
        width, height: integers, size of the original image.
        downsample: an integer.
    Returns:
        a numpy float array with shape [height/downsample, width/downsample, 17].
    """
    out = get_heatmaps_batch([(keypoints, boxes)], int(width), int(height), int(downsample))
    return out[0].cpu().numpy()
