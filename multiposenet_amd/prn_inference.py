"""Keypoint assignment at inference: heatmaps + person boxes -> PRN -> per-person keypoints (create_pb.py:86-142).

The reference builds this inline in its frozen-graph export (`create_pb.py`): normalise the sigmoid heatmaps per image and
channel, `tf.image.crop_and_resize` every detected box to 56x36, run `prn`, softmax over positions, `argmax_2d`.
Here it is three HIP launches around `PoseResidualNet.predict` (include/mpn.h: mpn_heatmap_minmax, mpn_prn_crop,
mpn_prn_decode); there is no CPU fallback.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr
from .prn import CROP_SIZE, NUM_KEYPOINTS


@_lib.device_guarded("crops", "__call__")
class KeypointAssigner:
    """scores, positions = KeypointAssigner(prn_net)(heatmaps, boxes, num_boxes)

    heatmaps: f32 [b, h, w, 17] sigmoid heatmaps (create_pb.py:73); boxes: f32 [b, max_boxes, 4] normalised
    (ymin, xmin, ymax, xmax); num_boxes: int [b]. Returns `keypoint_scores` f32 [n, 17] and `keypoint_positions`
    f32 [n, 17, 2] with n = b * max_boxes rows in (image, slot) order - rows of slots >= num_boxes[i] are those of an
    all-zero crop; `compact=True` drops them (one device-to-host copy of num_boxes), giving the reference's
    [sum(num_boxes), ...] outputs."""

    def __init__(self, prn_net, threshold=0.2):
        self.net = prn_net
        # (prn_net may be None when only crops()/decode() are used: launches then go to the current device)
        self.device = prn_net.device if prn_net is not None else torch.device("cuda", torch.cuda.current_device())
        self.threshold = float(threshold)
        self._keys = None

    def crops(self, heatmaps, boxes, box_ind):
        """create_pb.py:90-109. boxes f32 [n,4], box_ind int32 [n] (entries outside [0,b) give zero crops) -> f32 [n,56,36,17]."""
        b, h, w, c = heatmaps.shape
        if heatmaps.dtype != torch.float32 or not heatmaps.is_contiguous() or c != NUM_KEYPOINTS:
            raise ValueError("heatmaps must be contiguous float32 [b,h,w,17]")
        n = boxes.shape[0]
        if self._keys is None or self._keys.numel() < b * c * 2:
            self._keys = torch.empty(b * c * 2, dtype=torch.int32, device=heatmaps.device)
        call("mpn_heatmap_minmax", ptr(heatmaps), b, h, w, c, ptr(self._keys), stream_ptr())
        out = torch.empty((n, CROP_SIZE[0], CROP_SIZE[1], c), dtype=torch.float32, device=heatmaps.device)
        call("mpn_prn_crop", ptr(heatmaps), ptr(self._keys), ptr(boxes), ptr(box_ind), n, b, h, w, c, CROP_SIZE[0], CROP_SIZE[1],
             self.threshold, ptr(out), stream_ptr())
        return out

    @staticmethod
    def decode(logits):
        """create_pb.py:114-138. logits f32 [n,56,36,17] -> (scores [n,17], positions [n,17,2])."""
        n, h, w, c = logits.shape
        logits = logits.contiguous()
        scores = torch.empty((n, c), dtype=torch.float32, device=logits.device)
        pos = torch.empty((n, c, 2), dtype=torch.float32, device=logits.device)
        call("mpn_prn_decode", ptr(logits), n, h, w, c, ptr(scores), ptr(pos), stream_ptr())
        return scores, pos

    def crops_of_slots(self, heatmaps, boxes, num_boxes, slot0, n):
        """Crops of slots slot0 .. slot0 + n - 1 of a detector's padded output (boxes [b,max,4], num_boxes int32 [b]); padding slots
        and slots past the array are zero crops (mpn_prn_crop_slots: create_pb.py:96-104 without slices and concat)."""
        b, h, w, c = heatmaps.shape
        if heatmaps.dtype != torch.float32 or not heatmaps.is_contiguous() or c != NUM_KEYPOINTS:
            raise ValueError("heatmaps must be contiguous float32 [b,h,w,17]")
        if self._keys is None or self._keys.numel() < b * c * 2:
            self._keys = torch.empty(b * c * 2, dtype=torch.int32, device=heatmaps.device)
        call("mpn_heatmap_minmax", ptr(heatmaps), b, h, w, c, ptr(self._keys), stream_ptr())
        out = torch.empty((n, CROP_SIZE[0], CROP_SIZE[1], c), dtype=torch.float32, device=heatmaps.device)
        call("mpn_prn_crop_slots", ptr(heatmaps), ptr(self._keys), ptr(boxes), ptr(num_boxes), int(slot0), int(n), boxes.shape[1], b, h, w, c,
             CROP_SIZE[0], CROP_SIZE[1], self.threshold, ptr(out), stream_ptr())
        return out

    def __call__(self, heatmaps, boxes, num_boxes, compact=False):
        b, max_boxes = boxes.shape[0], boxes.shape[1]
        dev = heatmaps.device
        nb = torch.as_tensor(num_boxes, device=dev).to(torch.int32).contiguous()
        flat = boxes.to(torch.float32).contiguous()
        n, B = b * max_boxes, self.net.valid
        scores, positions = [], []
        for s in range(0, n, B):   # the network is built for a fixed batch: the last chunk runs past the array (zero crops)
            k = min(B, n - s)
            sc, po = self.decode(self.net.predict(self.crops_of_slots(heatmaps, flat, nb, s, B)))
            scores.append(sc[:k])
            positions.append(po[:k])
        scores, positions = torch.cat(scores), torch.cat(positions)     # (data movement only)
        if compact:   # the reference's [sum(num_boxes), ...] rows: live slots in (image, slot) order - an index gather
            slot = torch.arange(max_boxes, device=dev, dtype=torch.int32).view(1, max_boxes)
            keep = (slot < nb.view(b, 1)).reshape(-1).nonzero().view(-1)
            scores, positions = scores[keep], positions[keep]
        return scores, positions
