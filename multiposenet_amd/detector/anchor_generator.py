"""AnchorGenerator with the reference's constructor and call contract (detector/anchor_generator.py:12-116)."""
import itertools

import numpy as np


class AnchorGenerator:
    def __init__(self, strides=[8, 16, 32, 64, 128], scales=[32, 64, 128, 256, 512], scale_multipliers=[1.0, 1.4142],
                 aspect_ratios=[1.0, 2.0, 0.5]):
        assert len(strides) == len(scales)
        self.strides, self.scales = strides, scales
        self.scale_multipliers, self.aspect_ratios = scale_multipliers, aspect_ratios
        self.num_anchors_per_location = len(aspect_ratios) * len(scale_multipliers)

    def __call__(self, image_height, image_width):
        """Returns a float32 numpy array [num_anchors, 4], boxes (ymin, xmin, ymax, xmax) with normalised coordinates, in the
        order of reshape_and_concatenate (level, y, x, anchor). float32 arithmetic step by step like the TF graph
        (anchor_generator.py:53-116, tile_anchors :119-166); also sets num_anchors_per_feature_map and raw_anchors."""
        f = np.float32
        ih, iw = f(image_height), f(image_width)
        pairs = list(itertools.product(self.scale_multipliers, self.aspect_ratios))
        ratios = np.array([a for _, a in pairs], dtype=f)
        self.num_anchors_per_feature_map, self.raw_anchors = [], []
        for i, stride in enumerate(self.strides):
            h, w = int(np.ceil(ih / f(stride))), int(np.ceil(iw / f(stride)))
            self.num_anchors_per_feature_map.append(h * w * self.num_anchors_per_location)
            scales = np.array([m * self.scales[i] for m, _ in pairs], dtype=f)
            rs = np.sqrt(ratios)
            heights, widths = scales / rs, scales * rs
            oy = f(0.5) * (ih - (f(h) - f(1.0)) * f(stride))
            ox = f(0.5) * (iw - (f(w) - f(1.0)) * f(stride))
            yc = np.arange(h).astype(f) * f(stride) + oy
            xc = np.arange(w).astype(f) * f(stride) + ox
            xg, yg = np.meshgrid(xc, yc)
            centers = np.stack([yg, xg], axis=2)[:, :, None, :].repeat(len(scales), axis=2)
            sizes = np.stack([heights, widths], axis=1)[None, None].repeat(h, 0).repeat(w, 1)
            self.raw_anchors.append(np.concatenate([centers - f(0.5) * sizes, centers + f(0.5) * sizes], axis=3).reshape(-1, 4).astype(f))
        anchors = np.concatenate(self.raw_anchors, axis=0) / np.array([ih, iw, ih, iw], dtype=f)
        return anchors.astype(f)
