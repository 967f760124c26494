"""Oracle for the heatmap peak decode (TEST INFRASTRUCTURE, not shipped code).

CPU/numpy restatement of `get_keypoints` (reference inference/utils.py:29-52) and of
`argmax_2d` + scores (reference create_pb.py:120-142). Pinned: bit-identical to the
outputs of the imported reference on tests/golden/decode_goldens.npz.
"""
import numpy as np

NUM_KEYPOINTS = 17  # detector/constants.py:10


def get_keypoints(heatmaps, box, threshold):
    """inference/utils.py:29-52. heatmaps [h,w,17] float, box [4], threshold -> int32 [17,3]."""
    keypoints = np.zeros([NUM_KEYPOINTS, 3], dtype='int32')          # utils.py:38
    ymin, xmin, ymax, xmax = box                                       # utils.py:40
    height, width = ymax - ymin, xmax - xmin                           # utils.py:41
    h, w, _ = heatmaps.shape                                           # utils.py:42
    flat = heatmaps.reshape(h * w, -1)
    maxima = flat.max(axis=0)          # NaN propagates like ndarray.max()   (utils.py:46)
    argmaxima = flat.argmax(axis=0)    # first occurrence in row-major order  (utils.py:47)
    for j in range(NUM_KEYPOINTS):
        if maxima[j] > threshold:                                      # strict (utils.py:46)
            y, x = np.unravel_index(argmaxima[j], (h, w))
            y = np.clip(int(y * height / h), 0, height)                # utils.py:48
            x = np.clip(int(x * width / w), 0, width)                  # utils.py:49
            keypoints[j] = np.array([x, y, 1])                         # utils.py:50
    return keypoints


def get_keypoints_batch(heatmaps, boxes, threshold):
    return np.stack([get_keypoints(hm, b, threshold) for hm, b in zip(heatmaps, boxes)])


def scores_and_indices(heatmaps):
    """Per-channel max and first-occurrence flat argmax, [B,h,w,C] -> ([B,C] max, [B,C] idx)
    (create_pb.py:128-131,138: tf.argmax picks the smallest index on ties)."""
    B, h, w, C = heatmaps.shape
    flat = heatmaps.reshape(B, h * w, C)
    return flat.max(axis=1), flat.argmax(axis=1).astype(np.int32)
