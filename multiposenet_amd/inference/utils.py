"""Heatmap peak decode - same surface as the reference's `inference/utils.py`.

`get_keypoints(heatmaps, box, threshold)` keeps the signature and result of
inference/utils.py:29-52; the arithmetic (per-channel max, first-occurrence argmax,
threshold, scaling into the box) runs in the HIP kernel `mpn_heatmap_decode`.
`get_keypoints_batch` is the batched device-resident form the training/serving loop uses.
"""
import numpy as np

from .. import _lib

NUM_KEYPOINTS = 17  # detector/constants.py:10


def _numpy_threshold(threshold, np_dtype):
    """`mask.max() > threshold` (utils.py:46): numpy compares a float32/16 scalar with a
    Python float in the ARRAY's precision (weak scalar), with a numpy scalar/array by
    ordinary promotion. Returns the f32 value the kernel must compare against."""
    if isinstance(threshold, (float, int)) and not isinstance(threshold, np.generic):
        return float(np.asarray(threshold, dtype=np_dtype).astype(np.float32))
    t = np.asarray(threshold)
    common = np.result_type(t.dtype, np_dtype)
    if common == np.float64 and np_dtype != np.float64:
        # numpy promotes the f32/f16 max to f64 and compares exactly. For an f32 value m:
        # m > t  <=>  m > floor32(t), the largest f32 that is <= t.
        t32 = np.float32(t)
        if np.float64(t32) > np.float64(t):
            t32 = np.nextafter(t32, np.float32(-np.inf))
        return float(t32)
    return float(np.float32(t))


def _box_hw(box):
    """(height, width) exactly as utils.py:40-41 computes them (in the box's own dtype)."""
    box = np.asarray(box)
    ymin, xmin, ymax, xmax = box
    return float(ymax - ymin), float(xmax - xmin)


@_lib.device_guarded("__call__")
class KeypointDecoder:
    """Owns the (zero-initialised) decode workspace and output buffers for a batch size."""

    def __init__(self, batch, device=None):
        import torch
        self.B = int(batch)
        self.device = _lib.current_device() if device is None else torch.device(device)
        nbytes = _lib.lib().mpn_heatmap_decode_workspace_bytes(self.B)
        self.workspace = torch.zeros(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        self.xyv = torch.empty((self.B, NUM_KEYPOINTS, 3), dtype=torch.int32, device=self.device)
        self.score = torch.empty((self.B, NUM_KEYPOINTS), dtype=torch.float32, device=self.device)
        self.index = torch.empty((self.B, NUM_KEYPOINTS), dtype=torch.int32, device=self.device)

    def __call__(self, heatmaps, box_hw, threshold):
        """heatmaps: device tensor [B,h,w,17] (f32/bf16/f16, contiguous); box_hw: device f64 [B,2];
        threshold: f32 value. Returns (xyv int32 [B,17,3], score f32 [B,17], index int32 [B,17])."""
        import torch
        if heatmaps.dim() != 4 or heatmaps.shape[0] != self.B or heatmaps.shape[3] != NUM_KEYPOINTS:
            raise ValueError(f"heatmaps must be [{self.B},h,w,{NUM_KEYPOINTS}], got {tuple(heatmaps.shape)}")
        if not heatmaps.is_contiguous():
            raise ValueError("heatmaps must be contiguous NHWC")
        if box_hw.dtype != torch.float64 or tuple(box_hw.shape) != (self.B, 2):
            raise ValueError("box_hw must be float64 [B,2]")
        B, h, w, C = heatmaps.shape
        _lib.call("mpn_heatmap_decode", _lib.ptr(heatmaps), _lib.dtype_code(heatmaps.dtype), B, h, w, C,
                  _lib.ptr(box_hw), float(threshold), _lib.ptr(self.xyv), _lib.ptr(self.score),
                  _lib.ptr(self.index), _lib.ptr(self.workspace), self.workspace.numel(), _lib.stream_ptr())
        return self.xyv, self.score, self.index


_decoders = {}


def _decoder(batch, device):
    key = (batch, str(device))
    if key not in _decoders:
        _decoders[key] = KeypointDecoder(batch, device)
    return _decoders[key]


def get_keypoints_batch(heatmaps, boxes, threshold, return_scores=False):
    """Batched decode. heatmaps: [B,h,w,17] torch CUDA tensor or numpy array; boxes: [B,4]
    (ymin,xmin,ymax,xmax); returns int32 [B,17,3] (x,y,visible) (+ f32 [B,17] scores)."""
    import torch
    if isinstance(heatmaps, np.ndarray):
        np_dtype = heatmaps.dtype
        if np_dtype == np.float64:
            raise ValueError("float64 heatmaps are not supported (reference outputs are float32)")
        hm = torch.from_numpy(np.ascontiguousarray(heatmaps)).to(_lib.current_device())   # (one process per GPU: the rank's device)
    else:
        hm = heatmaps
        np_dtype = {torch.float32: np.float32, torch.float16: np.float16}.get(hm.dtype, np.float32)
    if hm.dtype == torch.bfloat16:
        thr = float(torch.tensor(float(threshold), dtype=torch.bfloat16).float()) \
            if isinstance(threshold, (float, int)) else float(np.float32(threshold))
    else:
        thr = _numpy_threshold(threshold, np_dtype)
    boxes = np.asarray(boxes)
    hw = np.array([_box_hw(b) for b in boxes], dtype=np.float64).reshape(-1, 2)
    box_hw = torch.from_numpy(hw).to(hm.device)
    dec = _decoder(hm.shape[0], hm.device)
    xyv, score, _ = dec(hm, box_hw, thr)
    # the decoder (workspace + output buffers) is cached per (batch, device): hand out copies, so that a second call with
    # the same batch size does not overwrite the first call's result. (KeypointDecoder itself is single-stream.)
    if return_scores:
        return xyv.clone(), score.clone()
    return xyv.clone()


def get_keypoints(heatmaps, box, threshold):
    """
    Drop-in for inference/utils.py:29 `get_keypoints`.

    Arguments:
        heatmaps: a numpy float array with shape [h, w, 17].
        box: a numpy array with shape [4].
        threshold: a float number.
    Returns:
        a numpy int array with shape [17, 3].
    """
    heatmaps = np.asarray(heatmaps)
    if heatmaps.ndim != 3 or heatmaps.shape[2] != NUM_KEYPOINTS:
        raise ValueError(f"heatmaps must have shape [h, w, 17], got {heatmaps.shape}")
    xyv = get_keypoints_batch(heatmaps[None], np.asarray(box)[None], threshold)
    return xyv[0].cpu().numpy()
