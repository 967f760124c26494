"""Detector with the reference's surface (inference/detector.py:5-61) for the keypoint outputs of the frozen graph."""
import numpy as np
import torch

from ..net import KeypointNet


class Detector:
    def __init__(self, model_path, gpu_memory_fraction=0.25, visible_device_list='0', dtype=torch.bfloat16):
        """
        Arguments:
            model_path: path to a build-native weight file (.npz, keys = the reference's variable names, HWIO kernels:
                what `KeypointNet.state_dict()` saves / a TF checkpoint exported elsewhere), or None for seeded random
                weights. (The reference loads a frozen .pb, inference/detector.py:13-19.)
            gpu_memory_fraction: accepted for signature compatibility, unused (buffers are sized per input shape).
            visible_device_list: a string, the GPU index.
        """
        device = f"cuda:{int(str(visible_device_list).split(',')[0])}"
        values = None
        if model_path is not None:
            with np.load(model_path) as z:
                values = {k: z[k] for k in z.files}
        self.net = KeypointNet(values=values, dtype=dtype, device=device)

    def __call__(self, image, score_threshold=0.05):
        """
        Arguments:
            image: a numpy uint8 array with shape [height, width, 3], that represents a RGB image.
            score_threshold: a float number.
        Returns the reference's dict; only the keypoint-path entries are computed:
            'keypoint_heatmaps' [h/4, w/4, 17] = sigmoid(logits[..., :17]), 'segmentation_masks' [h/4, w/4]
            (create_pb.py:73-76). Person boxes and PRN outputs (RetinaNet / PRN, out of scope) come back empty.
        """
        h, w, _ = image.shape
        assert h % 128 == 0 and w % 128 == 0                      # inference/detector.py:45
        if image.dtype != np.uint8:
            raise ValueError("image must be uint8")
        x = torch.from_numpy(np.ascontiguousarray(image[None])).to(self.net.device)
        heat, seg = self.net.predict(x)                            # uint8 -> /255 -> 2x-1 fused into the stem conv
        return {
            'keypoint_heatmaps': heat[0].cpu().numpy(), 'segmentation_masks': seg[0].cpu().numpy(),
            'boxes': np.zeros([0, 4], np.float32), 'scores': np.zeros([0], np.float32), 'num_boxes': np.int32(0),
            'keypoint_scores': np.zeros([0], np.float32), 'keypoint_positions': np.zeros([0, 17, 2], np.float32),
        }
