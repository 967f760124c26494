"""Detector with the reference's surface (inference/detector.py:5-61) for the keypoint outputs of the frozen graph."""
import numpy as np
import torch

from ..net import KeypointNet


class Detector:
    def __init__(self, model_path, gpu_memory_fraction=0.25, visible_device_list='0', dtype=torch.bfloat16, prn_path=None,
                 max_boxes=32):
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
        # prn_path: `.npz` with PRN/fc{1,2}/{weights,biases} (the frozen graph of create_pb.py holds both models): with it,
        # person boxes handed to __call__ get their keypoints assigned (create_pb.py:86-142)
        self.assigner = None
        if prn_path is not None:
            from ..prn import PoseResidualNet
            from ..prn_inference import KeypointAssigner
            from ..checkpoint import load_npz
            prn_net = PoseResidualNet(batch=max_boxes, dtype=dtype, device=device)
            load_npz(prn_path, prn_net, with_optimizer=False)
            self.assigner = KeypointAssigner(prn_net)

    def __call__(self, image, score_threshold=0.05, boxes=None, scores=None):
        """
        Arguments:
            image: a numpy uint8 array with shape [height, width, 3], that represents a RGB image.
            score_threshold: a float number.
        Returns the reference's dict; only the keypoint-path entries are computed:
            'keypoint_heatmaps' [h/4, w/4, 17] = sigmoid(logits[..., :17]), 'segmentation_masks' [h/4, w/4]
            (create_pb.py:73-76). Person boxes come from the caller (`boxes`, `scores`: the RetinaNet head is out of scope); with them
            and `prn_path` the PRN outputs 'keypoint_scores' [n,17] / 'keypoint_positions' [n,17,2] are computed, otherwise empty.
        """
        h, w, _ = image.shape
        assert h % 128 == 0 and w % 128 == 0                      # inference/detector.py:45
        if image.dtype != np.uint8:
            raise ValueError("image must be uint8")
        x = torch.from_numpy(np.ascontiguousarray(image[None])).to(self.net.device)
        heat, seg = self.net.predict(x)                            # uint8 -> /255 -> 2x-1 fused into the stem conv
        if boxes is not None and self.assigner is not None and len(boxes):
            # boxes [n,4] normalised (ymin, xmin, ymax, xmax) from a person detector (the reference's RetinaNet head is not
            # part of this build), scores [n] or None; same filtering as inference/detector.py:55-60
            boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
            scores = np.ones(len(boxes), np.float32) if scores is None else np.asarray(scores, np.float32)
            keep = scores > score_threshold
            boxes, scores = boxes[keep], scores[keep]
            kscore = np.zeros([0, 17], np.float32)
            kpos = np.zeros([0, 17, 2], np.float32)
            if len(boxes):
                db = torch.from_numpy(boxes[None]).to(self.net.device)
                ks, kp = self.assigner(heat.contiguous(), db, torch.tensor([len(boxes)], device=self.net.device), compact=True)
                kscore, kpos = ks.cpu().numpy(), kp.cpu().numpy()
            return {
                'keypoint_heatmaps': heat[0].cpu().numpy(), 'segmentation_masks': seg[0].cpu().numpy(),
                'boxes': boxes, 'scores': scores, 'num_boxes': np.int32(len(boxes)),
                'keypoint_scores': kscore, 'keypoint_positions': kpos,
            }
        return {
            'keypoint_heatmaps': heat[0].cpu().numpy(), 'segmentation_masks': seg[0].cpu().numpy(),
            'boxes': np.zeros([0, 4], np.float32), 'scores': np.zeros([0], np.float32), 'num_boxes': np.int32(0),
            'keypoint_scores': np.zeros([0], np.float32), 'keypoint_positions': np.zeros([0, 17, 2], np.float32),
        }
