"""Oracle for the RetinaNet person-detector head (BASELINE config 4).

TEST INFRASTRUCTURE ONLY - never imported by the product path (multiposenet_amd).

CPU restatement (PyTorch-CPU primitives in TensorFlow-1.15 semantics, f32 or f64; numpy for the integer / box logic) of
    detector/retinanet.py:13-217, detector/box_predictor.py:6-142, detector/anchor_generator.py:12-166,
    detector/training_target_creation.py:5-159, detector/utils/box_utils.py:14-139, detector/utils/nms.py:6-61,
    detector/fpn.py:36-55 (min_level=3, add_coarse_features=True), person_detector_model.py:8-81.
Backward passes come from torch autograd. PARITY UNPINNED: the arithmetic lives in tensorflow==1.15 (absent from
/root/reference, not installable here) and the reference holds no tests / golden vectors for it; the TF op semantics used
are those restated in oracle/network.py and oracle/tf_semantics_np.py. `tf.image.non_max_suppression` is restated from
tensorflow/core/kernels/non_max_suppression_op.cc (greedy by descending score; IOU() without epsilon; a candidate enters
only with score > score_threshold); among EQUAL scores this restatement takes the lower index first - TF's order there is
that of its priority queue and is not documented.
"""
import itertools
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import network as onet

DEPTH = 128            # retinanet.py:10
TOWER_DEPTH = 64       # retinanet.py:46 (depth=64)
LEVELS = (3, 4, 5, 6, 7)
STRIDES = [8, 16, 32, 64, 128]               # retinanet.py:38-42
SCALES = [32, 64, 128, 256, 512]
SCALE_MULTIPLIERS = [1.0, 1.4142]
ASPECT_RATIOS = [1.0, 2.0, 0.5]
NUM_ANCHORS_PER_LOCATION = 6
EPSILON = 1e-8                               # constants.py:16
SCALE_FACTORS = [10.0, 10.0, 5.0, 5.0]       # constants.py:19
POSITIVES_THRESHOLD = 0.5                    # constants.py:31-32
NEGATIVES_THRESHOLD = 0.5


# ----------------------------------------------------------------------------- parameters
def head_param_shapes(depth_multiplier=1.0):
    """Ordered {reference variable name: shape} of the detector head (scopes of person_detector_model.py: 'fpn', 'box_net',
    'class_net', the per-level batch norms)."""
    s = OrderedDict()

    def bn(prefix, c):
        for n in ("gamma", "beta", "moving_mean", "moving_variance"):
            s[f"{prefix}/{n}"] = (c,)
    feat = {3: onet.depth(256, depth_multiplier), 4: onet.depth(512, depth_multiplier), 5: onet.depth(1024, depth_multiplier)}
    s["fpn/lateral5/kernel"] = (1, 1, feat[5], DEPTH)                 # fpn.py:38
    s["fpn/p5/kernel"] = (3, 3, DEPTH, DEPTH)                         # fpn.py:39
    s["fpn/p6/kernel"] = (3, 3, feat[5], DEPTH)                       # fpn.py:43
    bn("fpn/pre_p7_bn", DEPTH)                                        # fpn.py:44
    s["fpn/p7/kernel"] = (3, 3, DEPTH, DEPTH)                         # fpn.py:45
    for i in (4, 3):                                                  # fpn.py:49-53
        s[f"fpn/lateral{i}/kernel"] = (1, 1, feat[i], DEPTH)
        s[f"fpn/p{i}/kernel"] = (3, 3, DEPTH, DEPTH)
    for l in LEVELS:                                                  # retinanet.py:29-32
        bn(f"p{l}_batch_norm", DEPTH)
    for net, out_name, cout in (("box_net", "encoded_boxes", 4 * NUM_ANCHORS_PER_LOCATION), ("class_net", "logits", NUM_ANCHORS_PER_LOCATION)):
        for i in range(4):                                            # box_predictor.py:101-103,128-130
            s[f"{net}/conv3x3_{i}/kernel"] = (3, 3, DEPTH if i == 0 else TOWER_DEPTH, TOWER_DEPTH)
            for l in LEVELS:
                bn(f"{net}/batch_norm_{i}_for_level_{l}", TOWER_DEPTH)
        s[f"{net}/{out_name}/kernel"] = (3, 3, TOWER_DEPTH, cout)     # box_predictor.py:109-116,132-139
        s[f"{net}/{out_name}/bias"] = (cout,)
    return s


def init_head_params(seed=0, depth_multiplier=1.0, dtype=np.float32):
    """Seeded initial values of the reference's initialiser families: variance scaling for conv2d_same kernels
    (layer_utils.py:37), N(0, 0.01^2) for the two output convolutions, bias -log(99) for the logits (box_predictor.py:105-114)."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in head_param_shapes(depth_multiplier).items():
        if name.endswith("/gamma") or name.endswith("moving_variance"):
            v = np.ones(shape)
        elif name.endswith("/beta") or name.endswith("moving_mean"):
            v = np.zeros(shape)
        elif name == "class_net/logits/bias":
            v = np.full(shape, -math.log((1.0 - 0.01) / 0.01))
        elif name == "box_net/encoded_boxes/bias":
            v = np.zeros(shape)
        elif name in ("class_net/logits/kernel", "box_net/encoded_boxes/kernel"):
            v = rs.randn(*shape) * 0.01
        else:
            v = rs.randn(*shape) * math.sqrt(1.0 / (shape[0] * shape[1] * shape[2])) * 1.2
        out[name] = v.astype(dtype)
    return out


# ----------------------------------------------------------------------------- anchors (anchor_generator.py)
def tile_anchors(grid_height, grid_width, scales, aspect_ratios, stride, offset):
    """anchor_generator.py:119-166, float32 step by step. Returns [h*w*N, 4] absolute (ymin, xmin, ymax, xmax)."""
    f = np.float32
    ratio_sqrts = np.sqrt(aspect_ratios.astype(f))
    heights = scales.astype(f) / ratio_sqrts
    widths = scales.astype(f) * ratio_sqrts
    y_centers = np.arange(grid_height).astype(f) * f(stride) + f(offset[0])
    x_centers = np.arange(grid_width).astype(f) * f(stride) + f(offset[1])
    xc, yc = np.meshgrid(x_centers, y_centers)
    centers = np.stack([yc, xc], axis=2)[:, :, None, :].repeat(len(scales), axis=2)
    sizes = np.stack([heights, widths], axis=1)[None, None].repeat(grid_height, 0).repeat(grid_width, 1)
    boxes = np.concatenate([centers - f(0.5) * sizes, centers + f(0.5) * sizes], axis=3)
    return boxes.reshape(-1, 4).astype(f)


def generate_anchors(image_height, image_width):
    """anchor_generator.py:42-116. Returns (anchors [A,4] float32 normalised, [(h, w)] per level)."""
    f = np.float32
    ih, iw = f(image_height), f(image_width)
    pairs = list(itertools.product(SCALE_MULTIPLIERS, ASPECT_RATIOS))
    aspect_ratios = np.array([a for _, a in pairs], dtype=f)
    anchors, shapes = [], []
    for i, stride in enumerate(STRIDES):
        h = int(np.ceil(ih / f(stride)))
        w = int(np.ceil(iw / f(stride)))
        scales = np.array([m * SCALES[i] for m, _ in pairs], dtype=f)
        offset_y = f(0.5) * (ih - (f(h) - f(1.0)) * f(stride))
        offset_x = f(0.5) * (iw - (f(w) - f(1.0)) * f(stride))
        anchors.append(tile_anchors(h, w, scales, aspect_ratios, stride, (offset_y, offset_x)))
        shapes.append((h, w))
    anchors = np.concatenate(anchors, axis=0)
    scaler = np.array([ih, iw, ih, iw], dtype=f)
    return (anchors / scaler).astype(f), shapes


# ----------------------------------------------------------------------------- boxes (box_utils.py), float32 numpy
def iou(boxes1, boxes2):
    """box_utils.py:14-47: [N,4] x [M,4] -> [N,M], float32."""
    f = np.float32
    b1, b2 = boxes1.astype(f), boxes2.astype(f)
    ymin1, xmin1, ymax1, xmax1 = [b1[:, i:i + 1] for i in range(4)]
    ymin2, xmin2, ymax2, xmax2 = [b2[:, i:i + 1].T for i in range(4)]
    ih = np.maximum(f(0.0), np.minimum(ymax1, ymax2) - np.maximum(ymin1, ymin2))
    iw = np.maximum(f(0.0), np.minimum(xmax1, xmax2) - np.maximum(xmin1, xmin2))
    inter = ih * iw
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    unions = a1[:, None] + a2[None, :] - inter
    return np.clip(inter / (unions + f(EPSILON)), f(0.0), f(1.0)).astype(f)


def _center(b):
    h, w = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    return b[:, 0] + np.float32(0.5) * h, b[:, 1] + np.float32(0.5) * w, h, w


def encode(boxes, anchors):
    """box_utils.py:78-110, float32."""
    f = np.float32
    ya, xa, ha, wa = _center(anchors.astype(f))
    y, x, h, w = _center(boxes.astype(f))
    ha, wa, h, w = ha + f(EPSILON), wa + f(EPSILON), h + f(EPSILON), w + f(EPSILON)
    ty = (y - ya) / ha * f(SCALE_FACTORS[0])
    tx = (x - xa) / wa * f(SCALE_FACTORS[1])
    th = np.log(h / ha) * f(SCALE_FACTORS[2])
    tw = np.log(w / wa) * f(SCALE_FACTORS[3])
    return np.stack([ty, tx, th, tw], axis=1).astype(f)


def decode(codes, anchors):
    """box_utils.py:113-139, float32."""
    f = np.float32
    ya, xa, ha, wa = _center(anchors.astype(f))
    c = codes.astype(f)
    ty, tx, th, tw = c[:, 0] / f(SCALE_FACTORS[0]), c[:, 1] / f(SCALE_FACTORS[1]), c[:, 2] / f(SCALE_FACTORS[2]), c[:, 3] / f(SCALE_FACTORS[3])
    h, w = np.exp(th) * ha, np.exp(tw) * wa
    yc, xc = ty * ha + ya, tx * wa + xa
    return np.stack([yc - f(0.5) * h, xc - f(0.5) * w, yc + f(0.5) * h, xc + f(0.5) * w], axis=1).astype(f)


# ----------------------------------------------------------------------------- targets (training_target_creation.py)
def match_boxes(anchors, groundtruth_boxes, positives_threshold=POSITIVES_THRESHOLD, negatives_threshold=NEGATIVES_THRESHOLD,
                force_match_groundtruth=True):
    """training_target_creation.py:45-123. Returns int32 [A] in {-2, -1, 0..N-1}."""
    sim = iou(groundtruth_boxes, anchors)                         # [N, A]
    matches = sim.argmax(axis=0).astype(np.int32)                 # first occurrence on ties (tf.argmax)
    matched_vals = sim.max(axis=0)
    is_positive = (matched_vals >= np.float32(positives_threshold)).astype(np.int32)
    if positives_threshold == negatives_threshold:
        is_negative = 1 - is_positive
        matches = matches * is_positive + (-1 * is_negative)
    else:
        is_negative = (np.float32(negatives_threshold) > matched_vals).astype(np.int32)
        to_ignore = (1 - is_positive) * (1 - is_negative)
        matches = matches * is_positive + (-1 * is_negative) + (-2 * to_ignore)
    if force_match_groundtruth:
        forced_ids = sim.argmax(axis=1).astype(np.int32)          # [N]
        A = anchors.shape[0]
        indicators = np.zeros((sim.shape[0], A), np.int32)
        indicators[np.arange(sim.shape[0]), forced_ids] = 1
        row_ids = indicators.argmax(axis=0).astype(np.int32)      # (before the is_okay mask, as the reference does: :107-108)
        is_okay = (sim.max(axis=1) >= np.float32(0.05)).astype(np.int32)
        indicators = indicators * is_okay[:, None]
        mask = indicators.max(axis=0) > 0
        matches = np.where(mask, row_ids, matches).astype(np.int32)
    return matches


def get_training_targets(anchors, groundtruth_boxes):
    """training_target_creation.py:5-42 + create_targets :126-159. Returns (float32 [A,4], int32 [A])."""
    A = anchors.shape[0]
    if groundtruth_boxes.shape[0] > 0:
        matches = match_boxes(anchors, groundtruth_boxes)
    else:
        matches = np.full(A, -1, np.int32)
    targets = np.zeros((A, 4), np.float32)
    m = matches >= 0
    if m.any():
        targets[m] = encode(groundtruth_boxes[matches[m]], anchors[m])
    return targets, matches


# ----------------------------------------------------------------------------- network (fpn.py, box_predictor.py)
def fpn_retina(features, p, is_training, updates=None, taps=None):
    """fpn.py:36-55 with min_level=3, add_coarse_features=True, scope 'fpn'. features: NCHW c3, c4, c5 (activated).
    (onet._raw / _act / _w16: no-ops unless onet.storage_emulation(dtype) is active - then this restatement rounds where the 16-bit
    build stores, as oracle/network.py does for the keypoint net: raw convolution outputs and the FPN sums are stored tensors, normalised
    activations exist only as gradients, the matrix cores multiply 16-bit copies of the kernels.)"""
    R, A, W16 = onet._raw, onet._act, onet._w16
    x = R(onet.conv2d_same(features["c5"], W16(p["fpn/lateral5/kernel"])))
    out = {"p5": R(onet.conv2d_same(x, W16(p["fpn/p5/kernel"])))}
    p6 = R(onet.conv2d_same(features["c5"], W16(p["fpn/p6/kernel"]), stride=2))
    pre_p7 = A(F.relu(onet.batch_norm(p6, p, "fpn/pre_p7_bn", is_training, updates)))
    out["p6"], out["p7"] = p6, R(onet.conv2d_same(pre_p7, W16(p["fpn/p7/kernel"]), stride=2))
    if taps is not None:
        taps["x5"] = x
    for i in (4, 3):
        lateral = onet.conv2d_same(features[f"c{i}"], W16(p[f"fpn/lateral{i}/kernel"]))
        x = R(onet.nearest_neighbor_upsample(x) + lateral)        # (one store: the lateral's epilogue adds the upsampled sum)
        if taps is not None:
            taps[f"x{i}"] = x
        out[f"p{i}"] = R(onet.conv2d_same(x, W16(p[f"fpn/p{i}/kernel"])))
    return out


def tower(x, p, net, out_name, level, is_training, updates=None, taps=None):
    """box_net / class_net (box_predictor.py:93-142): 4 x (shared conv3x3 + per-level batch-norm + ReLU), then a 3x3 'same'
    convolution with bias. x NCHW; returns NHWC [b,h,w,cout]."""
    R, A, W16 = onet._raw, onet._act, onet._w16
    for i in range(4):
        x = R(onet.conv2d_same(x, W16(p[f"{net}/conv3x3_{i}/kernel"])))
        if taps is not None:
            taps[f"{net}/conv{i}/l{level}"] = x
        x = A(F.relu(onet.batch_norm(x, p, f"{net}/batch_norm_{i}_for_level_{level}", is_training, updates)))
    # padding='same', stride 1 == pad 1; the build stores the convolution WITHOUT its bias (the loss / NMS kernels add it in f32)
    y = R(onet.conv2d_same(x, W16(p[f"{net}/{out_name}/kernel"])))
    if taps is not None:
        taps[f"{net}/out/l{level}"] = y
    y = y + p[f"{net}/{out_name}/bias"].view(1, -1, 1, 1)
    return y.permute(0, 2, 3, 1)


def head_forward(backbone_features, p, is_training, updates=None, taps=None):
    """retinanet.py:24-58. backbone_features: NCHW dict (c3, c4, c5 used). Returns (encoded_boxes [b,A,4], class_predictions
    [b,A], per-level NHWC outputs) in the anchor order of reshape_and_concatenate (box_predictor.py:55-90)."""
    enriched = fpn_retina(backbone_features, p, is_training, updates, taps)
    if taps is not None:
        for n, v in enriched.items():
            taps[n] = v
    normalized = {n: onet._act(F.relu(onet.batch_norm(x, p, f"{n}_batch_norm", is_training, updates))) for n, x in enriched.items()}
    boxes, logits = [], []
    for l in LEVELS:
        boxes.append(tower(normalized[f"p{l}"], p, "box_net", "encoded_boxes", l, is_training, updates, taps))
        logits.append(tower(normalized[f"p{l}"], p, "class_net", "logits", l, is_training, updates, taps))
    b = boxes[0].shape[0]
    enc = torch.cat([t.reshape(b, -1, 4) for t in boxes], dim=1)
    cls = torch.cat([t.reshape(b, -1) for t in logits], dim=1)
    return enc, cls, (boxes, logits)


def forward(images, backbone_params, p, is_training, depth_multiplier=1.0, updates=None, taps=None):
    """person_detector_model.py:13-22: the backbone runs with is_training=False (frozen)."""
    feats = onet.mobilenet_v1(images, backbone_params, False, depth_multiplier)
    return head_forward(feats, p, is_training, updates, taps)


# ----------------------------------------------------------------------------- losses (retinanet.py:86-217)
def focal_loss(predictions, targets, weights, gamma=2.0, alpha=0.25):
    pos = targets == 1.0
    x = predictions
    nlp = torch.clamp(x, min=0) - x * targets + torch.log1p(torch.exp(-x.abs()))      # sigmoid_cross_entropy_with_logits
    prob = torch.sigmoid(x)
    p_t = torch.where(pos, prob, 1.0 - prob)
    mod = torch.pow(1.0 - p_t, gamma)
    wl = torch.where(pos, alpha * nlp, (1.0 - alpha) * nlp)
    return weights * mod * wl


def localization_loss(predictions, targets, weights):
    d = (predictions - targets).abs()
    loss = torch.where(d < 1.0, 0.5 * d * d, d - 0.5)
    return weights * loss.sum(dim=2)


def losses_fn(encoded_boxes, class_predictions, regression_targets, matches, gamma=2.0, alpha=0.25):
    """retinanet.py:100-144. matches int [b,A]; returns dict of the two normalised scalar losses."""
    is_matched = (matches >= 0).to(encoded_boxes.dtype)
    not_ignore = (matches >= -1).to(encoded_boxes.dtype)
    cls = focal_loss(class_predictions, is_matched, not_ignore, gamma, alpha).sum()
    loc = localization_loss(encoded_boxes, regression_targets, is_matched).sum()
    normalizer = torch.clamp(is_matched.sum(), min=1.0)
    return {"localization_loss": loc / normalizer, "classification_loss": cls / normalizer}


def total_loss_fn(enc, cls, regression_targets, matches, params, all_variables=None):
    """person_detector_model.py:33-45: weighted losses + weight decay over every 'weights'/'kernel' variable."""
    ls = losses_fn(enc, cls, regression_targets, matches, params.get("gamma", 2.0), params.get("alpha", 0.25))
    total = params["localization_loss_weight"] * ls["localization_loss"] + params["classification_loss_weight"] * ls["classification_loss"]
    if all_variables is not None and params.get("weight_decay", 0.0) > 0.0:
        total = total + onet.weight_decay_loss(all_variables, params["weight_decay"])
    return total, ls


# ----------------------------------------------------------------------------- post-processing (nms.py, retinanet.py:60-84)
def _nms_iou(a, b):
    """IOU() of tensorflow/core/kernels/non_max_suppression_op.cc, float32."""
    f = np.float32
    ymin_i, xmin_i, ymax_i, xmax_i = min(a[0], a[2]), min(a[1], a[3]), max(a[0], a[2]), max(a[1], a[3])
    ymin_j, xmin_j, ymax_j, xmax_j = min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])
    area_i = f(ymax_i - ymin_i) * f(xmax_i - xmin_i)
    area_j = f(ymax_j - ymin_j) * f(xmax_j - xmin_j)
    if area_i <= 0 or area_j <= 0:
        return f(0.0)
    iy = max(f(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), f(0.0))
    ix = max(f(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), f(0.0))
    inter = f(iy * ix)
    return f(inter / f(f(area_i + area_j) - inter))


def non_max_suppression(boxes, scores, max_output_size, iou_threshold, score_threshold):
    """tf.image.non_max_suppression (greedy). Returns selected indices (into boxes) in selection order."""
    order = sorted([i for i in range(len(scores)) if scores[i] > np.float32(score_threshold)], key=lambda i: (-scores[i], i))
    keep = []
    for i in order:
        if len(keep) >= max_output_size:
            break
        if all(not (_nms_iou(boxes[i], boxes[j]) > np.float32(iou_threshold)) for j in keep):
            keep.append(i)
    return keep


def get_predictions(encoded_boxes, class_predictions, anchors, score_threshold=0.05, iou_threshold=0.5, max_detections=25):
    """retinanet.py:60-84 + nms.py:6-61 on numpy float32 arrays [b,A,4], [b,A]. Scores = sigmoid in float32 of the logits."""
    f = np.float32
    b = encoded_boxes.shape[0]
    out_b = np.zeros((b, max_detections, 4), f)
    out_s = np.zeros((b, max_detections), f)
    out_n = np.zeros((b,), np.int32)
    for i in range(b):
        scores = (f(1.0) / (f(1.0) + np.exp(-class_predictions[i].astype(f)))).astype(f)
        conf = scores >= f(score_threshold)
        boxes = np.clip(decode(encoded_boxes[i][conf], anchors[conf]), f(0.0), f(1.0))
        sc = scores[conf]
        keep = non_max_suppression(boxes, sc, max_detections, iou_threshold, score_threshold)
        n = len(keep)
        out_b[i, :n], out_s[i, :n], out_n[i] = boxes[keep], sc[keep], n
    return out_b, out_s, out_n
