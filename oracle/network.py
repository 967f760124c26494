"""Oracle for the keypoint network, its losses and its optimizer step.

TEST INFRASTRUCTURE ONLY - never imported by the product path (multiposenet_amd).

CPU restatement (PyTorch-CPU primitives arranged to TensorFlow-1.15 semantics, f32 or
f64) of the reference's hot path:
    detector/backbones/mobilenet_v1.py, detector/fpn.py, detector/keypoint_subnet.py,
    detector/utils/layer_utils.py, keypoints_model.py, create_pb.py:73-76.
Backward passes come from torch autograd over these forward restatements.

PARITY UNPINNED for this part: the arithmetic lives in tensorflow==1.15 (README.md:15),
which is absent from /root/reference and cannot be installed here, and the reference
holds no tests/golden vectors for it. Every "TF-1.15 semantic" below is restated from
TensorFlow's documented op behaviour; `oracle/tf_semantics_np.py` re-derives the
layout/padding/resize rules and - loop by loop, without torch, citing the TF 1.15 source file of each op -
fused batch-norm (+ moving-average update), Adam, cosine decay, the losses and the mask halving, pinned by
hand-computed known-answer vectors and compared with this file in tests/test_oracle_network.py.
(The decode part, oracle/decode.py, IS pinned by goldens from the imported reference.)

Layout: NHWC at the API edge, NCHW inside (detector/constants.py:7), exactly like the
reference. Parameters use the reference's variable names and HWIO shapes.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BATCH_NORM_MOMENTUM = 0.95   # mobilenet_v1.py:7, layer_utils.py:5
BATCH_NORM_EPSILON = 1e-3    # mobilenet_v1.py:8, layer_utils.py:6
NUM_KEYPOINTS = 17           # constants.py:10
DEPTH = 128                  # keypoint_subnet.py:7

# (stride, filters) of the 13 depthwise-separable blocks, mobilenet_v1.py:59-65
STRIDES_AND_FILTERS = [
    (1, 64), (2, 128), (1, 128), (2, 256), (1, 256), (2, 512),
    (1, 512), (1, 512), (1, 512), (1, 512), (1, 512), (2, 1024), (1, 1024)]


def depth(x, depth_multiplier):
    return max(int(x * depth_multiplier), 8)   # mobilenet_v1.py:25-27


# ----------------------------------------------------------------------------- parameters
def param_shapes(depth_multiplier=1.0):
    """Ordered {reference variable name: shape} of every variable of the keypoint model
    (trainable ones and batch-norm moving statistics)."""
    s = OrderedDict()

    def bn(prefix, c):
        s[prefix + "/gamma"] = (c,)
        s[prefix + "/beta"] = (c,)
        s[prefix + "/moving_mean"] = (c,)
        s[prefix + "/moving_variance"] = (c,)

    c = depth(32, depth_multiplier)
    s["MobilenetV1/Conv2d_0/weights"] = (3, 3, 3, c)                 # mobilenet_v1.py:56
    bn("MobilenetV1/Conv2d_0/BatchNorm", c)
    for i, (_, f) in enumerate(STRIDES_AND_FILTERS, 1):
        s[f"MobilenetV1/Conv2d_{i}_depthwise/depthwise_weights"] = (3, 3, c, 1)   # :96-100
        bn(f"MobilenetV1/Conv2d_{i}_depthwise/BatchNorm", c)
        f = depth(f, depth_multiplier)
        s[f"MobilenetV1/Conv2d_{i}_pointwise/weights"] = (1, 1, c, f)             # :73
        bn(f"MobilenetV1/Conv2d_{i}_pointwise/BatchNorm", f)
        c = f
    feat = {2: depth(128, depth_multiplier), 3: depth(256, depth_multiplier),
            4: depth(512, depth_multiplier), 5: depth(1024, depth_multiplier)}
    s["keypoint_fpn/lateral5/kernel"] = (1, 1, feat[5], DEPTH)        # fpn.py:38
    s["keypoint_fpn/p5/kernel"] = (3, 3, DEPTH, DEPTH)               # fpn.py:39
    for i in (4, 3, 2):                                               # fpn.py:49-53
        s[f"keypoint_fpn/lateral{i}/kernel"] = (1, 1, feat[i], DEPTH)
        s[f"keypoint_fpn/p{i}/kernel"] = (3, 3, DEPTH, DEPTH)
    for l in (2, 3, 4, 5):                                            # keypoint_subnet.py:24-27
        bn(f"p{l}_batch_norm", DEPTH)
    for l in (2, 3, 4, 5):                                            # keypoint_subnet.py:30-35,75-78
        s[f"phi_subnet_{l}/conv1/kernel"] = (3, 3, DEPTH, DEPTH)
        bn(f"phi_subnet_{l}/bn1", DEPTH)
        s[f"phi_subnet_{l}/conv2/kernel"] = (3, 3, DEPTH, DEPTH)
        bn(f"phi_subnet_{l}/bn2", DEPTH)
    s["final_conv3x3/kernel"] = (3, 3, 4 * DEPTH, 64)                # keypoint_subnet.py:38
    bn("final_bn", 64)                                                # :39
    s["heatmaps/kernel"] = (1, 1, 64, NUM_KEYPOINTS + 1)             # :49-54
    s["heatmaps/bias"] = (NUM_KEYPOINTS + 1,)
    return s


def is_trainable(name):
    return not (name.endswith("moving_mean") or name.endswith("moving_variance"))


def init_params(seed=0, depth_multiplier=1.0, dtype=np.float32):
    """Deterministic initialisation following the reference's initialiser FAMILIES
    (not TF's random streams): variance-scaling for conv kernels (layer_utils.py:37),
    N(0, 1e-4) for the heatmaps kernel and bias -log(99) x17 + 0 (keypoint_subnet.py:41-53),
    gamma=1, beta=0, moving (0, 1)."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in param_shapes(depth_multiplier).items():
        if name.endswith("/gamma") or name.endswith("moving_variance"):
            v = np.ones(shape)
        elif name.endswith("/beta") or name.endswith("moving_mean"):
            v = np.zeros(shape)
        elif name == "heatmaps/bias":
            v = np.array([-math.log(99.0)] * NUM_KEYPOINTS + [0.0])
        elif name == "heatmaps/kernel":
            v = rs.randn(*shape) * 1e-4
        else:
            fan_in = shape[0] * shape[1] * (shape[2] if not name.endswith("depthwise_weights") else 1)
            v = rs.randn(*shape) * math.sqrt(1.0 / fan_in) * 1.2
        out[name] = v.astype(dtype)
    return out


def randomize_bn(params, seed=1):
    """Non-trivial gamma/beta/moving statistics so that parity tests exercise them."""
    rs = np.random.RandomState(seed)
    for k in params:
        c = params[k].shape
        if k.endswith("/gamma"):
            params[k] = (0.7 + 0.6 * rs.rand(*c)).astype(params[k].dtype)
        elif k.endswith("/beta"):
            params[k] = (0.3 * rs.randn(*c)).astype(params[k].dtype)
        elif k.endswith("moving_mean"):
            params[k] = (0.2 * rs.randn(*c)).astype(params[k].dtype)
        elif k.endswith("moving_variance"):
            params[k] = (0.5 + rs.rand(*c)).astype(params[k].dtype)
    return params


# ----------------------------------------------------------------------------- 16-bit storage emulation (tests only)
# The throughput build keeps every tensor that crosses a kernel boundary in bf16 (raw conv outputs,
# the FPN sums, the upsampled concat slices; backward: the gradients written by the data-gradient and batch-norm-apply
# kernels) and multiplies bf16 copies of the dense kernels; accumulation, batch-norm arithmetic, depthwise / stem / head
# weights and all master variables stay f32. `storage_emulation(torch.bfloat16)` makes this restatement round at the same
# places (autograd: the value is rounded on the way forward, the gradient on the way back), so that a bf16 step can be held
# against an oracle that differs from it by summation order only - at random initialisation the gradients are small
# residuals of large cancelling sums, and a few percent of storage error in the activations moves them by O(1) against the
# UNROUNDED oracle (tests/test_network_gpu.py::test_bf16_step_fused_and_unfused_reductions_against_the_oracle).
_STORAGE = None


class storage_emulation:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _STORAGE
        self.prev, _STORAGE = _STORAGE, self.dtype
        return self

    def __exit__(self, *a):
        global _STORAGE
        _STORAGE = self.prev


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype, fwd, bwd):
        ctx.dtype, ctx.bwd = dtype, bwd
        return x.to(dtype).to(x.dtype) if fwd else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dtype).to(g.dtype) if ctx.bwd else g), None, None, None


def _raw(x):
    """A tensor a kernel writes to HBM (rounded forward; its gradient is what a batch-norm-apply / data-gradient kernel wrote)."""
    return x if _STORAGE is None else _Round.apply(x, _STORAGE, True, True)


def _act(x):
    """A normalised activation: never stored (consumers re-create it on load), but its GRADIENT is a stored tensor."""
    return x if _STORAGE is None else _Round.apply(x, _STORAGE, False, True)


def _w16(w):
    """The 16-bit copy of a dense kernel that the matrix cores multiply (the gradient reaches the f32 master unrounded)."""
    return w if _STORAGE is None else _Round.apply(w, _STORAGE, True, False)


# ----------------------------------------------------------------------------- TF-1.15 op semantics
def _hwio_to_oihw(w):
    return w.permute(3, 2, 0, 1)


def tf_same_padding(size, k, stride):
    """TF 'SAME': out = ceil(size/stride); pad_total = max((out-1)*stride + k - size, 0);
    pad_before = pad_total // 2 (the extra pixel goes AFTER)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2


def conv2d_tf_same(x, w_hwio, stride):
    """slim.conv2d(padding='SAME') (mobilenet_v1.py:56,73): no bias when a normalizer is set."""
    k = w_hwio.shape[0]
    pt, pb = tf_same_padding(x.shape[2], k, stride)
    pl, pr = tf_same_padding(x.shape[3], k, stride)
    x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, _hwio_to_oihw(w_hwio), stride=stride)


def depthwise_conv2d_tf_same(x, w_hwc1, stride):
    """tf.nn.depthwise_conv2d(x, W[3,3,C,1], strides, 'SAME') (mobilenet_v1.py:101)."""
    k, _, c, _ = w_hwc1.shape
    pt, pb = tf_same_padding(x.shape[2], k, stride)
    pl, pr = tf_same_padding(x.shape[3], k, stride)
    x = F.pad(x, (pl, pr, pt, pb))
    w = w_hwc1.permute(2, 3, 0, 1)   # [C,1,3,3]
    return F.conv2d(x, w, stride=stride, groups=c)


def conv2d_same(x, w_hwio, stride=1, bias=None):
    """layer_utils.py:19-39: explicit symmetric zero pad 1 for k=3, then VALID conv, no bias."""
    k = w_hwio.shape[0]
    assert k in (1, 3) and stride in (1, 2)
    if k == 3:
        x = F.pad(x, (1, 1, 1, 1))
    return F.conv2d(x, _hwio_to_oihw(w_hwio), bias=bias, stride=stride)


def batch_norm(x, p, prefix, training, updates=None):
    """tf.layers.batch_normalization(axis=1, momentum=.95, epsilon=1e-3, fused=True)
    (mobilenet_v1.py:29-38, layer_utils.py:9-15).
    training: biased batch statistics normalise; TF-1.15 fused semantic: the moving variance is
    updated with the UNBIASED (n/(n-1)) batch variance; moving = moving*m + batch*(1-m)."""
    gamma, beta = p[prefix + "/gamma"], p[prefix + "/beta"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if updates is not None:
            n = x.shape[0] * x.shape[2] * x.shape[3]
            m = BATCH_NORM_MOMENTUM
            with torch.no_grad():
                updates[prefix + "/moving_mean"] = p[prefix + "/moving_mean"] * m + mean * (1 - m)
                updates[prefix + "/moving_variance"] = (p[prefix + "/moving_variance"] * m
                                                        + var * (n / max(n - 1, 1)) * (1 - m))
    else:
        mean, var = p[prefix + "/moving_mean"], p[prefix + "/moving_variance"]
    scale = gamma * torch.rsqrt(var + BATCH_NORM_EPSILON)
    return (x - mean[None, :, None, None]) * scale[None, :, None, None] + beta[None, :, None, None]


def nearest_neighbor_upsample(x):
    """fpn.py:58-76: tf.image.resize_nearest_neighbor(align_corners=False): out[i,j]=in[i//2,j//2]."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def resize_bilinear_legacy(x, out_h, out_w):
    """tf.image.resize_bilinear, TF-1.15 legacy (align_corners=False, half_pixel_centers=False):
    src = dst * (in/out); lo = floor(src); hi = min(lo+1, in-1); frac = src - lo. x is NCHW.
    Order of the two lerps: TF (resize_bilinear_op.cc compute_lerp) interpolates along x first, then y; this restatement
    does y first, then x. The result is the same bilinear form; the two orders differ by rounding only (<= 1 ulp of the
    working precision - tests/test_oracle_network.py compares them against oracle/tf_semantics_np.resize_bilinear_tf)."""
    def axis(n_in, n_out):
        scale = n_in / n_out
        src = torch.arange(n_out, dtype=torch.float64) * scale
        lo = torch.floor(src).long().clamp(max=n_in - 1)
        hi = (lo + 1).clamp(max=n_in - 1)
        frac = (src - lo.double()).to(x.dtype)
        return lo, hi, frac
    ylo, yhi, fy = axis(x.shape[2], out_h)
    xlo, xhi, fx = axis(x.shape[3], out_w)
    top = x[:, :, ylo, :]
    bot = x[:, :, yhi, :]
    rows = top + (bot - top) * fy[None, None, :, None]
    left = rows[:, :, :, xlo]
    right = rows[:, :, :, xhi]
    return left + (right - left) * fx[None, None, None, :]


# ----------------------------------------------------------------------------- network
def mobilenet_v1(images, p, is_training, depth_multiplier=1.0, updates=None, taps=None):
    """mobilenet_v1.py:11-79. images: [b,h,w,3] in [0,1] (NHWC). Returns NCHW c2..c5."""
    x = 2.0 * images - 1.0                                          # :41
    x = x.permute(0, 3, 1, 2)                                       # :53
    name = "MobilenetV1/Conv2d_0"
    x = _raw(conv2d_tf_same(x, p[name + "/weights"], 2))            # :56
    if taps is not None:
        taps[name + "/raw"] = x
    x = _act(F.relu6(batch_norm(x, p, name + "/BatchNorm", is_training, updates)))
    feats = {}
    for i, (stride, _) in enumerate(STRIDES_AND_FILTERS, 1):        # :66-74
        name = f"MobilenetV1/Conv2d_{i}_depthwise"
        x = _raw(depthwise_conv2d_tf_same(x, p[name + "/depthwise_weights"], stride))
        if taps is not None:
            taps[name + "/raw"] = x
        x = _act(F.relu6(batch_norm(x, p, name + "/BatchNorm", is_training, updates)))
        name = f"MobilenetV1/Conv2d_{i}_pointwise"
        x = _raw(conv2d_tf_same(x, _w16(p[name + "/weights"]), 1))
        if taps is not None:
            taps[name + "/raw"] = x
        x = _act(F.relu6(batch_norm(x, p, name + "/BatchNorm", is_training, updates)))
        feats[name] = x
    return {"c2": feats["MobilenetV1/Conv2d_3_pointwise"], "c3": feats["MobilenetV1/Conv2d_5_pointwise"],
            "c4": feats["MobilenetV1/Conv2d_11_pointwise"], "c5": feats["MobilenetV1/Conv2d_13_pointwise"]}


def feature_pyramid_network(features, p, scope="keypoint_fpn", min_level=2, taps=None):
    """fpn.py:36-55 with add_coarse_features=False (the keypoint configuration)."""
    x = _raw(conv2d_same(features["c5"], _w16(p[f"{scope}/lateral5/kernel"])))          # :38
    if taps is not None:
        taps["x5"] = x
    enriched = {"p5": _raw(conv2d_same(x, _w16(p[f"{scope}/p5/kernel"])))}              # :39
    for i in reversed(range(min_level, 5)):                                 # :49-53
        lateral = conv2d_same(features[f"c{i}"], _w16(p[f"{scope}/lateral{i}/kernel"]))
        x = _raw(nearest_neighbor_upsample(x) + lateral)                    # (one store: the lateral's epilogue adds the upsampled sum)
        if taps is not None:
            taps[f"x{i}"] = x
        enriched[f"p{i}"] = _raw(conv2d_same(x, _w16(p[f"{scope}/p{i}/kernel"])))
    return enriched


def phi_subnet(x, p, scope, is_training, upsample, updates=None, taps=None):
    """keypoint_subnet.py:65-91."""
    x = _raw(conv2d_same(x, _w16(p[scope + "/conv1/kernel"])))
    if taps is not None:
        taps[scope + "/y1"] = x
    x = _act(F.relu(batch_norm(x, p, scope + "/bn1", is_training, updates)))
    x = _raw(conv2d_same(x, _w16(p[scope + "/conv2/kernel"])))
    if taps is not None:
        taps[scope + "/y2"] = x
    x = _act(F.relu(batch_norm(x, p, scope + "/bn2", is_training, updates)))
    x = resize_bilinear_legacy(x, upsample * x.shape[2], upsample * x.shape[3])      # :86
    return _raw(x) if upsample > 1 else x      # (the upsampled levels are stored activated; level 2 stays the raw y2)


def keypoint_subnet(backbone_features, p, is_training, updates=None, taps=None):
    """keypoint_subnet.py:11-62. Returns (heatmaps NHWC [b,h/4,w/4,18] logits,
    enriched_features NHWC dict p2..p5 - the PRE-batch-norm FPN outputs)."""
    enriched = feature_pyramid_network(backbone_features, p, taps=taps)     # :20-23
    normalized = {n: _act(F.relu(batch_norm(x, p, f"{n}_batch_norm", is_training, updates)))
                  for n, x in enriched.items()}                             # :24-27
    ups = []
    for level in range(2, 6):                                               # :30-35
        ups.append(phi_subnet(normalized[f"p{level}"], p, f"phi_subnet_{level}", is_training,
                              2 ** (level - 2), updates, taps))
    x = torch.cat(ups, dim=1)                                               # :37
    if taps is not None:
        taps["concat"] = x
    x = _raw(conv2d_same(x, _w16(p["final_conv3x3/kernel"])))               # :38
    if taps is not None:
        taps["final"] = x
    x = _act(F.relu(batch_norm(x, p, "final_bn", is_training, updates)))    # :39
    heat = conv2d_same(x, p["heatmaps/kernel"], bias=p["heatmaps/bias"])    # :49-54
    return heat.permute(0, 2, 3, 1), {n: v.permute(0, 2, 3, 1) for n, v in enriched.items()}   # :56-62


def forward(images, p, is_training, depth_multiplier=1.0, updates=None, taps=None):
    """taps: optional dict that receives the RAW (pre-batch-norm) output of every conv, NCHW."""
    feats = mobilenet_v1(images, p, is_training, depth_multiplier, updates, taps)
    return keypoint_subnet(feats, p, is_training, updates, taps)


# ----------------------------------------------------------------------------- losses (keypoints_model.py)
def focal_loss(heatmaps, num_boxes, predictions, alpha=2.0, beta=4.0):
    """keypoints_model.py:141-178 (CornerNet focal loss on logits). Returns [b,h,w]."""
    y, x = heatmaps, predictions
    pos = (y == 1.0)
    z = pos.to(x.dtype)
    # tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))
    ce = torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs()))
    y_hat = torch.sigmoid(x)
    weights = torch.where(pos, (1.0 - y_hat) ** alpha, (1.0 - y) ** beta * y_hat ** alpha)
    normalizer = num_boxes.to(x.dtype).reshape(-1, 1, 1) + 1.0
    return (weights * ce).sum(3) / normalizer


def l2_loss(t):
    return (t * t).sum() / 2.0   # tf.nn.l2_loss


def losses_fn(heat, enriched, labels):
    """keypoints_model.py:31-79. heat [b,h,w,18] logits, enriched NHWC dict, labels dict
    (heatmaps [b,h,w,17], loss_masks [b,h,w], segmentation_masks [b,h,w], num_boxes [b])."""
    losses = OrderedDict()
    heatmaps = labels["heatmaps"]
    normalizer = float(heatmaps.shape[0])
    seg = labels["segmentation_masks"].unsqueeze(3)
    lm = labels["loss_masks"].unsqueeze(3)
    pred_heat = heat[..., :17]
    pred_seg = heat[..., 17:18]
    fl = focal_loss(heatmaps, labels["num_boxes"], pred_heat)
    losses["focal_loss"] = (lm.squeeze(3) * fl).sum() / normalizer                   # :46-52
    losses["regression_loss"] = 1e-3 * l2_loss(lm * (pred_seg - seg)) / normalizer   # :54-55
    for level in range(2, 6):                                                        # :59-74
        x = enriched[f"p{level}"][..., 0:1]
        losses[f"segmentation_loss_at_level_{level}"] = 1e-5 * l2_loss(lm * (x - seg)) / normalizer
        # tf.image.resize_bilinear(., [h//2, w//2]) legacy semantic == [::2, ::2]
        seg = seg[:, ::2, ::2, :]
        lm = lm[:, ::2, ::2, :]
    total = sum(losses.values())
    return total, losses


def per_pixel_reg_loss(heat, labels):
    """keypoints_model.py:81-90 (eval metric)."""
    hm = labels["heatmaps"]
    b, h, w, _ = hm.shape
    lm = labels["loss_masks"].unsqueeze(3)
    return l2_loss(lm * (torch.sigmoid(heat[..., :17]) - hm)) / (b * h * w)


def weight_decay_loss(p, weight_decay):
    """keypoints_model.py:129-138."""
    tot = 0.0
    for k, v in p.items():
        if ("weights" in k or "kernel" in k) and "depthwise_weights" not in k:
            tot = tot + weight_decay * l2_loss(v)
    return tot


# ----------------------------------------------------------------------------- optimizer (keypoints_model.py:107-120)
def cosine_decay(initial_lr, global_step, decay_steps, alpha=1e-4):
    step = min(global_step, decay_steps)
    cosine = 0.5 * (1.0 + math.cos(math.pi * step / decay_steps))
    return initial_lr * ((1.0 - alpha) * cosine + alpha)


def adam_step(param, grad, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8, clip=200.0):
    """TF-1.15 AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps),
    after tf.clip_by_value(g, -200, 200) (keypoints_model.py:119). numpy, in place."""
    g = np.clip(grad, -clip, clip)
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m[...] = beta1 * m + (1.0 - beta1) * g
    v[...] = beta2 * v + (1.0 - beta2) * g * g
    param[...] = param - (lr_t * m / (np.sqrt(v) + eps)).astype(param.dtype)


def train_step(params_np, m_np, v_np, images, labels, global_step, hp, dtype=torch.float32):
    """One full TRAIN step of model_fn (keypoints_model.py:6-126) on numpy state, in place.
    hp: dict with initial_learning_rate, num_steps, weight_decay, depth_multiplier.
    Returns (total_loss, losses dict, grads dict)."""
    p = {k: torch.tensor(v, dtype=dtype, requires_grad=is_trainable(k)) for k, v in params_np.items()}
    updates = {}
    heat, enriched = forward(torch.as_tensor(images, dtype=dtype), p, True,
                             hp.get("depth_multiplier", 1.0), updates)
    lab = {k: torch.as_tensor(v) if k == "num_boxes" else torch.as_tensor(v, dtype=dtype)
           for k, v in labels.items()}
    total, losses = losses_fn(heat, enriched, lab)
    if hp.get("weight_decay", 0.0) > 0.0:
        total = total + weight_decay_loss(p, hp["weight_decay"])
    total.backward()
    lr = cosine_decay(hp["initial_learning_rate"], global_step, hp["num_steps"])
    grads = {}
    for k in params_np:
        if is_trainable(k):
            g = p[k].grad.detach().numpy().astype(params_np[k].dtype)
            grads[k] = g
            adam_step(params_np[k], g, m_np[k], v_np[k], lr, global_step + 1)
    for k, vv in updates.items():
        params_np[k][...] = vv.detach().numpy()
    return float(total.detach()), {k: float(v.detach()) for k, v in losses.items()}, grads
