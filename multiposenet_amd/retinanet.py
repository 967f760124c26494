"""PersonDetectorNet: the RetinaNet person-detector head on the frozen MobileNet backbone (BASELINE
config 4) - forward, anchor matching, losses, backward, TF-Adam step and the NMS post-processing, on the HIP kernels.

Mirrors person_detector_model.py:8-81 (backbone with is_training=False and excluded from the optimizer's var_list),
detector/retinanet.py:13-217, detector/fpn.py:36-55 (min_level=3, add_coarse_features=True, scope 'fpn'),
detector/box_predictor.py:6-142 (towers shared over the five levels, batch-norm per level), detector/anchor_generator.py,
detector/training_target_creation.py and detector/utils/nms.py. Variable names and HWIO shapes are the reference's.

Everything dense runs on the kernels of the keypoint path: the 3x3 towers as ONE grouped launch per stage over the five
pyramid levels (mpn_conv_fwd_grouped - shared weights, per-level batch-norm affine on load), batch-norm statistics from the
convolution epilogues with one batched finalize per stage, grouped batch-norm backward passes; the stride-2 3x3
convolutions p6 / p7 are a gather (mpn_patchify3x3s2) + the 1x1 kernel. Image height and width must be multiples of 128
(detector/constants.py:4) - BASELINE's 800 x 1333 is padded to 896 x 1408.
"""
import ctypes
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, ops
from ._lib import ACT_RELU, call, ptr, stream_ptr
from .net import KeypointNet, _Arena, depth

DEPTH = 128                 # retinanet.py:10
TOWER_DEPTH = 64            # retinanet.py:46
LEVELS = (3, 4, 5, 6, 7)
STRIDES = [8, 16, 32, 64, 128]          # retinanet.py:38-42
SCALES = [32, 64, 128, 256, 512]
SCALE_MULTIPLIERS = [1.0, 1.4142]
ASPECT_RATIOS = [1.0, 2.0, 0.5]
APL = 6                                 # anchors per location
POSITIVES_THRESHOLD = 0.5               # constants.py:31-32
NEGATIVES_THRESHOLD = 0.5
NETS = (("box_net", "encoded_boxes", 4 * APL), ("class_net", "logits", APL))
LOSS_NAMES = ["localization_loss", "classification_loss", "regularization_loss", "total_loss"]


def head_variable_shapes(depth_multiplier=1.0):
    """Ordered {reference variable name: shape} of the head (trainables + batch-norm moving statistics)."""
    s = OrderedDict()

    def bn(prefix, c):
        for n in ("gamma", "beta", "moving_mean", "moving_variance"):
            s[f"{prefix}/{n}"] = (c,)
    feat = {3: depth(256, depth_multiplier), 4: depth(512, depth_multiplier), 5: depth(1024, depth_multiplier)}
    s["fpn/lateral5/kernel"] = (1, 1, feat[5], DEPTH)
    s["fpn/p5/kernel"] = (3, 3, DEPTH, DEPTH)
    s["fpn/p6/kernel"] = (3, 3, feat[5], DEPTH)
    bn("fpn/pre_p7_bn", DEPTH)
    s["fpn/p7/kernel"] = (3, 3, DEPTH, DEPTH)
    for i in (4, 3):
        s[f"fpn/lateral{i}/kernel"] = (1, 1, feat[i], DEPTH)
        s[f"fpn/p{i}/kernel"] = (3, 3, DEPTH, DEPTH)
    for l in LEVELS:
        bn(f"p{l}_batch_norm", DEPTH)
    for net, out_name, cout in NETS:
        for i in range(4):
            s[f"{net}/conv3x3_{i}/kernel"] = (3, 3, DEPTH if i == 0 else TOWER_DEPTH, TOWER_DEPTH)
            for l in LEVELS:
                bn(f"{net}/batch_norm_{i}_for_level_{l}", TOWER_DEPTH)
        s[f"{net}/{out_name}/kernel"] = (3, 3, TOWER_DEPTH, cout)
        s[f"{net}/{out_name}/bias"] = (cout,)
    return s


def _trainable(name):
    return not (name.endswith("moving_mean") or name.endswith("moving_variance"))


_BN0 = tuple(f"{net}/batch_norm_0_for_level_" for net in ("box_net", "class_net"))


def _arena_order(shapes):
    """The arenas' order (an internal matter: checkpoints go by name): batch_norm_0 of the box tower and of the class tower lie side
    by side, level by level and vector by vector - the two towers' first convolutions read the same tensor and run as ONE 128 -> 128
    convolution (PersonDetectorNet.merge_tower0), whose batch-norm is then one 128-channel layer over adjacent variables."""
    rest = OrderedDict((k, v) for k, v in shapes.items() if not k.startswith(_BN0))
    for l in LEVELS:
        for kind in ("gamma", "beta", "moving_mean", "moving_variance"):
            for pre in _BN0:
                rest[f"{pre}{l}/{kind}"] = shapes[f"{pre}{l}/{kind}"]
    return rest


def initial_head_values(seed=0, depth_multiplier=1.0):
    """Seeded values of the reference's initialiser families (layer_utils.py:37; box_predictor.py:105-114,132-139)."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in head_variable_shapes(depth_multiplier).items():
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
        out[name] = v.astype(np.float32)
    return out


def generate_anchors(image_height, image_width):
    """AnchorGenerator.__call__ (anchor_generator.py:42-116), float32 step by step: ([A,4] normalised (ymin, xmin, ymax, xmax),
    [(h, w)] per level). Host-side: a few hundred KB computed once per image size."""
    import itertools
    f = np.float32
    ih, iw = f(image_height), f(image_width)
    pairs = list(itertools.product(SCALE_MULTIPLIERS, ASPECT_RATIOS))
    ratios = np.array([a for _, a in pairs], dtype=f)
    out, shapes = [], []
    for i, stride in enumerate(STRIDES):
        h, w = int(np.ceil(ih / f(stride))), int(np.ceil(iw / f(stride)))
        scales = np.array([m * SCALES[i] for m, _ in pairs], dtype=f)
        rs_ = np.sqrt(ratios)
        heights, widths = scales / rs_, scales * rs_                          # tile_anchors, :142-147
        oy = f(0.5) * (ih - (f(h) - f(1.0)) * f(stride))
        ox = f(0.5) * (iw - (f(w) - f(1.0)) * f(stride))
        yc = np.arange(h).astype(f) * f(stride) + oy
        xc = np.arange(w).astype(f) * f(stride) + ox
        xg, yg = np.meshgrid(xc, yc)
        centers = np.stack([yg, xg], axis=2)[:, :, None, :].repeat(len(scales), axis=2)
        sizes = np.stack([heights, widths], axis=1)[None, None].repeat(h, 0).repeat(w, 1)
        out.append(np.concatenate([centers - f(0.5) * sizes, centers + f(0.5) * sizes], axis=3).reshape(-1, 4).astype(f))
        shapes.append((h, w))
    anchors = np.concatenate(out, axis=0) / np.array([ih, iw, ih, iw], dtype=f)
    return anchors.astype(f), shapes


class _Conv:
    """A dense conv of the head: reference variable view (+ gradient view) and packed MFMA operands. `as1x1`: a 3x3 HWIO
    kernel used as the [1,1,9*Cin,Cout] matrix behind mpn_patchify3x3s2 (the stride-2 convolutions)."""

    def __init__(self, name, w, dw, dtype, as1x1=False, pad_cout=0):
        self.name, self.w, self.dw = name, w, dw
        k, _, cin, cout = w.shape
        self.pad = None
        if pad_cout:      # class_net/logits has 6 output channels: the kernels want multiples of 8 - two zero columns
            self.pad = torch.zeros((k, k, cin, pad_cout), dtype=torch.float32, device=w.device)
            self.dpad = torch.zeros_like(self.pad)
            cout = pad_cout
        src = self.pad if self.pad is not None else w
        if as1x1:
            src = src.view(1, 1, k * k * cin, cout)
        self.ksize, self.cin, self.cout = src.shape[0], src.shape[2], src.shape[3]
        self.src = src
        self.refresh_pad()
        self.packed = ops.PackedConv(src, dtype)

    def refresh_pad(self):
        if self.pad is not None:
            self.pad[..., :self.w.shape[3]].copy_(self.w)

    def repack(self):
        if getattr(self, "packed_unused", False):
            return
        self.refresh_pad()
        self.packed.repack()


class _MergedConv:
    """The first convolutions of the two towers as one: HWIO kernels side by side along the output channels (a staging tensor
    refreshed with the variables), packed for the forward pass and for the data gradient. The weight gradients stay per tower."""

    def __init__(self, convs, dtype):
        self.convs = convs
        self.src = torch.cat([c.w for c in convs], dim=3).contiguous()
        self.ksize, self.cin, self.cout = self.src.shape[0], self.src.shape[2], self.src.shape[3]
        self.packed = ops.PackedConv(self.src, dtype)
        self.name = "+".join(c.name for c in convs)
        self.dsrc = torch.zeros_like(self.src)      # the merged weight gradient; split_grad() hands the halves to the variables' gradients
        self.pad = None
        for c in convs:
            c.packed_unused = True      # (nothing reads the separate packed operands: not refreshed)

    def split_grad(self):
        o = 0
        for c in self.convs:
            c.dw.copy_(self.dsrc[..., o:o + c.cout])
            o += c.cout

    def repack(self):
        torch.cat([c.w for c in self.convs], dim=3, out=self.src)
        self.packed.repack()


@_lib.device_guarded("_init", "load_state_dict", "repack_weights", "forward", "create_targets", "compute_losses", "backward",
                     "optimizer_step", "train_step", "predict", "nms", "head_forward")
class PersonDetectorNet:
    def __init__(self, backbone_values=None, head_values=None, depth_multiplier=1.0, dtype=torch.bfloat16, device="cuda:0", seed=0,
                 backbone=None):
        """backbone: a KeypointNet whose MobileNet this detector SHARES (the joint inference graph of create_pb.py:64-71 runs
        one backbone under the keypoint subnet and the RetinaNet head) instead of building its own frozen copy."""
        dev = torch.device(device) if backbone is None else backbone.device
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if backbone is not None:
            dtype, depth_multiplier = backbone.dtype, backbone.dm
        self.device, self.dtype, self.dm = dev, dtype, depth_multiplier
        self._init(backbone_values, head_values, seed, backbone)

    def _init(self, backbone_values, head_values, seed, backbone=None):
        # the frozen backbone: a KeypointNet's MobileNet part in inference mode (person_detector_model.py:14-17)
        self.backbone = backbone if backbone is not None else KeypointNet(depth_multiplier=self.dm, dtype=self.dtype, device=self.device, seed=seed)
        if backbone_values is not None:
            self.backbone.load_state_dict({k: v for k, v in backbone_values.items() if k.startswith("MobilenetV1/")}, strict=False)
        shapes = head_variable_shapes(self.dm)
        order = _arena_order(shapes)
        self._train_arena = _Arena(OrderedDict((k, v) for k, v in order.items() if _trainable(k)), self.device)
        self._stat_arena = _Arena(OrderedDict((k, v) for k, v in order.items() if not _trainable(k)), self.device)
        self.theta, self.grad = self._train_arena.new(), self._train_arena.new()
        self.adam_m, self.adam_v = self._train_arena.new(), self._train_arena.new()
        self.moving = self._stat_arena.new()

        def named(arena, flat):      # (dictionaries in the reference's variable order, whatever the arena's)
            v = arena.views(flat)
            return OrderedDict((k, v[k]) for k in shapes if k in v)
        self.vars, self.grads = named(self._train_arena, self.theta), named(self._train_arena, self.grad)
        self.stats = named(self._stat_arena, self.moving)
        self.global_step = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._convs = []
        self._wversion = 0
        self.fuse_conv_bn = True       # set before the first backward pass of a shape (the finalize tables are built once)
        # The first convolutions of the box and class towers read the same tensor (box_predictor.py:101-103 under both scopes): ONE
        # 128 -> 128 convolution on the 128-channel tile kernel instead of two 128 -> 64 ones, ONE 128 -> 128 data gradient whose
        # contraction over the 128 merged channels IS the sum the two towers send into p{l}_batch_norm (and reduces for that layer).
        # MPN_RETINA_MERGE=0: the two towers separately (A/B runs). Fixed before _build_layers.
        self.merge_tower0 = os.environ.get("MPN_RETINA_MERGE", "1") != "0"
        # the head's inference affines (45 small launches) and the p6 operand cast are recomputed on EVERY inference pass
        # unless the owner opts in (inference/detector.py does and compares `var_version` before each graph replay): the clean
        # flag is host state, a replayed hipGraph of a train step cannot clear it
        self.cache_inference_affine = False
        self._infer_clean = False
        self.var_version = 0
        self.backbone.cache_inference_affine = True     # frozen here: its inference affines change only with its variables
        self._l2 = None
        self._wd = None
        self._built = False
        self.tower0m = None
        self.load_state_dict(head_values if head_values is not None else initial_head_values(seed, self.dm))
        self._build_layers()
        self._bufs = {}

    # ------------------------------------------------------------------ variables
    def state_dict(self):
        out = OrderedDict()
        for k, v in list(self.vars.items()) + list(self.stats.items()):
            out[k] = v.detach().cpu().numpy().copy()
        return out

    def load_state_dict(self, values, strict=True):
        for k, v in values.items():
            dst = self.vars.get(k, self.stats.get(k))
            if dst is None:
                if strict:
                    raise KeyError(f"unknown variable {k}")
                continue
            v = np.asarray(v, dtype=np.float32)
            if tuple(v.shape) != tuple(dst.shape):
                raise ValueError(f"{k}: shape {v.shape} != {tuple(dst.shape)}")
            dst.copy_(torch.from_numpy(v))
        if strict:
            missing = [k for k in list(self.vars) + list(self.stats) if k not in values]
            if missing:
                raise KeyError(f"missing variables: {missing[:5]}...")
        self.mark_variables_changed()
        if self._built:
            self.repack_weights()

    def mark_variables_changed(self):
        """The head's variables or moving statistics changed (see KeypointNet.mark_variables_changed)."""
        self.var_version = getattr(self, "var_version", 0) + 1
        self._infer_clean = False

    def _bn(self, prefix, act=ACT_RELU):
        bn = ops.BNState(self.vars[prefix + "/gamma"], self.vars[prefix + "/beta"], self.stats[prefix + "/moving_mean"],
                         self.stats[prefix + "/moving_variance"], act)
        bn.dgamma, bn.dbeta = self.grads[prefix + "/gamma"], self.grads[prefix + "/beta"]
        bn.name = prefix
        return bn

    def _conv(self, name, **kw):
        c = _Conv(name, self.vars[name], self.grads[name], self.dtype, **kw)
        self._convs.append(c)
        return c

    def _build_layers(self):
        self.lateral = {l: self._conv(f"fpn/lateral{l}/kernel") for l in (5, 4, 3)}
        self.pconv = {l: self._conv(f"fpn/p{l}/kernel") for l in (5, 4, 3)}
        self.pconv[6] = self._conv("fpn/p6/kernel", as1x1=True)
        self.pconv[7] = self._conv("fpn/p7/kernel", as1x1=True)
        self.pre_p7_bn = self._bn("fpn/pre_p7_bn")
        self.p_bn = {l: self._bn(f"p{l}_batch_norm") for l in LEVELS}
        self.tower, self.tower_bn, self.out_conv, self.out_bias, self.out_dbias = {}, {}, {}, {}, {}
        self.bn0m = {l: self._bn_pair(l) for l in LEVELS} if self.merge_tower0 else None
        for k, (net, out_name, cout) in enumerate(NETS):
            self.tower[net] = [self._conv(f"{net}/conv3x3_{i}/kernel") for i in range(4)]
            self.tower_bn[net] = [{l: self._bn(f"{net}/batch_norm_{i}_for_level_{l}") for l in LEVELS} for i in range(4)]
            if self.merge_tower0:
                for l in LEVELS:
                    bn = self.bn0m[l].channel_slice(k * TOWER_DEPTH, (k + 1) * TOWER_DEPTH)
                    bn.name = f"{net}/batch_norm_0_for_level_{l}"
                    self.tower_bn[net][0][l] = bn
            self.out_conv[net] = self._conv(f"{net}/{out_name}/kernel", pad_cout=8 if cout % 8 else 0)
            self.out_bias[net] = self.vars[f"{net}/{out_name}/bias"]
            self.out_dbias[net] = self.grads[f"{net}/{out_name}/bias"]
        self.all_bn = [self.pre_p7_bn] + [self.p_bn[l] for l in LEVELS] + \
            [self.tower_bn[net][i][l] for net, _, _ in NETS for i in range(4) for l in LEVELS]
        self.tower0m = _MergedConv([self.tower[net][0] for net, _, _ in NETS], self.dtype) if self.merge_tower0 else None
        self._built = True

    def _bn_pair(self, l):
        """batch_norm_0_for_level_l of the two towers as ONE 128-channel layer (box channels first): views over adjacent variables."""
        names = [f"{pre}{l}" for pre in _BN0]

        def pair(flat, arena, kind):
            (o0, n0, _), (o1, n1, _) = arena.offsets[f"{names[0]}/{kind}"], arena.offsets[f"{names[1]}/{kind}"]
            if o1 != o0 + n0:
                raise RuntimeError("the towers' first batch-norms must lie side by side in the arena")
            return flat[o0:o1 + n1]
        bn = ops.BNState(pair(self.theta, self._train_arena, "gamma"), pair(self.theta, self._train_arena, "beta"),
                         pair(self.moving, self._stat_arena, "moving_mean"), pair(self.moving, self._stat_arena, "moving_variance"), ACT_RELU)
        bn.dgamma, bn.dbeta = pair(self.grad, self._train_arena, "gamma"), pair(self.grad, self._train_arena, "beta")
        bn.name = f"box_net+class_net/batch_norm_0_for_level_{l}"
        return bn

    def repack_weights(self):
        self._wversion += 1
        for c in self._convs:
            c.repack()
        if self.tower0m is not None:
            self.tower0m.repack()

    def _fused_out_bn(self, net):
        """The output convolution's data gradient also reduces for the tower's last batch-norm."""
        return self.fuse_conv_bn and ops.conv_bwd_data_bn_supported(self.out_conv[net].cout, TOWER_DEPTH, 3, self.dtype)

    def _fused_p_bn(self):
        """The merged first-layer data gradient (128 -> 128) also reduces for p{l}_batch_norm."""
        return self.merge_tower0 and self.fuse_conv_bn and ops.conv_bwd_data_bn_supported(2 * TOWER_DEPTH, DEPTH, 3, self.dtype)

    def _fused_conv_bn(self):
        """The towers' 3x3 data gradients also reduce for the batch-norm they feed (mpn_conv_bwd_data_bn_grouped)."""
        return self.fuse_conv_bn and ops.conv_bwd_data_bn_supported(TOWER_DEPTH, TOWER_DEPTH, 3, self.dtype)

    # ------------------------------------------------------------------ buffers
    def _buffers(self, N, H, W):
        key = (N, H, W)
        b = self._bufs.get(key)
        if b is not None:
            return b
        if H % 128 or W % 128:
            raise ValueError(f"image height and width must be multiples of 128 (got {H}x{W})")
        dt, dev = self.dtype, self.device

        def act(h, w, c):
            return torch.empty((N, h, w, c), dtype=dt, device=dev)
        anchors, shapes = generate_anchors(H, W)
        lv = {l: shapes[i] for i, l in enumerate(LEVELS)}
        b = {"shape": key, "lv": lv, "A": anchors.shape[0], "anchors": torch.from_numpy(anchors).to(dev)}
        b["bb"] = None      # the backbone's buffers: allocated by forward() (head_forward on shared features needs none)
        b["x"] = {l: act(*lv[l], DEPTH) for l in (3, 4, 5)}
        b["p"] = {l: act(*lv[l], DEPTH) for l in LEVELS}
        c5 = self.pconv[6].cin // 9
        b["patches6"] = act(*lv[6], 9 * c5)
        b["patches7"] = act(*lv[7], 9 * DEPTH)
        b["t"] = {net: [{l: act(*lv[l], TOWER_DEPTH) for l in LEVELS} for _ in range(4)] for net, _, _ in NETS}
        if self.merge_tower0:      # the towers' first raw outputs: channel slices of ONE tensor per level (box first)
            b["t0m"] = {l: act(*lv[l], 2 * TOWER_DEPTH) for l in LEVELS}
            for k, (net, _, _) in enumerate(NETS):
                b["t"][net][0] = {l: b["t0m"][l][..., k * TOWER_DEPTH:(k + 1) * TOWER_DEPTH] for l in LEVELS}
        b["out"] = {"box_net": {l: act(*lv[l], 4 * APL) for l in LEVELS}, "class_net": {l: act(*lv[l], 8) for l in LEVELS}}
        nbn = _lib.lib().mpn_bn_stats_num_parts
        b["stat_lv"] = {l: torch.empty(max(ops.conv_num_parts(N, *lv[l], 3), ops.conv_num_parts(N, *lv[l], 1), nbn(N * lv[l][0] * lv[l][1])) * 2 * DEPTH,
                                       dtype=torch.float32, device=dev) for l in LEVELS}
        b["stat_pre7"] = torch.empty(nbn(N * lv[6][0] * lv[6][1]) * 2 * DEPTH, dtype=torch.float32, device=dev)
        # batched finalizes: one table per stage
        cnt = {l: N * lv[l][0] * lv[l][1] for l in LEVELS}
        # rows the producing kernel writes (ops.conv_stats_rows: one per block for the persistent 3x3 kernel, one per tile otherwise)
        def rows3(l, cin, cout):
            return ops.conv_stats_rows(N, lv[l][0], lv[l][1], cin, cout, 3, self.dtype)
        fin = {}
        rows1 = {l: ops.conv_num_parts(N, *lv[l], 1) for l in (6, 7)}     # p6 / p7 come out of the 1x1 kernel
        fin["p345"] = ops.BnFinalizeBatch([(self.p_bn[l], b["stat_lv"][l], rows3(l, DEPTH, DEPTH), cnt[l]) for l in (3, 4, 5)], dev)
        # the raw p6 feeds TWO batch-norms (p6_batch_norm and fpn/pre_p7_bn): the same partial sums, two finalizes
        fin["p6"] = ops.BnFinalizeBatch([(bn, b["stat_lv"][6], rows1[6], cnt[6]) for bn in (self.p_bn[6], self.pre_p7_bn)], dev)
        fin["p7"] = ops.BnFinalizeBatch([(self.p_bn[7], b["stat_lv"][7], rows1[7], cnt[7])], dev)
        for net, _, _ in NETS:
            for i in range(4):
                cin_i = DEPTH if i == 0 else TOWER_DEPTH
                if not (i == 0 and self.merge_tower0):
                    fin[(net, i)] = ops.BnFinalizeBatch([(self.tower_bn[net][i][l], b["stat_lv"][l], rows3(l, cin_i, TOWER_DEPTH), cnt[l]) for l in LEVELS], dev)
                # batch_norm_0..2 are reduced inside the data gradient of the tower convolution above them (conv rows, raw x),
                # batch_norm_3 inside the data gradient of the output convolution (the tiled kernel: 8 / 24 -> 64 channels)
                if (i < 3 and self._fused_conv_bn()) or (i == 3 and self._fused_out_bn(net)):
                    k_i = TOWER_DEPTH if i < 3 else self.out_conv[net].cout      # channels of the gradient that data gradient reads
                    fin[("d", net, i)] = ops.BnBwdFinalizeBatch([(self.tower_bn[net][i][l], b["stat_lv"][l], rows3(l, k_i, TOWER_DEPTH), cnt[l], True) for l in LEVELS], dev)
                else:
                    fin[("d", net, i)] = ops.BnBwdFinalizeBatch([(self.tower_bn[net][i][l], b["stat_lv"][l], nbn(cnt[l]), cnt[l]) for l in LEVELS], dev)
        if self.merge_tower0:
            fin[("m", 0)] = ops.BnFinalizeBatch([(self.bn0m[l], b["stat_lv"][l], rows3(l, DEPTH, 2 * TOWER_DEPTH), cnt[l]) for l in LEVELS], dev)
        if self._fused_p_bn():     # p{l}_batch_norm is reduced inside the merged first-layer data gradient (conv rows, raw x)
            fin["dp"] = ops.BnBwdFinalizeBatch([(self.p_bn[l], b["stat_lv"][l], rows3(l, 2 * TOWER_DEPTH, DEPTH), cnt[l], True) for l in LEVELS], dev)
        else:
            fin["dp"] = ops.BnBwdFinalizeBatch([(self.p_bn[l], b["stat_lv"][l], nbn(cnt[l]), cnt[l]) for l in LEVELS], dev)
        b["fin"] = fin
        b["levels_hw"] = ((ctypes.c_int * 5)(*[lv[l][0] for l in LEVELS]), (ctypes.c_int * 5)(*[lv[l][1] for l in LEVELS]))
        b["losses"] = torch.zeros(4, dtype=torch.float32, device=dev)
        b["loss_sums"] = torch.zeros(32, dtype=torch.float32, device=dev)
        b["loss_part"] = torch.empty(_lib.lib().mpn_retina_loss_num_parts(N, b["A"]) * 32, dtype=torch.float32, device=dev)
        b["matches"] = torch.empty((N, b["A"]), dtype=torch.int32, device=dev)
        b["targets"] = torch.empty((N, b["A"], 4), dtype=torch.float32, device=dev)
        b["num_matched"] = torch.zeros(1, dtype=torch.int32, device=dev)
        self._bufs[key] = b
        return b

    def _grad_buffers(self, b):
        if "g" in b:
            return b["g"]
        N = b["shape"][0]
        dt, dev, lv = self.dtype, self.device, b["lv"]
        g = {"out": {net: {l: torch.empty_like(b["out"][net][l]) for l in LEVELS} for net, _, _ in NETS},
             "t": {net: [{l: torch.empty_like(b["t"][net][i][l]) for l in LEVELS} for i in range(4)] for net, _, _ in NETS},
             "pn": {net: {l: torch.empty_like(b["p"][l]) for l in LEVELS} for net, _, _ in NETS[:1 if self.merge_tower0 else 2]},
             "x": {l: torch.empty_like(b["x"][l]) for l in (3, 4, 5)},
             "patches7": torch.empty_like(b["patches7"]),
             "pre7": torch.empty_like(b["p"][6])}
        if self.merge_tower0:      # the gradients of the towers' first raw outputs: channel slices of one tensor per level
            g["t0m"] = {l: torch.empty_like(b["t0m"][l]) for l in LEVELS}
            for k, (net, _, _) in enumerate(NETS):
                g["t"][net][0] = {l: g["t0m"][l][..., k * TOWER_DEPTH:(k + 1) * TOWER_DEPTH] for l in LEVELS}
        # weight-gradient slabs: a conv shared by the five levels owns five consecutive regions -> ONE reduction job
        sites = []   # (conv, [(key, nparts)], n)
        merged0 = [self.tower[net][0] for net, _, _ in NETS] if self.merge_tower0 else []
        for c in [c for c in self._convs if c not in merged0] + ([self.tower0m] if self.merge_tower0 else []):
            n = c.src.numel()
            if c is self.tower0m or c in self.tower["box_net"] or c in self.tower["class_net"] or c in self.out_conv.values():
                # (the five levels' weight gradients come from ONE grid: ops.conv_bwd_weight_grouped)
                nps = ops.conv_wgrad_grouped_num_parts(N, [lv[l] for l in LEVELS], c.cin, c.cout, 3, dt)
                parts = [((c.name, l), np_) for l, np_ in zip(LEVELS, nps)]
            else:
                l = int(c.name.split("/")[1].replace("lateral", "").replace("p", ""))
                parts = [((c.name, l), ops.conv_wgrad_num_parts(N, *lv[l], c.cin, c.cout, c.ksize, dt))]
            sites.append((c, parts, n))
        total = sum(sum(np_ for _, np_ in parts) * n for _, parts, n in sites)
        slab = torch.empty(total, dtype=torch.float32, device=dev)
        g["slab"], jobs, off = {}, [], 0
        for c, parts, n in sites:
            first = off
            for key, np_ in parts:
                g["slab"][key] = slab[off:off + np_ * n]
                off += np_ * n
            out = c.dsrc if c is self.tower0m else (c.dpad if c.pad is not None else c.dw)
            jobs.append((slab[first:off], sum(np_ for _, np_ in parts), n, out.view(-1)))
        g["reducer"] = ops.SlabReducer(jobs, dev)
        b["g"] = g
        return g

    # ------------------------------------------------------------------ forward
    def forward(self, images, is_training):
        """images [N,H,W,3] f32 in [0,1] (or uint8). Fills the per-level raw outputs b['out'][net][l]; returns the buffer set."""
        N, H, W, _ = images.shape
        b = self._buffers(N, H, W)
        if b["bb"] is None:
            b["bb"] = self.backbone._buffers(N, H, W, head=False)
        feats = self.backbone.backbone_forward(images, False, b["bb"])        # frozen: moving statistics
        return self.head_forward(feats, b, is_training, images)

    def head_forward(self, feats, b, is_training, images=None):
        """RetinaNet.__init__ (retinanet.py:14-58) on backbone features {'c3','c4','c5': (raw NHWC tensor, Affine)}."""
        if not is_training:
            if not (self.cache_inference_affine and self._infer_clean):      # (45 small launches)
                for bn in self.all_bn:
                    ops.bn_inference_affine(bn)
                self._infer_clean = True
        else:
            self.mark_variables_changed()   # (the finalizes move the statistics and overwrite the affines)
        fin, spl = b["fin"], b["stat_lv"]
        st = (lambda l: spl[l]) if is_training else (lambda l: None)
        prev = None
        for l in (5, 4, 3):
            raw, aff = feats[f"c{l}"]
            ops.conv_fwd(raw, self.lateral[l].packed.fwd, DEPTH, 1, aff, out=b["x"][l], up_res=prev)                 # fpn.py:38,50-51
            prev = b["x"][l]
        ops.conv_fwd_grouped([b["x"][l] for l in (3, 4, 5)], [self.pconv[l].packed.fwd for l in (3, 4, 5)], DEPTH, 3, [None] * 3,
                             [b["p"][l] for l in (3, 4, 5)], [st(l) for l in (3, 4, 5)])                             # fpn.py:39,52
        raw5, aff5 = feats["c5"]
        if not is_training and self._p6_skinny(b):
            self._p6_split_k(raw5, aff5, b)                                                                           # fpn.py:43
        else:
            ops.patchify3x3s2(raw5, b["patches6"], aff5)
            ops.conv_fwd(b["patches6"], self.pconv[6].packed.fwd, DEPTH, 1, None, out=b["p"][6], stats_part=st(6))   # fpn.py:43
        if is_training:
            fin["p345"].run()
            fin["p6"].run()                                                                                           # p6_batch_norm and pre_p7_bn
        ops.patchify3x3s2(b["p"][6], b["patches7"], self.pre_p7_bn.affine)                                            # fpn.py:44
        ops.conv_fwd(b["patches7"], self.pconv[7].packed.fwd, DEPTH, 1, None, out=b["p"][7], stats_part=st(7))       # fpn.py:45
        if is_training:
            fin["p7"].run()
        if self.merge_tower0:       # conv3x3_0 of both towers in one launch: box channels 0..63, class channels 64..127
            ops.conv_fwd_grouped([b["p"][l] for l in LEVELS], [self.tower0m.packed.fwd] * 5, 2 * TOWER_DEPTH, 3,
                                 [self.p_bn[l].affine for l in LEVELS], [b["t0m"][l] for l in LEVELS], [st(l) for l in LEVELS])
            if is_training:
                fin[("m", 0)].run()
        for net, _, cout in NETS:
            xs, affs = [b["p"][l] for l in LEVELS], [self.p_bn[l].affine for l in LEVELS]                            # retinanet.py:29-32
            for i in range(4):                                                                                        # box_predictor.py:101-103
                if i == 0 and self.merge_tower0:
                    xs, affs = [b["t"][net][0][l] for l in LEVELS], [self.tower_bn[net][0][l].affine for l in LEVELS]
                    continue
                c = self.tower[net][i]
                outs = [b["t"][net][i][l] for l in LEVELS]
                ops.conv_fwd_grouped(xs, [c.packed.fwd] * 5, TOWER_DEPTH, 3, affs, outs, [st(l) for l in LEVELS])
                if is_training:
                    fin[(net, i)].run()
                xs, affs = outs, [self.tower_bn[net][i][l].affine for l in LEVELS]
            oc = self.out_conv[net]
            ops.conv_fwd_grouped(xs, [oc.packed.fwd] * 5, oc.cout, 3, affs, [b["out"][net][l] for l in LEVELS], [None] * 5)
        self._last = (b, feats, images)
        return b

    # ---- fpn/p6 at small batch (inference): a 9 * 1024-deep contraction over a few dozen pixels - as a 1x1 convolution that is ONE
    # 128-pixel tile on ONE CU (167 us at 640 x 640, batch 1). Same trick as the PRN's K = 34 272 contractions: the split-K
    # "weight gradient" of a 1x1 convolution whose pixel axis is K, operands K-major (X^T and the HWIO matrix as it is stored).
    def _p6_skinny(self, b):
        n, (h6, w6) = b["shape"][0], b["lv"][6]
        return self.dtype != torch.float32 and n * h6 * w6 <= 256

    def _p6_split_k(self, raw5, aff5, b):
        c6 = self.pconv[6]
        n, (h6, w6) = b["shape"][0], b["lv"][6]
        M, K, dc, f32c = n * h6 * w6, c6.cin, _lib.dtype_code(self.dtype), _lib.dtype_code(torch.float32)
        s = b.get("p6_sk")
        if s is None:
            Mp = (M + 7) // 8 * 8
            dev = self.device
            s = b["p6_sk"] = {"Mp": Mp, "xpad": torch.zeros((Mp, K), dtype=self.dtype, device=dev),
                              "xt": torch.empty((K, Mp), dtype=self.dtype, device=dev),
                              "out": torch.empty((Mp, DEPTH), dtype=torch.float32, device=dev),
                              "wop": torch.empty((K, DEPTH), dtype=self.dtype, device=dev), "wver": -1}
            nparts = ops.conv_wgrad_num_parts(1, 1, K, Mp, DEPTH, 1, self.dtype)
            s["slab"] = torch.empty(nparts * Mp * DEPTH, dtype=torch.float32, device=dev)
        # the HWIO matrix [9 * C5, 128] in the storage type: refreshed on every pass, or (owner opted in, as for the affines)
        # once per repack of the variables
        if not self.cache_inference_affine or s["wver"] != self._wversion:
            call("mpn_cast", ptr(c6.src), f32c, ptr(s["wop"]), dc, K * DEPTH, stream_ptr())
            s["wver"] = self._wversion
        Mp = s["Mp"]
        ops.patchify3x3s2(raw5, s["xpad"][:M].view(n, h6, w6, K), aff5)                 # rows M..Mp stay zero
        call("mpn_transpose_cast", ptr(s["xpad"]), dc, ptr(s["xt"]), dc, Mp, K, stream_ptr())
        ops.conv_bwd_weight(s["xt"].view(1, 1, K, Mp), s["wop"].view(1, 1, K, DEPTH), 1, None, s["out"].view(1, 1, Mp, DEPTH), s["slab"])
        call("mpn_cast", ptr(s["out"]), f32c, ptr(b["p"][6]), dc, M * DEPTH, stream_ptr())

    def raw_predictions(self, b):
        """{'encoded_boxes': [N,A,4], 'class_predictions': [N,A]} f32 in the reference's anchor order (box_predictor.py:55-90),
        biases added - a host-friendly view for tests; the loss and NMS kernels read the per-level tensors directly."""
        N = b["shape"][0]
        enc = torch.cat([(b["out"]["box_net"][l].float() + self.out_bias["box_net"]).reshape(N, -1, 4) for l in LEVELS], dim=1)
        cls = torch.cat([(b["out"]["class_net"][l][..., :APL].float() + self.out_bias["class_net"]).reshape(N, -1) for l in LEVELS], dim=1)
        return {"encoded_boxes": enc, "class_predictions": cls}

    # ------------------------------------------------------------------ targets, losses
    def _level_ptrs(self, d_box, d_cls):
        PA = ctypes.c_void_p * 5
        return PA(*[ptr(d_cls[l]) for l in LEVELS]), PA(*[ptr(d_box[l]) for l in LEVELS])

    def create_targets(self, groundtruth, b=None):
        """retinanet.py:146-166: groundtruth {'boxes': f32 [N,max,4] normalised, 'num_boxes': int32 [N]} -> matches, targets.
        b: the buffer set of the forward pass the targets belong to (default: the most recent head_forward)."""
        b = self._last[0] if b is None else b
        boxes, nb = groundtruth["boxes"], groundtruth["num_boxes"]
        N, maxn = boxes.shape[0], boxes.shape[1]
        if boxes.dtype != torch.float32 or not boxes.is_contiguous() or nb.dtype != torch.int32:
            raise ValueError("groundtruth boxes must be contiguous float32 [N,max,4], num_boxes int32 [N]")
        need = _lib.lib().mpn_retina_match_workspace_bytes(N, maxn)
        ws = b.get("match_ws")
        if ws is None or ws.numel() < need:
            ws = b["match_ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
        call("mpn_retina_match", ptr(b["anchors"]), ptr(boxes), ptr(nb), N, b["A"], maxn, POSITIVES_THRESHOLD, NEGATIVES_THRESHOLD,
             ptr(b["matches"]), ptr(b["targets"]), ptr(b["num_matched"]), ptr(ws), ws.numel(), stream_ptr())
        return b["targets"], b["matches"]

    def compute_losses(self, params, with_grad=True, b=None):
        """retinanet.py:86-144 + person_detector_model.py:33-45. Call after forward + create_targets. Returns f32[4]
        (LOSS_NAMES) on the device; with_grad also fills the gradients of the raw tower outputs and the bias gradients.
        b: the buffer set of the forward pass to score (default: the most recent head_forward)."""
        b = self._last[0] if b is None else b
        N = b["shape"][0]
        g = self._grad_buffers(b) if with_grad else None
        lg, bx = self._level_ptrs(b["out"]["box_net"], b["out"]["class_net"])
        if with_grad:
            dlg, dbx = self._level_ptrs(g["out"]["box_net"], g["out"]["class_net"])
        else:
            dlg = dbx = None
        hs, ws_ = b["levels_hw"]
        lw, cw = float(params.get("localization_loss_weight", 1.0)), float(params.get("classification_loss_weight", 1.0))
        call("mpn_retina_loss", lg, bx, dlg, dbx, hs, ws_, _lib.dtype_code(self.dtype), ptr(self.out_bias["class_net"]),
             ptr(self.out_bias["box_net"]), ptr(b["matches"]), ptr(b["targets"]), ptr(b["num_matched"]), N,
             float(params.get("gamma", 2.0)), float(params.get("alpha", 0.25)), lw, cw, ptr(b["loss_part"]), stream_ptr())
        ops.reduce_partials(b["loss_part"], b["loss_part"].numel() // 32, 32, b["loss_sums"])
        losses = b["losses"]
        losses[2:3].zero_()           # (ATen's fill KERNEL - not a memset node, which can race inside a replayed graph: DESIGN 2 - and not an
                                      #  element assignment from a Python scalar, which is a host copy and not capturable)
        wd = float(params.get("weight_decay", 0.0))
        if wd > 0.0:   # add_weight_decay (keypoints_model.py:129-138) sees EVERY kernel, the frozen backbone's included
            if self._l2 is None:
                ts = [w for k, w in self.vars.items() if "kernel" in k]
                ts += [w for k, w in self.backbone.vars.items()
                       if k.startswith("MobilenetV1/") and "weights" in k and "depthwise_weights" not in k]
                self._l2 = ops.L2LossBatch(ts)
            self._l2.run(wd, losses[2:3])
        # normaliser 1 / max(num_matched, 1), the weighted total and the output convolutions' bias gradients: one launch
        call("mpn_retina_loss_finalize", ptr(b["loss_sums"]), ptr(b["num_matched"]), lw, cw, ptr(losses),
             ptr(self.out_dbias["class_net"]) if with_grad else None, ptr(self.out_dbias["box_net"]) if with_grad else None, stream_ptr())
        return losses

    # ------------------------------------------------------------------ backward
    def backward(self, weight_decay=0.0):
        """Gradients of the total loss w.r.t. every head variable -> self.grad (after compute_losses). The backbone is frozen
        (person_detector_model.py:71-72): no gradient flows into c3..c5."""
        b, feats, _ = self._last
        g = self._grad_buffers(b)
        N, lv, spl, fin, slab = b["shape"][0], b["lv"], b["stat_lv"], b["fin"], g["slab"]
        sps = [spl[l] for l in LEVELS]
        none5 = [None] * 5
        for net, _, _ in NETS:
            oc = self.out_conv[net]
            bn3 = self.tower_bn[net][3]
            ops.conv_bwd_weight_grouped([b["t"][net][3][l] for l in LEVELS], [g["out"][net][l] for l in LEVELS], 3,
                                        [bn3[l].affine for l in LEVELS], [slab[(oc.name, l)] for l in LEVELS])
            fused, fused_out = self._fused_conv_bn(), self._fused_out_bn(net)
            if fused_out:
                ops.conv_bwd_data_bn_grouped([g["out"][net][l] for l in LEVELS], [oc.packed.bwd] * 5, TOWER_DEPTH, [bn3[l] for l in LEVELS],
                                             [b["t"][net][3][l] for l in LEVELS], [g["t"][net][3][l] for l in LEVELS], sps)
            else:
                ops.conv_fwd_grouped([g["out"][net][l] for l in LEVELS], [oc.packed.bwd] * 5, TOWER_DEPTH, 3, none5,
                                     [g["t"][net][3][l] for l in LEVELS], none5)
            for i in (3, 2, 1, 0):
                bns = [self.tower_bn[net][i][l] for l in LEVELS]
                dAs, xs = [g["t"][net][i][l] for l in LEVELS], [b["t"][net][i][l] for l in LEVELS]
                if not ((fused and i < 3) or (fused_out and i == 3)):      # (else: reduced by the data gradient above)
                    ops.bn_bwd_reduce_grouped(bns, dAs, xs, sps)
                fin[("d", net, i)].run()
                if i == 0 and self.merge_tower0:
                    continue         # (the rest of the first layer: both towers at once, below)
                ops.bn_bwd_apply_grouped(bns, dAs, xs)
                c = self.tower[net][i]
                if i > 0:
                    xin, ain = [b["t"][net][i - 1][l] for l in LEVELS], [self.tower_bn[net][i - 1][l].affine for l in LEVELS]
                    dst = [g["t"][net][i - 1][l] for l in LEVELS]
                else:
                    xin, ain = [b["p"][l] for l in LEVELS], [self.p_bn[l].affine for l in LEVELS]
                    dst = [g["pn"][net][l] for l in LEVELS]
                ops.conv_bwd_weight_grouped(xin, dAs, 3, ain, [slab[(c.name, l)] for l in LEVELS])
                if fused and i > 0:   # the data gradient also reduces for batch_norm_{i-1} (and writes its gradient masked)
                    ops.conv_bwd_data_bn_grouped(dAs, [c.packed.bwd] * 5, c.cin, [self.tower_bn[net][i - 1][l] for l in LEVELS], xin, dst, sps)
                else:
                    ops.conv_fwd_grouped(dAs, [c.packed.bwd] * 5, c.cin, 3, none5, dst, none5)
        # the two towers meet at act(bn(p_l)): sum, then through p{l}_batch_norm
        gp = [g["pn"]["box_net"][l] for l in LEVELS]
        pbns, bp = [self.p_bn[l] for l in LEVELS], [b["p"][l] for l in LEVELS]
        if self.merge_tower0:
            gm, tm = [g["t0m"][l] for l in LEVELS], [b["t0m"][l] for l in LEVELS]
            ops.bn_bwd_apply_grouped([self.bn0m[l] for l in LEVELS], gm, tm)           # both towers' batch_norm_0: one 128-channel pass
            pa = [self.p_bn[l].affine for l in LEVELS]
            ops.conv_bwd_weight_grouped(bp, gm, 3, pa, [slab[(self.tower0m.name, l)] for l in LEVELS])      # both towers' conv3x3_0
            # ONE 128 -> 128 data gradient: the contraction over the merged channels is the sum of the two towers' gradients
            if self._fused_p_bn():
                ops.conv_bwd_data_bn_grouped(gm, [self.tower0m.packed.bwd] * 5, DEPTH, pbns, bp, gp, sps)
            else:
                ops.conv_fwd_grouped(gm, [self.tower0m.packed.bwd] * 5, DEPTH, 3, none5, gp, none5)
                ops.bn_bwd_reduce_grouped(pbns, gp, bp, sps)
        else:
            for l in LEVELS:
                ops.add_inplace(g["pn"]["box_net"][l], g["pn"]["class_net"][l])
            ops.bn_bwd_reduce_grouped(pbns, gp, bp, sps)
        fin["dp"].run()
        ops.bn_bwd_apply_grouped(pbns, gp, bp)
        gpl = {l: g["pn"]["box_net"][l] for l in LEVELS}              # gradient w.r.t. the raw p_l
        # ---- coarse branch: p7 = conv(patches(act(pre_p7_bn(p6)))), p6 = conv(patches(c5))
        c7, c6 = self.pconv[7], self.pconv[6]
        ops.conv_bwd_weight(b["patches7"], gpl[7], 1, None, None, slab[(c7.name, 7)], reduce=False)
        ops.conv_fwd(gpl[7], c7.packed.bwd, c7.cin, 1, None, out=g["patches7"])
        ops.unpatchify3x3s2(g["patches7"], g["pre7"])
        ops.bn_backward(self.pre_p7_bn, g["pre7"], b["p"][6], b["stat_pre7"])
        ops.add_inplace(gpl[6], g["pre7"])
        ops.conv_bwd_weight(b["patches6"], gpl[6], 1, None, None, slab[(c6.name, 6)], reduce=False)
        # ---- top-down path reversed
        for l in (3, 4, 5):
            ops.conv_bwd_weight(b["x"][l], gpl[l], 3, None, None, slab[(self.pconv[l].name, l)], reduce=False)
        ops.conv_fwd_grouped([gpl[l] for l in (3, 4, 5)], [self.pconv[l].packed.bwd for l in (3, 4, 5)], DEPTH, 3, [None] * 3,
                             [g["x"][l] for l in (3, 4, 5)], [None] * 3)
        for l in (3, 4, 5):
            if l > 3:
                ops.sumpool2x2(g["x"][l - 1], g["x"][l], accumulate=True)
            raw, aff = feats[f"c{l}"]
            ops.conv_bwd_weight(raw, g["x"][l], 1, aff, None, slab[(self.lateral[l].name, l)], reduce=False)
        g["reducer"].run()
        if self.merge_tower0:
            self.tower0m.split_grad()
        oc = self.out_conv["class_net"]
        oc.dw.copy_(oc.dpad[..., :oc.w.shape[3]])                     # drop the two padding columns
        if weight_decay > 0.0:
            if self._wd is None:
                ks = [k for k in self.vars if "kernel" in k]
                self._wd = ops.AxpyBatch([self.vars[k] for k in ks], [self.grads[k] for k in ks])
            self._wd.run(weight_decay)

    def optimizer_step(self, initial_learning_rate, num_steps, grad_scale=1.0):
        """person_detector_model.py:59-75: cosine decay, TF-Adam (NO gradient clipping in this model), head variables only."""
        ops.adam_prepare(self.global_step, self.hyper, initial_learning_rate, num_steps)
        ops.adam_step(self.theta, self.grad, self.adam_m, self.adam_v, self.hyper, grad_scale=grad_scale, clip=float("inf"))
        self.mark_variables_changed()
        self.repack_weights()

    def train_step(self, images, groundtruth, params):
        """One TRAIN step of person_detector_model.model_fn. Returns the f32[4] losses tensor (LOSS_NAMES)."""
        self.forward(images, True)
        self.create_targets(groundtruth)
        losses = self.compute_losses(params)
        self.backward(float(params.get("weight_decay", 0.0)))
        self.optimizer_step(float(params["initial_learning_rate"]), int(params["num_steps"]))
        return losses

    # ------------------------------------------------------------------ inference
    def predict(self, images, score_threshold=0.05, iou_threshold=0.5, max_detections=25):
        """retinanet.py:60-84: {'boxes': [N,max,4], 'scores': [N,max], 'num_boxes': [N]} (device tensors)."""
        return self.nms(self.forward(images, False), score_threshold, iou_threshold, max_detections)

    def nms(self, b, score_threshold=0.05, iou_threshold=0.5, max_detections=25):
        """get_predictions on the raw outputs of the last forward over buffer set `b`. Besides the reference's three keys the dict
        holds 'overflow': the call's int32[1] overflow word (a view into the workspace, 0 after a good call); whoever reads the
        outputs on the host passes the dict through `check_nms` first."""
        N, A = b["shape"][0], b["A"]
        need = _lib.lib().mpn_retina_nms_workspace_bytes(N, A)
        ws = b.get("nms_ws")
        if ws is None or ws.numel() < need:
            ws = b["nms_ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
        boxes = torch.empty((N, max_detections, 4), dtype=torch.float32, device=self.device)
        scores = torch.empty((N, max_detections), dtype=torch.float32, device=self.device)
        num = torch.empty((N,), dtype=torch.int32, device=self.device)
        lg, bx = self._level_ptrs(b["out"]["box_net"], b["out"]["class_net"])
        hs, ws_ = b["levels_hw"]
        call("mpn_retina_nms", lg, bx, hs, ws_, _lib.dtype_code(self.dtype), ptr(self.out_bias["class_net"]), ptr(self.out_bias["box_net"]),
             ptr(b["anchors"]), N, float(score_threshold), float(iou_threshold), int(max_detections), ptr(boxes), ptr(scores), ptr(num),
             ptr(ws), ws.numel(), stream_ptr())
        off = _lib.lib().mpn_retina_nms_overflow_offset(N, A)
        return {"boxes": boxes, "scores": scores, "num_boxes": num, "overflow": ws[off:off + 4].view(torch.int32)}

    @staticmethod
    def check_nms(pred):
        """Raises if the NMS call behind `pred` ran out of list slots (MPN_ERR_WORKSPACE: a candidate counter that did not start
        at zero). Synchronises on the overflow word; a graph replay has no other way to report it."""
        if int(pred["overflow"].item()) != 0:
            raise RuntimeError("mpn_retina_nms: a candidate list overflowed its workspace (MPN_ERR_WORKSPACE); the detections are incomplete")
        return pred
