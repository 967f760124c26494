"""KeypointNet: the device-resident keypoint model (MobileNet-v1 -> FPN -> keypoint subnet),
its losses, backward pass and optimizer step, orchestrating the HIP kernels of libmpn_hip.so.

Mirrors, layer for layer, the reference graph built by keypoints_model.py:14-21,
detector/backbones/mobilenet_v1.py, detector/fpn.py and detector/keypoint_subnet.py; variable
names and HWIO shapes are the reference's (so a TF checkpoint exported to .npz loads as is).

Layout decisions (MI355X-first):
  * activations NHWC in bf16 (throughput) or f32 (parity), accumulation always f32;
  * every conv writes its RAW output once; batch-norm is a per-channel affine that the NEXT
    kernel applies while loading, so normalised activations never round-trip through HBM;
  * all trainable parameters live in ONE flat f32 arena (+ grad / m / v arenas of the same
    layout): Adam is a single fused pass and the data-parallel all-reduce sees one buffer;
  * all buffers are allocated once per input shape, so a whole train step is hipGraph-capturable.
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, ops
from ._lib import ACT_NONE, ACT_RELU, ACT_RELU6

NUM_KEYPOINTS = 17   # detector/constants.py:10
DOWNSAMPLE = 4       # detector/constants.py:13
DIVISOR = 128        # detector/constants.py:4
DEPTH = 128          # detector/keypoint_subnet.py:7
STRIDES_AND_FILTERS = [(1, 64), (2, 128), (1, 128), (2, 256), (1, 256), (2, 512), (1, 512), (1, 512),
                       (1, 512), (1, 512), (1, 512), (2, 1024), (1, 1024)]   # mobilenet_v1.py:59-65
FEATURE_BLOCKS = {3: "c2", 5: "c3", 11: "c4", 13: "c5"}                     # mobilenet_v1.py:76-79


def depth(x, depth_multiplier):
    return max(int(x * depth_multiplier), 8)   # mobilenet_v1.py:25-27


def variable_shapes(depth_multiplier=1.0):
    """Ordered {reference variable name: shape} (trainables + batch-norm moving statistics)."""
    s = OrderedDict()

    def bn(prefix, c):
        for n in ("gamma", "beta", "moving_mean", "moving_variance"):
            s[f"{prefix}/{n}"] = (c,)

    c = depth(32, depth_multiplier)
    s["MobilenetV1/Conv2d_0/weights"] = (3, 3, 3, c)
    bn("MobilenetV1/Conv2d_0/BatchNorm", c)
    for i, (_, f) in enumerate(STRIDES_AND_FILTERS, 1):
        s[f"MobilenetV1/Conv2d_{i}_depthwise/depthwise_weights"] = (3, 3, c, 1)
        bn(f"MobilenetV1/Conv2d_{i}_depthwise/BatchNorm", c)
        f = depth(f, depth_multiplier)
        s[f"MobilenetV1/Conv2d_{i}_pointwise/weights"] = (1, 1, c, f)
        bn(f"MobilenetV1/Conv2d_{i}_pointwise/BatchNorm", f)
        c = f
    feat = {2: depth(128, depth_multiplier), 3: depth(256, depth_multiplier),
            4: depth(512, depth_multiplier), 5: depth(1024, depth_multiplier)}
    s["keypoint_fpn/lateral5/kernel"] = (1, 1, feat[5], DEPTH)
    s["keypoint_fpn/p5/kernel"] = (3, 3, DEPTH, DEPTH)
    for i in (4, 3, 2):
        s[f"keypoint_fpn/lateral{i}/kernel"] = (1, 1, feat[i], DEPTH)
        s[f"keypoint_fpn/p{i}/kernel"] = (3, 3, DEPTH, DEPTH)
    for l in (2, 3, 4, 5):
        bn(f"p{l}_batch_norm", DEPTH)
    for l in (2, 3, 4, 5):
        s[f"phi_subnet_{l}/conv1/kernel"] = (3, 3, DEPTH, DEPTH)
        bn(f"phi_subnet_{l}/bn1", DEPTH)
        s[f"phi_subnet_{l}/conv2/kernel"] = (3, 3, DEPTH, DEPTH)
        bn(f"phi_subnet_{l}/bn2", DEPTH)
    s["final_conv3x3/kernel"] = (3, 3, 4 * DEPTH, 64)
    bn("final_bn", 64)
    s["heatmaps/kernel"] = (1, 1, 64, NUM_KEYPOINTS + 1)
    s["heatmaps/bias"] = (NUM_KEYPOINTS + 1,)
    return s


def internal_shapes(depth_multiplier=1.0):
    """(shapes as the arena holds them, {name: (axis, reference size)} of the padded ones). The stem's matrix-core kernels
    work on 16-channel blocks; mobilenet_v1.py:25-27 gives the stem max(int(32 m), 8) channels - 24 at depth_multiplier 0.75,
    8 at 0.25. The variables that carry that width (Conv2d_0 and its batch-norm, Conv2d_1_depthwise and its batch-norm, the
    input side of Conv2d_1_pointwise) are PADDED to the next multiple of 16 inside the arena. The pad is all zeros (gamma
    and beta too), which is a fixed point of the step: zero stem kernels give a zero raw output, x-hat = 0, activation 0; every
    gradient of a pad element is a product with one of those zeros (dx carries gamma * invstd = 0, the masks are closed at
    0 < 0), so Adam's slots and updates stay exactly 0. state_dict / checkpoints see the reference shapes only."""
    ref = variable_shapes(depth_multiplier)
    c0 = depth(32, depth_multiplier)
    c0p = (c0 + 15) // 16 * 16
    if c0p == c0:
        return ref, {}
    shapes, pads = OrderedDict(), {}
    for name, shape in ref.items():
        axis = None
        if name.startswith("MobilenetV1/Conv2d_0/") or name.startswith("MobilenetV1/Conv2d_1_depthwise/BatchNorm/"):
            axis = len(shape) - 1
        elif name in ("MobilenetV1/Conv2d_1_depthwise/depthwise_weights", "MobilenetV1/Conv2d_1_pointwise/weights"):
            axis = 2
        if axis is not None:
            assert shape[axis] == c0
            pads[name] = (axis, c0)
            shape = tuple(c0p if i == axis else d for i, d in enumerate(shape))
        shapes[name] = shape
    return shapes, pads


def is_trainable(name):
    return not (name.endswith("moving_mean") or name.endswith("moving_variance"))


def initial_values(seed=0, depth_multiplier=1.0):
    """Seeded initial values following the reference's initialiser families: variance scaling for conv
    kernels (layer_utils.py:37), N(0,1e-4) `heatmaps/kernel` and bias -log(99)x17 + 0
    (keypoint_subnet.py:41-53), gamma 1 / beta 0 / moving (0,1)."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in variable_shapes(depth_multiplier).items():
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
        out[name] = v.astype(np.float32)
    return out


DP_DEEP_FROM_BLOCK = 7    # data-parallel exchange: backbone blocks 7..13 (2.9 M of the backbone's 3.2 M parameters, the SHORT end of
                          # its backward pass: 32x32 and 16x16 maps) travel while blocks 6..1 and the stem (0.3 M parameters on the
                          # large maps, most of the backward time) are still computed


def backbone_deep_begin_of(arena, first_block=DP_DEEP_FROM_BLOCK):
    """Offset in a flat arena of the trainable variables where backbone block `first_block` begins (the arena is laid out in
    network order: Conv2d_0, Conv2d_1_depthwise, ... Conv2d_13_pointwise, then the head end)."""
    return arena.offsets[f"MobilenetV1/Conv2d_{first_block}_depthwise/depthwise_weights"][0]


class _Arena:
    """Flat f32 device arena with named, 16-byte aligned views."""

    def __init__(self, shapes, device):
        self.offsets = OrderedDict()
        off = 0
        for name, shape in shapes.items():
            n = int(np.prod(shape))
            self.offsets[name] = (off, n, tuple(shape))
            off += (n + 3) // 4 * 4
        self.size = off
        self.device = device

    def new(self):
        return torch.zeros(self.size, dtype=torch.float32, device=self.device)

    def views(self, flat):
        return OrderedDict((k, flat[o:o + n].view(shape)) for k, (o, n, shape) in self.offsets.items())


class _Conv:
    """One dense conv: reference variable + packed MFMA operands."""

    def __init__(self, name, w, dw, dtype):
        self.name, self.w, self.dw = name, w, dw
        self.ksize, _, self.cin, self.cout = w.shape
        self.packed = ops.PackedConv(w, dtype)


def backbone_grad_end_of(arena):
    """Offset in a flat arena of the trainable variables where the backbone's (`MobilenetV1/*`) end."""
    end = 0
    for k, (o, n, _) in arena.offsets.items():
        if k.startswith("MobilenetV1/"):
            end = max(end, o + (n + 3) // 4 * 4)
    return end


@ops._lib.device_guarded("_init", "load_state_dict", "repack_weights", "prepare_inference", "forward", "predict",
                         "backbone_forward", "subnet_forward", "compute_losses", "backward", "add_weight_decay_gradients",
                         "add_weight_decay_loss", "optimizer_step")
class KeypointNet:
    def __init__(self, values=None, depth_multiplier=1.0, dtype=torch.bfloat16, device="cuda:0", seed=0):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("dtype must be torch.float32 or torch.bfloat16")
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.dtype, self.device, self.dm = dtype, dev, depth_multiplier
        self._init(values, depth_multiplier, dtype, seed)

    def _init(self, values, depth_multiplier, dtype, seed):
        shapes, self._pads = internal_shapes(depth_multiplier)
        # every batch-norm width a multiple of 16 (the stem's is padded to one inside the arena): the matrix-core kernels are
        # exercised - end to end against the oracle - at such widths only (0.25, 0.5, 0.75, 1.0, ...); 0.375 would hand 24- and
        # 48-channel pointwise layers to them
        for n, s in shapes.items():
            if n.endswith("gamma") and (s[0] % 16 != 0):
                raise ValueError(f"depth_multiplier={depth_multiplier}: {n} has {s[0]} channels; every layer behind the stem "
                                 f"must be a multiple of 16 channels wide (depth multipliers that are multiples of 0.25)")
        self._train_arena = _Arena(OrderedDict((k, v) for k, v in shapes.items() if is_trainable(k)), self.device)
        self._stat_arena = _Arena(OrderedDict((k, v) for k, v in shapes.items() if not is_trainable(k)), self.device)
        self.theta = self._train_arena.new()
        self.grad = self._train_arena.new()
        self.adam_m = self._train_arena.new()
        self.adam_v = self._train_arena.new()
        self.moving = self._stat_arena.new()
        self.vars = self._train_arena.views(self.theta)
        self.grads = self._train_arena.views(self.grad)
        self.stats = self._stat_arena.views(self.moving)
        self.global_step = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.load_state_dict(values if values is not None else initial_values(seed, depth_multiplier))
        self._build_layers()
        self._bufs = {}
        self.overlap_wgrad = False    # opt-in: weight gradients on a second HIP stream (measured 4% SLOWER at bs32: the
                                      # kernels already fill the chip, concurrent ones only contend; see backward)
        self._wstream = None

    # ------------------------------------------------------------------ variables
    def state_dict(self):
        """{reference variable name: numpy array} (HWIO kernels, as in a TF checkpoint)."""
        out = OrderedDict()
        for k, v in list(self.vars.items()) + list(self.stats.items()):
            out[k] = self.unpad(k, v).detach().cpu().numpy().copy()
        return out

    def unpad(self, name, t):
        """The reference-shaped part of an arena view (variable, gradient or Adam slot) of `name` (internal_shapes)."""
        if name in self._pads:
            axis, n = self._pads[name]
            return t.narrow(axis, 0, n)
        return t

    def load_state_dict(self, values, strict=True):
        for k, v in values.items():
            dst = self.vars.get(k, self.stats.get(k))
            if dst is None:
                if strict:
                    raise KeyError(f"unknown variable {k}")
                continue
            v = np.asarray(v, dtype=np.float32)
            if k in self._pads:
                dst.zero_()                      # the pad: zeros, gamma included (see internal_shapes)
                dst = self.unpad(k, dst)
            if tuple(v.shape) != tuple(dst.shape):
                raise ValueError(f"{k}: shape {v.shape} != {tuple(dst.shape)}")
            dst.copy_(torch.from_numpy(v))
        if strict:
            missing = [k for k in list(self.vars) + list(self.stats) if k not in values]
            if missing:
                raise KeyError(f"missing variables: {missing[:5]}...")
        self.mark_variables_changed()
        if hasattr(self, "convs"):
            self.repack_weights()

    def mark_variables_changed(self):
        """Variables or moving statistics changed: cached inference affines are stale and `var_version` moves (users that
        replay captured device work - inference/detector.py - compare it). Every method of this object that changes them
        calls this; so must whoever changes them from outside: a replayed hipGraph of a train step (train.Trainer.step does),
        a direct write into `vars` / `stats`."""
        self.var_version = getattr(self, "var_version", 0) + 1
        self._infer_clean = False

    def _bn(self, prefix, act):
        bn = ops.BNState(self.vars[prefix + "/gamma"], self.vars[prefix + "/beta"], self.stats[prefix + "/moving_mean"],
                         self.stats[prefix + "/moving_variance"], act)
        bn.dgamma, bn.dbeta = self.grads[prefix + "/gamma"], self.grads[prefix + "/beta"]
        bn.name = prefix
        return bn

    def _conv(self, name):
        c = _Conv(name, self.vars[name], self.grads[name], self.dtype)
        self.convs.append(c)
        return c

    def _build_layers(self):
        self.convs = []
        self.stem_w = self.vars["MobilenetV1/Conv2d_0/weights"]
        self.stem_dw = self.grads["MobilenetV1/Conv2d_0/weights"]
        self.stem_bn = self._bn("MobilenetV1/Conv2d_0/BatchNorm", ACT_RELU6)
        self.blocks = []
        for i, (stride, _) in enumerate(STRIDES_AND_FILTERS, 1):
            d, p = f"MobilenetV1/Conv2d_{i}_depthwise", f"MobilenetV1/Conv2d_{i}_pointwise"
            self.blocks.append(dict(
                i=i, stride=stride,
                dw_w=self.vars[d + "/depthwise_weights"], dw_dw=self.grads[d + "/depthwise_weights"],
                dw_bn=self._bn(d + "/BatchNorm", ACT_RELU6),
                pw=self._conv(p + "/weights"), pw_bn=self._bn(p + "/BatchNorm", ACT_RELU6)))
        self.lateral = {l: self._conv(f"keypoint_fpn/lateral{l}/kernel") for l in (5, 4, 3, 2)}
        self.pconv = {l: self._conv(f"keypoint_fpn/p{l}/kernel") for l in (5, 4, 3, 2)}
        self.p_bn = {l: self._bn(f"p{l}_batch_norm", ACT_RELU) for l in (2, 3, 4, 5)}
        self.phi = {}
        for l in (2, 3, 4, 5):
            s = f"phi_subnet_{l}"
            self.phi[l] = dict(conv1=self._conv(s + "/conv1/kernel"), bn1=self._bn(s + "/bn1", ACT_RELU),
                               conv2=self._conv(s + "/conv2/kernel"), bn2=self._bn(s + "/bn2", ACT_RELU))
        self.final_conv = self._conv("final_conv3x3/kernel")
        self.final_bn = self._bn("final_bn", ACT_RELU)
        # The concat tensor (keypoint_subnet.py:37) holds level 2 RAW: phi_subnet_2/conv2 writes its output straight into
        # channels 0..127 (an upsampling factor of 1 is a copy) and the consumers - final_conv3x3 and its weight gradient -
        # apply bn2's affine + ReLU on load like every other consumer of a raw conv output. Channels 128..511 hold the
        # ACTIVATED, bilinearly upsampled levels 3..5 (non-negative: scale 1, shift 0 and the ReLU are the identity there).
        # bn2 of level 2 keeps its scale / shift in the head of the composite vectors, so every finalize updates them in place.
        self.concat_scale = torch.ones(4 * DEPTH, dtype=torch.float32, device=self.device)
        self.concat_shift = torch.zeros(4 * DEPTH, dtype=torch.float32, device=self.device)
        self.phi[2]["bn2"].scale = self.concat_scale[:DEPTH]
        self.phi[2]["bn2"].shift = self.concat_shift[:DEPTH]
        self.concat_affine = ops.Affine(self.concat_scale, self.concat_shift, ACT_RELU)
        self.heat_w, self.heat_b = self.vars["heatmaps/kernel"], self.vars["heatmaps/bias"]
        # `heatmaps/kernel` and `heatmaps/bias` are adjacent in the arena, so the head backward (dW then db,
        # contiguous) writes straight into the gradient arena.
        ok, nk, _ = self._train_arena.offsets["heatmaps/kernel"]
        ob, nb, _ = self._train_arena.offsets["heatmaps/bias"]
        assert ob == ok + nk, "heatmaps/kernel and heatmaps/bias must be adjacent in the arena"
        self._head_grad = self.grad[ok:ob + nb]
        self.fuse_dw_bn = True    # depthwise data gradients also reduce for the batch-norm they feed (mpn_dwconv_bwd_data_bn)
        self.fuse_pw_wide = False # ... also on 128-channel layers (Conv2d_3_pointwise, lateral2): measured equal to the two-pass backward in the step
        self.fuse_pw_apply = True # ... with the layer's own batch-norm apply pass folded in where the kernel takes it (Cin <= 32, Cout <= 64)
        self.fuse_pw_bwd = True   # thin pointwise layers (Cin <= 64, Cout <= 128): weight + data gradient + reduction in one pass (mpn_conv1x1_bwd_fused)
        self.fuse_dw_bwd_s2 = True  # ... and the stride-2 ones (even maps; the FPN lateral's gradient added inside: mpn_dwconv_bwd_fused_s2)
        self.fuse_dw_bwd = True   # stride-1 depthwise layers: data gradient + that reduction + weight gradient in ONE walk (mpn_dwconv_bwd_fused)
        # ... and so do the subnet's 3x3 data gradients (mpn_conv_bwd_data_bn_grouped: bn1 under conv2's, p{l}_batch_norm under
        # conv1's); set before the first backward pass of a shape (the finalize tables are built once)
        self.fuse_conv_bn = True
        # (forming their own INPUT on load - the batch-norm apply pass inside the data gradient, round 5 - was measured slower in the step,
        #  7.55 against 7.43 ms, profiles/r05_apply_on_load.txt, and is gone since round 6: DESIGN's table of negatives)
        self._l2 = None           # the regularisation term's batched launch (add_weight_decay_loss)
        self._wd = None           # ... and its gradient's (add_weight_decay_gradients)
        self.cache_inference_affine = False   # see prepare_inference
        self._infer_clean = False
        # the stem kernel writes its own batch-norm partial sums (mpn_stem_conv_fwd_stats). With the VALU stem kernel this was a
        # wash (the separate 38 us statistics pass left the 134 MB stem output in the memory-side cache for the first depthwise
        # layer); behind the matrix-core kernel it is worth 15 us per step (same-box A/B, round 3)
        self.fuse_stem_stats = True
        self.fuse_lateral_add = True   # ... and add the FPN lateral's gradient into c2..c4 (mpn_dwconv_bwd_data_add)
        self._build_pack_table()   # (outside any graph capture: it copies a small table to the device)
        self.all_bn = [self.stem_bn] + [b[k] for b in self.blocks for k in ("dw_bn", "pw_bn")] + \
            [self.p_bn[l] for l in (2, 3, 4, 5)] + [self.phi[l][k] for l in (2, 3, 4, 5) for k in ("bn1", "bn2")] + [self.final_bn]

    def _build_pack_table(self):
        """Device table of packing jobs (forward + data-gradient image of every conv) for the one-launch packer."""
        import ctypes
        lib = ops._lib.lib()
        nb = lib.mpn_conv_pack_desc_bytes()
        jobs = [(c, t) for c in self.convs for t in (0, 1)]
        host = (ctypes.c_ubyte * (nb * len(jobs)))()
        begin = 0
        dc = ops._lib.dtype_code(self.dtype)
        for j, (c, t) in enumerate(jobs):
            out = c.packed.bwd if t else c.packed.fwd
            blocks = lib.mpn_conv_pack_desc_fill(ctypes.byref(host, j * nb), ops.ptr(c.w), c.cin, c.cout, c.ksize, t, dc,
                                                 ops.ptr(out), begin)
            assert blocks > 0
            begin += blocks
        self._pack_table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(self.device)
        self._pack_jobs, self._pack_blocks = len(jobs), begin

    def repack_weights(self):
        """Refresh the packed (bf16/f32, MFMA tile order) copies after the f32 masters changed: one launch."""
        if ops.LAUNCH_JOBS_ALONE:      # (test switch: one packing launch per image instead of the batched table)
            for c in self.convs:
                c.packed.repack()
            return
        ops.call("mpn_conv_pack_weights_batched", ops.ptr(self._pack_table), self._pack_jobs, self._pack_blocks,
                 ops._lib.dtype_code(self.dtype), ops.stream_ptr())

    # ------------------------------------------------------------------ buffers (allocated once per input shape)
    def _buffers(self, N, H, W, head=True):
        """head=False: the backbone's buffers only (the person detector runs the frozen backbone under its own head)."""
        key = (N, H, W) if head else (N, H, W, "backbone")
        b = self._bufs.get(key)
        if b is not None:
            return b
        if H % DIVISOR or W % DIVISOR:
            raise ValueError(f"image height and width must be multiples of {DIVISOR} (got {H}x{W})")   # detector.py:45
        dt, dev = self.dtype, self.device

        def act(h, w, c):
            return torch.empty((N, h, w, c), dtype=dt, device=dev)

        b = {"shape": (N, H, W)}
        h, w = H // 2, W // 2
        c = self.stem_w.shape[3]
        b["stem"] = act(h, w, c)
        b["dw"], b["pw"], b["hw"] = [], [], [(h, w)]
        for blk in self.blocks:
            if blk["stride"] == 2:
                h, w = h // 2, w // 2
            b["dw"].append(act(h, w, c))
            c = blk["pw"].cout
            b["pw"].append(act(h, w, c))
            b["hw"].append((h, w))
        if not head:
            nbn0 = ops._lib.lib().mpn_bn_stats_num_parts
            b["stat_part"] = torch.empty(nbn0(N * (H // 2) * (W // 2)) * 2 * b["stem"].shape[3], dtype=torch.float32, device=dev)
            self._bufs[key] = b
            return b
        lv = {l: (H >> l, W >> l) for l in (2, 3, 4, 5)}
        b["lv"] = lv
        b["x"] = {l: act(*lv[l], DEPTH) for l in lv}
        b["p"] = {l: act(*lv[l], DEPTH) for l in lv}
        b["y1"] = {l: act(*lv[l], DEPTH) for l in lv}
        b["concat"] = act(*lv[2], 4 * DEPTH)
        b["y2"] = {l: act(*lv[l], DEPTH) for l in lv if l != 2}
        b["y2"][2] = b["concat"][..., :DEPTH]          # level 2 lives in its slice of the concat tensor (raw)
        b["final"] = act(*lv[2], 64)
        b["logits"] = torch.empty((N, lv[2][0], lv[2][1], NUM_KEYPOINTS + 1), dtype=torch.float32, device=dev)
        # scratch for batch-norm partial sums (forward statistics and backward reductions share it):
        # the largest producer decides
        nbn = ops._lib.lib().mpn_bn_stats_num_parts

        def rows(t):
            return t.numel() // t.shape[3]

        stat_floats = nbn(rows(b["stem"])) * 2 * b["stem"].shape[3]
        for i, blk in enumerate(self.blocks):
            hh, ww = b["hw"][i]
            cdw, cpw = b["dw"][i].shape[3], blk["pw"].cout
            stat_floats = max(stat_floats, ops.dwconv_num_parts(N, hh, ww, cdw, blk["stride"], dt) * 2 * cdw,
                              nbn(rows(b["dw"][i])) * 2 * cdw,
                              ops.conv_num_parts(N, *b["hw"][i + 1], 1) * 2 * cpw, nbn(rows(b["pw"][i])) * 2 * cpw,
                              ops.dwconv_bwd_data_bn_num_parts(N, hh, ww, cdw, blk["stride"], dt) * 2 * cdw,
                              ops.dwconv_wgrad_num_parts(N, hh, ww, cdw, blk["stride"], dt) * 2 * cdw,   # (the fused backward's rows)
                              # (the fused thin pointwise backward's batch-norm rows: one pair per split-K part)
                              ops.conv_wgrad_num_parts(N, *b["hw"][i + 1], cdw, cpw, 1, dt) * 2 * cdw)
        for l in lv:
            stat_floats = max(stat_floats, ops.conv_num_parts(N, *lv[l], 3) * 2 * DEPTH, nbn(N * lv[l][0] * lv[l][1]) * 2 * DEPTH)
        b["stat_part"] = torch.empty(stat_floats, dtype=torch.float32, device=dev)
        # per-level statistics scratch of the subnet (the levels' finalizes run batched, so their partials coexist)
        b["stat_lv"] = {l: torch.empty(max(ops.conv_num_parts(N, *lv[l], 3), nbn(N * lv[l][0] * lv[l][1])) * 2 * DEPTH,
                                       dtype=torch.float32, device=dev) for l in lv}
        b["fin"] = None     # batched finalize tables, built on first use (ops.BnFinalizeBatch / BnBwdFinalizeBatch)
        b["loss_part"] = torch.empty(ops._lib.lib().mpn_keypoint_loss_num_parts(N, *lv[2]) * 8, dtype=torch.float32, device=dev)
        b["losses"] = torch.zeros(8, dtype=torch.float32, device=dev)
        self._bufs[key] = b
        return b

    def _grad_buffers(self, b):
        if "g" in b:
            return b["g"]
        N, H, W = b["shape"]
        dt, dev = self.dtype, self.device
        g = {}
        g["dlogits"] = torch.empty_like(b["logits"])
        g["daux"] = {l: torch.empty((N, *b["lv"][l]), dtype=torch.float32, device=dev) for l in b["lv"]}
        g["final"] = torch.empty_like(b["final"])
        g["concat"] = torch.empty_like(b["concat"])
        for k in ("y1", "p", "x"):
            g[k] = {l: torch.empty_like(b[k][l]) for l in b["lv"]}
        g["y2"] = {l: torch.empty_like(b["y2"][l]) for l in b["lv"] if l != 2}
        g["y2"][2] = g["concat"][..., :DEPTH]          # ... and so does its gradient
        g["c"] = {}   # gradient w.r.t. the activated c_l (from the lateral convs)
        for i, name in FEATURE_BLOCKS.items():
            g["c"][name] = torch.empty_like(b["pw"][i - 1])
        g["pw"] = [torch.empty_like(t) for t in b["pw"]]
        g["dw"] = [torch.empty_like(t) for t in b["dw"]]
        g["stem"] = torch.empty_like(b["stem"])
        # weight-gradient split-K slabs: every layer gets its own region of one buffer, and ONE batched launch at the
        # end of backward reduces all of them into the gradient arena (ops.SlabReducer; 46 slabs = 87 launches otherwise)
        lib, dc = ops._lib.lib(), ops._lib.dtype_code(dt)
        sites = []   # (key, nparts, n, out)
        # the 3x3 convolutions that exist once per pyramid level share a grid per stage (ops.conv_bwd_weight_grouped): their slab
        # counts are those of that grid
        grouped_np = {}
        for group in self._wgrad_groups():
            hws = [self._conv_hw(b, c) for c in group]
            for c, np_ in zip(group, ops.conv_wgrad_grouped_num_parts(N, hws, group[0].cin, group[0].cout, group[0].ksize, dt)):
                grouped_np[id(c.dw)] = np_
        for c in self.convs:
            hw = self._conv_hw(b, c)
            np_ = grouped_np.get(id(c.dw))
            if np_ is None:
                np_ = ops.conv_wgrad_num_parts(N, hw[0], hw[1], c.cin, c.cout, c.ksize, dt)
            sites.append((id(c.dw), np_, c.w.numel(), c.dw))
        for i, blk in enumerate(self.blocks):
            hh, ww = b["hw"][i]
            cc = b["dw"][i].shape[3]
            sites.append((id(blk["dw_dw"]), lib.mpn_dwconv_wgrad_num_parts(N, hh, ww, cc, blk["stride"], dc), 9 * cc, blk["dw_dw"]))
        sites.append((id(self.stem_dw), lib.mpn_stem_conv_wgrad_num_parts(N, H, W), self.stem_w.numel(), self.stem_dw))
        sites.append((id(self._head_grad), lib.mpn_heatmap_head_bwd_num_parts(N * b["lv"][2][0] * b["lv"][2][1]),
                      self._head_grad.numel(), self._head_grad))
        total = sum(((np_ * n + 3) // 4) * 4 for _, np_, n, _ in sites)
        slab = torch.empty(total, dtype=torch.float32, device=dev)
        deep_keys = {id(blk[k]) if k == "dw_dw" else id(blk[k].dw) for blk in self.blocks if blk["i"] >= DP_DEEP_FROM_BLOCK
                     for k in ("pw", "dw_dw")}
        shallow_keys = {id(blk[k]) if k == "dw_dw" else id(blk[k].dw) for blk in self.blocks if blk["i"] < DP_DEEP_FROM_BLOCK
                        for k in ("pw", "dw_dw")} | {id(self.stem_dw)}
        g["slab"], jobs, off = {}, ([], [], []), 0
        for key, np_, n, out in sites:
            view = slab[off:off + np_ * n]
            g["slab"][key] = view
            jobs[1 if key in deep_keys else (2 if key in shallow_keys else 0)].append((view, np_, n, out.view(-1)))
            off += ((np_ * n + 3) // 4) * 4
        # three launches: the head / FPN gradients, then the deep backbone blocks', are complete (and can be all-reduced) while
        # the rest of the backbone still runs
        g["reducer"] = tuple(ops.SlabReducer(j, dev) for j in jobs)
        b["g"] = g
        return g

    def _wgrad_groups(self):
        LV = (2, 3, 4, 5)
        return [[self.phi[l]["conv2"] for l in LV], [self.phi[l]["conv1"] for l in LV], [self.pconv[l] for l in LV]]

    def _conv_hw(self, b, c):
        n = c.name
        if n.startswith("MobilenetV1"):
            i = int(n.split("Conv2d_")[1].split("_")[0])
            return b["hw"][i]
        if n.startswith("keypoint_fpn") or n.startswith("phi_subnet"):
            l = int(n.split("/")[0][-1]) if n.startswith("phi") else int(n.split("/")[1].replace("lateral", "").replace("p", ""))
            return b["lv"][l]
        return b["lv"][2]

    # ------------------------------------------------------------------ forward
    def _finish_bn(self, bn, b, nparts, count, training):
        if training:
            ops.bn_finalize(bn, b["stat_part"], nparts, count, training=True)

    def prepare_inference(self):
        """is_training=False: every batch-norm becomes the affine of its moving statistics. With cache_inference_affine (a
        frozen backbone under the person detector, the joint inference graph) the 40 small launches run once and again only
        after the variables changed (mark_variables_changed: load_state_dict, a training forward, an optimizer step, a
        replayed train step). The flag is host state and NOT part of a hipGraph: whoever captures an inference pass with the
        cache on owns the refresh - compare `var_version` before every replay and run this method eagerly when it moved (the
        affines live in persistent buffers, so an eager refresh serves the captured launches too): inference/detector.py."""
        if self.cache_inference_affine and self._infer_clean:
            return
        for bn in self.all_bn:
            ops.bn_inference_affine(bn)
        self._infer_clean = True

    def backbone_forward(self, images, is_training, b=None):
        """mobilenet_v1 (detector/backbones/mobilenet_v1.py:11-79). Returns {'c2'..'c5': (raw NHWC tensor, Affine)}."""
        N, H, W, _ = images.shape
        b = b or self._buffers(N, H, W)
        if not is_training:
            self.prepare_inference()
        else:
            self.mark_variables_changed()   # (the finalizes below overwrite the affines and move the statistics)
        sp = b["stat_part"]
        c0 = self.stem_w.shape[3]
        # training (opt-in, fuse_stem_stats): the stem kernel writes the batch-norm partial sums of its own output
        stem_rows = ops.stem_conv_fwd_num_parts(N, H, W, c0, self.dtype) if (is_training and self.fuse_stem_stats) else 0
        if stem_rows * 2 * c0 > sp.numel():
            stem_rows = 0
        stem = ops.stem_conv_fwd(images, self.stem_w, c0, self.dtype, out=b["stem"], stats_part=sp if stem_rows > 0 else None)
        if is_training:
            cnt = stem.numel() // stem.shape[3]
            if stem_rows > 0:
                ops.bn_finalize(self.stem_bn, sp, stem_rows, cnt)
            else:
                _, nparts = ops.bn_stats(stem, sp)
                ops.bn_finalize(self.stem_bn, sp, nparts, cnt)
        x, aff = stem, self.stem_bn.affine
        feats = {}
        for i, blk in enumerate(self.blocks):
            hin, win = b["hw"][i]
            h, w = b["hw"][i + 1]
            ydw = ops.dwconv_fwd(x, blk["dw_w"], blk["stride"], aff, out=b["dw"][i], stats_part=sp if is_training else None)
            if is_training:
                ops.bn_finalize(blk["dw_bn"], sp, ops.dwconv_num_parts(N, hin, win, ydw.shape[3], blk["stride"], self.dtype),
                                ydw.numel() // ydw.shape[3])
            with _lib.tagged("pointwise"):
                ypw = ops.conv_fwd(ydw, blk["pw"].packed.fwd, blk["pw"].cout, 1, blk["dw_bn"].affine, out=b["pw"][i],
                                   stats_part=sp if is_training else None)
            if is_training:
                ops.bn_finalize(blk["pw_bn"], sp, ops.conv_num_parts(N, h, w, 1), N * h * w)
            x, aff = ypw, blk["pw_bn"].affine
            if blk["i"] in FEATURE_BLOCKS:
                feats[FEATURE_BLOCKS[blk["i"]]] = (x, aff)
        return feats

    def subnet_forward(self, feats, is_training, b, inference_outputs=False):
        """KeypointSubnet (detector/keypoint_subnet.py:11-62) on top of feature_pyramid_network (detector/fpn.py:36-55)."""
        N = b["shape"][0]
        sp = b["stat_part"] if is_training else None
        sep = is_training
        prev = None
        LV = (2, 3, 4, 5)
        for l in (5, 4, 3, 2):
            raw, aff = feats[f"c{l}"]
            ops.conv_fwd(raw, self.lateral[l].packed.fwd, DEPTH, 1, aff, out=b["x"][l], up_res=prev)    # fpn.py:38,50-51
            prev = b["x"][l]
        # stage by stage over the four independent pyramid levels: each stage's four convolutions in ONE grid (the small
        # levels fill in at the level-2 kernel's throughput), their statistics in per-level scratch, ONE finalize launch
        fin, spl = None, [None] * 4
        if sep:
            if b["fin"] is None:
                b["fin"] = self._finalize_tables(b)
            fin, spl = b["fin"], [b["stat_lv"][l] for l in LV]
        ops.conv_fwd_grouped([b["x"][l] for l in LV], [self.pconv[l].packed.fwd for l in LV], DEPTH, 3, [None] * 4,
                             [b["p"][l] for l in LV], spl)                                                # fpn.py:39,52
        if sep:
            fin["p"].run()
        ops.conv_fwd_grouped([b["p"][l] for l in LV], [self.phi[l]["conv1"].packed.fwd for l in LV], DEPTH, 3,
                             [self.p_bn[l].affine for l in LV], [b["y1"][l] for l in LV], spl)
        if sep:
            fin["bn1"].run()
        ops.conv_fwd_grouped([b["y1"][l] for l in LV], [self.phi[l]["conv2"].packed.fwd for l in LV], DEPTH, 3,
                             [self.phi[l]["bn1"].affine for l in LV], [b["y2"][l] for l in LV], spl)      # (level 2: into the concat slice)
        if sep:
            fin["bn2"].run()
        for l in (3, 4, 5):
            ops.bilinear_up_fwd(b["y2"][l], 2 ** (l - 2), b["concat"], (l - 2) * DEPTH, self.phi[l]["bn2"].affine)   # :86 + :37
        return self._subnet_final(b, N, sp, sep, inference_outputs)

    def _subnet_final(self, b, N, sp, sep, inference_outputs):
        h, w = b["lv"][2]
        ops.conv_fwd(b["concat"], self.final_conv.packed.fwd, 64, 3, self.concat_affine, out=b["final"], stats_part=sp)   # :38
        if sep:
            ops.bn_finalize(self.final_bn, sp, ops.conv_stats_rows(N, h, w, b["concat"].shape[3], 64, 3, self.dtype), N * h * w)
        if inference_outputs:
            return ops.heatmap_head_fwd(b["final"], self.heat_w, self.heat_b, self.final_bn.affine, inference=True)
        return ops.heatmap_head_fwd(b["final"], self.heat_w, self.heat_b, self.final_bn.affine, out=b["logits"])

    def _finalize_tables(self, b):
        """Device tables of the batched finalizes (forward and backward) of the subnet's three stages x four levels."""
        N = b["shape"][0]
        nbn = ops._lib.lib().mpn_bn_stats_num_parts
        fwd = {k: [] for k in ("p", "bn1", "bn2")}
        bwd = {k: [] for k in ("p", "bn1", "bn2")}
        for l in (2, 3, 4, 5):
            h, w = b["lv"][l]
            cnt = N * h * w
            for k, bn in (("p", self.p_bn[l]), ("bn1", self.phi[l]["bn1"]), ("bn2", self.phi[l]["bn2"])):
                rows3 = ops.conv_stats_rows(N, h, w, DEPTH, DEPTH, 3, self.dtype)     # (every 3x3 layer of the subnet: DEPTH -> DEPTH)
                fwd[k].append((bn, b["stat_lv"][l], rows3, cnt))
                # p / bn1: reduced inside the data gradient that produces their gradient (conv rows, sum g * x with the raw x)
                if k != "bn2" and self._fused_conv_bn():
                    bwd[k].append((bn, b["stat_lv"][l], rows3, cnt, True))
                else:
                    bwd[k].append((bn, b["stat_lv"][l], nbn(cnt), cnt))
        out = {k: ops.BnFinalizeBatch(v, self.device) for k, v in fwd.items()}
        out.update({"d" + k: ops.BnBwdFinalizeBatch(v, self.device) for k, v in bwd.items()})
        return out

    def _fused_conv_bn(self):
        """The 3x3 data gradients of the subnet also reduce for the batch-norm they feed (mpn_conv_bwd_data_bn_grouped)."""
        return self.fuse_conv_bn and ops.conv_bwd_data_bn_supported(DEPTH, DEPTH, 3, self.dtype)

    def forward(self, images, is_training):
        """images: [N,H,W,3] f32 in [0,1] (or uint8). Returns (logits [N,H/4,W/4,18] f32 NHWC,
        enriched_features {'p2'..'p5': raw FPN outputs NHWC})."""
        N, H, W, _ = images.shape
        b = self._buffers(N, H, W)
        feats = self.backbone_forward(images, is_training, b)
        logits = self.subnet_forward(feats, is_training, b)
        self._last = (b, feats, images)
        return logits, {f"p{l}": b["p"][l] for l in (2, 3, 4, 5)}

    def predict(self, images):
        """Inference outputs of create_pb.py:73-76: (sigmoid keypoint heatmaps [N,h,w,17], segmentation masks [N,h,w])."""
        N, H, W, _ = images.shape
        b = self._buffers(N, H, W)
        feats = self.backbone_forward(images, False, b)
        return self.subnet_forward(feats, False, b, inference_outputs=True)

    # ------------------------------------------------------------------ loss + backward
    def compute_losses(self, labels, with_grad=True):
        b = self._last[0]
        g = self._grad_buffers(b) if with_grad else None
        ps = [b["p"][l] for l in (2, 3, 4, 5)]
        return ops.keypoint_loss(b["logits"], labels, ps, g["dlogits"] if g else None,
                                 [g["daux"][l] for l in (2, 3, 4, 5)] if g else None, b["loss_part"], b["losses"])

    def _wgrad(self, fn):
        """Run a weight-gradient launch sequence on the side stream: it only needs dy (just produced on the main stream)
        and forward activations, so it overlaps the main stream's critical chain (bn_backward -> dgrad -> ...).
        Captured into the hipGraph as a fork (event wait); `backward` joins once at the end."""
        main = torch.cuda.current_stream()
        if self._wstream is None:
            fn()
            return
        self._wstream.wait_stream(main)
        with torch.cuda.stream(self._wstream):
            fn()

    @property
    def backbone_deep_begin(self):
        """Offset in the flat gradient arena where backbone block DP_DEEP_FROM_BLOCK begins (see backward)."""
        return backbone_deep_begin_of(self._train_arena)

    @property
    def backbone_grad_end(self):
        """Offset in the flat gradient arena where the backbone's variables end (the arena is laid out backbone first)."""
        return backbone_grad_end_of(self._train_arena)

    def backward(self, part=None):
        """Gradients of the total loss w.r.t. every trainable variable -> self.grad (call after compute_losses).
        part=0: head + subnet + FPN only (their gradients are final when it returns); part=1: backbone blocks 13..7, after
        part 0; part=2: blocks 6..1 and the stem, after part 1; None: all three. Data-parallel training all-reduces the
        part-0 gradients while part 1 runs and the part-1 gradients (the bulk of the backbone's parameters) while part 2 runs.
        Two HIP streams: the main one carries the activation-gradient chain, the side one all weight gradients
        (MFMA split-K kernels + slab reductions), so HBM-bound batch-norm passes overlap MFMA-bound wgrad kernels."""
        b, feats, images = self._last
        g = self._grad_buffers(b)
        sp, slab = b["stat_part"], g["slab"]
        if self.overlap_wgrad and self._wstream is None:
            self._wstream = torch.cuda.Stream(device=self.device)
        W = self._wgrad
        if part in (None, 0):
            self._backward_head(b, g, feats, sp, slab, W)
            if self._wstream is not None:
                torch.cuda.current_stream().wait_stream(self._wstream)
            g["reducer"][0].run()
        for ph in (1, 2):
            if part in (None, ph):
                self._backward_backbone(b, g, images, sp, slab, W, ph)
                if self._wstream is not None:
                    torch.cuda.current_stream().wait_stream(self._wstream)   # join: every slab of this phase is written
                g["reducer"][ph].run()   # (after phase 2: every gradient is in the arena)

    def _backward_head(self, b, g, feats, sp, slab, W):
        # ---- head + final conv
        if self._fused_conv_bn() and ops.heatmap_head_bwd_bn_supported(b["final"].shape[3], self.dtype):
            # the head's backward kernel also reduces for final_bn (its input's batch-norm): one launch and one pass less
            rows = ops.heatmap_head_bwd(b["final"], g["dlogits"], self.heat_w, self.final_bn.affine, g["final"], self._head_grad,
                                        slab[id(self._head_grad)], reduce=False, bn_part=sp)
            ops.bn_backward(self.final_bn, g["final"], b["final"], sp, reduced_parts=rows, raw=True)
        else:
            ops.heatmap_head_bwd(b["final"], g["dlogits"], self.heat_w, self.final_bn.affine, g["final"], self._head_grad,
                                 slab[id(self._head_grad)], reduce=False)
            ops.bn_backward(self.final_bn, g["final"], b["final"], sp)
        W(lambda: ops.conv_bwd_weight(b["concat"], g["final"], 3, self.concat_affine, self.final_conv.dw, slab[id(self.final_conv.dw)], reduce=False))
        ops.conv_fwd(g["final"], self.final_conv.packed.bwd, 4 * DEPTH, 3, None, out=g["concat"])
        # ---- phi subnets + p{l}_batch_norm, stage by stage over the four levels (see subnet_forward): reductions into
        # per-level scratch, ONE finalize launch per stage, then the applies and the convolutions' gradients
        if b["fin"] is None:
            b["fin"] = self._finalize_tables(b)
        fin, spl = b["fin"], b["stat_lv"]
        LV = (2, 3, 4, 5)
        sps = [spl[l] for l in LV]
        bn2s, bn1s, pbns = [self.phi[l]["bn2"] for l in LV], [self.phi[l]["bn1"] for l in LV], [self.p_bn[l] for l in LV]
        gy2, gy1, gp = [g["y2"][l] for l in LV], [g["y1"][l] for l in LV], [g["p"][l] for l in LV]
        by2, by1, bp = [b["y2"][l] for l in LV], [b["y1"][l] for l in LV], [b["p"][l] for l in LV]
        for l in (3, 4, 5):   # (level 2's gradient IS the first slice of g["concat"])
            ops.bilinear_up_bwd(g["concat"], 2 ** (l - 2), (l - 2) * DEPTH, DEPTH, out=g["y2"][l])
        ops.bn_bwd_reduce_grouped(bn2s, gy2, by2, sps)
        fin["dbn2"].run()
        none4 = [None] * 4
        fused = self._fused_conv_bn()
        ops.bn_bwd_apply_grouped(bn2s, gy2, by2)
        W(lambda: ops.conv_bwd_weight_grouped(by1, gy2, 3, [self.phi[l]["bn1"].affine for l in LV],
                                              [slab[id(self.phi[l]["conv2"].dw)] for l in LV]))
        if fused:   # conv2's data gradient also reduces for bn1 (and writes the gradient masked by bn1's ReLU)
            ops.conv_bwd_data_bn_grouped(gy2, [self.phi[l]["conv2"].packed.bwd for l in LV], DEPTH, bn1s, by1, gy1, sps)
        else:
            ops.conv_fwd_grouped([g["y2"][l] for l in LV], [self.phi[l]["conv2"].packed.bwd for l in LV], DEPTH, 3, none4,
                                 [g["y1"][l] for l in LV], none4)
            ops.bn_bwd_reduce_grouped(bn1s, gy1, by1, sps)
        fin["dbn1"].run()
        ops.bn_bwd_apply_grouped(bn1s, gy1, by1)
        W(lambda: ops.conv_bwd_weight_grouped(bp, gy1, 3, [self.p_bn[l].affine for l in LV],
                                              [slab[id(self.phi[l]["conv1"].dw)] for l in LV]))
        if fused:
            ops.conv_bwd_data_bn_grouped(gy1, [self.phi[l]["conv1"].packed.bwd for l in LV], DEPTH, pbns, bp, gp, sps)
        else:
            ops.conv_fwd_grouped([g["y1"][l] for l in LV], [self.phi[l]["conv1"].packed.bwd for l in LV], DEPTH, 3, none4,
                                 [g["p"][l] for l in LV], none4)
            ops.bn_bwd_reduce_grouped(pbns, gp, bp, sps)
        fin["dp"].run()
        # ---- FPN (top-down path reversed)
        # the four 3x3 convolutions (fpn.py:39,52) are independent: ONE grid for their weight gradients, one for their data
        # gradients; the nearest-upsample gradients then chain the levels
        ops.bn_bwd_apply_grouped(pbns, gp, bp, [g["daux"][l] for l in LV])
        W(lambda: ops.conv_bwd_weight_grouped([b["x"][l] for l in LV], gp, 3, none4, [slab[id(self.pconv[l].dw)] for l in LV]))
        ops.conv_fwd_grouped(gp, [self.pconv[l].packed.bwd for l in LV], DEPTH, 3, none4, [g["x"][l] for l in LV], none4)
        for l in (2, 3, 4, 5):
            if l > 2:
                ops.sumpool2x2(g["x"][l - 1], g["x"][l], accumulate=True)        # grad of nearest 2x upsample
            raw, aff = feats[f"c{l}"]
            if self.fuse_pw_bwd and self.fuse_pw_wide and l < 5 and aff is not None and ops.conv1x1_bwd_fused_supported(raw.shape[3], DEPTH, self.dtype):
                # (lateral2: both gradients in one pass over c2 and its gradient; no batch-norm below this one to reduce for here -
                #  c2 has a second consumer, the sum is reduced by the backbone's backward)
                ops.conv1x1_bwd_fused(raw, g["x"][l], self.lateral[l].w, aff, g["c"][f"c{l}"], slab[id(self.lateral[l].dw)], None)
                continue
            W(lambda: ops.conv_bwd_weight(raw, g["x"][l], 1, aff, self.lateral[l].dw, slab[id(self.lateral[l].dw)], reduce=False))
            # c5 has one consumer (lateral5): its data gradient also reduces for Conv2d_13_pointwise's batch-norm - the backbone's
            # backward pass starts from that finalize (`sp` is not touched in between)
            g["c5_reduced"] = 0
            if l == 5 and self.fuse_conv_bn and ops.conv_bwd_data_bn_supported(DEPTH, raw.shape[3], 1, self.dtype):
                g["c5_reduced"] = ops.conv_bwd_data_bn(g["x"][l], self.lateral[l].packed.bwd, raw.shape[3], 1, self.blocks[-1]["pw_bn"], raw,
                                                       g["c"][f"c{l}"], sp)
            else:
                ops.conv_fwd(g["x"][l], self.lateral[l].packed.bwd, raw.shape[3], 1, None, out=g["c"][f"c{l}"])

    def _backward_backbone(self, b, g, images, sp, slab, W, phase):
        """phase 1: blocks 13 .. DP_DEEP_FROM_BLOCK; phase 2: the blocks below and the stem. The chain's state between the two
        (the gradient tensor, whether the last data gradient already reduced for the next batch-norm, whether a lateral's
        gradient is already added) is Python state handed over through `g` - static per shape, so both phases capture."""
        first = DP_DEEP_FROM_BLOCK - 1            # index of the first deep block
        if phase == 1:
            dA = g["c"]["c5"]
            reduced, raw_sums = g.get("c5_reduced", 0), True   # (the lateral's data gradient may have reduced for the last batch-norm)
            lateral_added = False
            rng = range(len(self.blocks) - 1, first - 1, -1)
        else:
            dA, reduced, raw_sums, lateral_added = g["bb_chain"]
            rng = range(first - 1, -1, -1)
        for i in rng:
            blk = self.blocks[i]
            if blk["i"] in FEATURE_BLOCKS and blk["i"] != 13 and not lateral_added:
                ops.add_inplace(dA, g["c"][FEATURE_BLOCKS[blk["i"]]])
            # the thin pointwise layers: weight gradient, data gradient and the reduction for the depthwise batch-norm below in ONE
            # pass over the layer's input and dY (each tensor once instead of twice); on the thinnest one the layer's own batch-norm
            # apply pass happens while dY is staged (two more passes over its output tensor less)
            pw_fused = self.fuse_pw_bwd and self.fuse_conv_bn and ops.conv1x1_bwd_fused_supported(blk["pw"].cin, blk["pw"].cout, self.dtype) and \
                (blk["pw"].cin <= 64 or self.fuse_pw_wide)
            pw_apply = pw_fused and self.fuse_pw_apply and ops.conv1x1_bwd_fused_apply_supported(blk["pw"].cin, blk["pw"].cout, self.dtype)
            ops.bn_backward(blk["pw_bn"], dA, b["pw"][i], sp, reduced_parts=reduced, raw=raw_sums and reduced > 0, apply=not pw_apply)
            raw_sums = False                                    # (depthwise data gradients sum g * xhat themselves)
            if pw_fused:
                with _lib.tagged("pointwise"):
                    rows = ops.conv1x1_bwd_fused(b["dw"][i], dA, blk["pw"].w, blk["dw_bn"], g["dw"][i], slab[id(blk["pw"].dw)], sp,
                                                 apply_bn=blk["pw_bn"] if pw_apply else None, y_raw=b["pw"][i] if pw_apply else None)
                ops.bn_backward(blk["dw_bn"], g["dw"][i], b["dw"][i], sp, reduced_parts=rows, raw=True)
            else:
                with _lib.tagged("pointwise"):
                    W(lambda: ops.conv_bwd_weight(b["dw"][i], dA, 1, blk["dw_bn"].affine, blk["pw"].dw, slab[id(blk["pw"].dw)], reduce=False))
                # the deep pointwise layers' data gradients also reduce for the depthwise batch-norm they feed
                if self.fuse_conv_bn and ops.conv_bwd_data_bn_supported(blk["pw"].cout, blk["pw"].cin, 1, self.dtype):
                    with _lib.tagged("pointwise"):
                        rows = ops.conv_bwd_data_bn(dA, blk["pw"].packed.bwd, blk["pw"].cin, 1, blk["dw_bn"], b["dw"][i], g["dw"][i], sp)
                    ops.bn_backward(blk["dw_bn"], g["dw"][i], b["dw"][i], sp, reduced_parts=rows, raw=True)
                else:
                    with _lib.tagged("pointwise"):
                        ops.conv_fwd(dA, blk["pw"].packed.bwd, blk["pw"].cin, 1, None, out=g["dw"][i])
                    ops.bn_backward(blk["dw_bn"], g["dw"][i], b["dw"][i], sp)
            xin = b["pw"][i - 1] if i > 0 else b["stem"]
            ain = self.blocks[i - 1]["pw_bn"].affine if i > 0 else self.stem_bn.affine
            dst = g["pw"][i - 1] if i > 0 else g["stem"]
            prev_feature = i > 0 and self.blocks[i - 1]["i"] in FEATURE_BLOCKS
            # stride-1 layers whose input is not an FPN feature: both gradients and the reduction for the batch-norm below in one
            # walk over dY, the raw input and dA (three tensor passes instead of five)
            if self.fuse_dw_bwd and self.fuse_dw_bn and blk["stride"] == 1 and not prev_feature and \
                    ops.dwconv_bwd_fused_supported(dst.shape[0], *b["hw"][i], dst.shape[3], 1, dst.dtype):
                prev_bn = self.blocks[i - 1]["pw_bn"] if i > 0 else self.stem_bn
                _, reduced = ops.dwconv_bwd_fused(xin, g["dw"][i], blk["dw_w"], prev_bn, None, out=dst, wpart=slab[id(blk["dw_dw"])],
                                                  bn_part=sp, reduce=False)
                lateral_added = False
                dA = dst
                continue
            prev_bn = self.blocks[i - 1]["pw_bn"] if i > 0 else self.stem_bn
            # stride-2 layers (even maps): the same in one walk, the FPN lateral's gradient into the feature map below added inside it
            if self.fuse_dw_bwd and self.fuse_dw_bwd_s2 and self.fuse_dw_bn and self.fuse_lateral_add and blk["stride"] == 2 and \
                    ops.dwconv_bwd_fused_supported(dst.shape[0], *b["hw"][i], dst.shape[3], 2, dst.dtype):
                addend = g["c"][FEATURE_BLOCKS[self.blocks[i - 1]["i"]]] if prev_feature else None
                _, reduced = ops.dwconv_bwd_fused(xin, g["dw"][i], blk["dw_w"], prev_bn, None, out=dst, wpart=slab[id(blk["dw_dw"])],
                                                  bn_part=sp, reduce=False, stride=2, addend=addend)
                lateral_added = addend is not None
                dA = dst
                continue
            W(lambda: ops.dwconv_bwd_weight(xin, g["dw"][i], blk["stride"], ain, blk["dw_dw"], slab[id(blk["dw_dw"])], reduce=False))
            # the data gradient also reduces for the batch-norm it feeds (one read of dA and one launch less), unless a
            # lateral's gradient still has to be added to dA first
            prev_x = b["pw"][i - 1] if i > 0 else b["stem"]
            # c2..c4 have two consumers (the next depthwise conv and an FPN lateral, mobilenet_v1.py:76-79): the lateral's
            # gradient is added inside the data-gradient kernel where it can be (stride-2 layers with even maps - all three
            # at the reference's input sizes), otherwise by add_inplace at the top of the next iteration
            addend = g["c"][FEATURE_BLOCKS[self.blocks[i - 1]["i"]]] if prev_feature else None
            lateral_added = addend is not None and self.fuse_lateral_add and \
                ops.dwconv_bwd_data_add_supported(dst.shape[0], *b["hw"][i], dst.shape[3], blk["stride"], dst.dtype)
            if not lateral_added:
                addend = None
            reduced = 0
            if self.fuse_dw_bn and (not prev_feature or lateral_added) and \
                    ops.dwconv_bwd_data_bn_num_parts(dst.shape[0], *b["hw"][i], dst.shape[3], blk["stride"], dst.dtype) > 0:
                _, reduced = ops.dwconv_bwd_data(g["dw"][i], blk["dw_w"], b["hw"][i], blk["stride"], out=dst, bn=prev_bn,
                                                 x_bn=prev_x, part=sp, addend=addend)
            else:
                ops.dwconv_bwd_data(g["dw"][i], blk["dw_w"], b["hw"][i], blk["stride"], out=dst, addend=addend)
            dA = dst
        if phase == 1:
            g["bb_chain"] = (dA, reduced, raw_sums, lateral_added)
            return
        ops.bn_backward(self.stem_bn, g["stem"], b["stem"], sp, reduced_parts=reduced)
        W(lambda: ops.stem_conv_bwd_weight(images, g["stem"], self.stem_dw, slab[id(self.stem_dw)], reduce=False))

    def add_weight_decay_gradients(self, weight_decay):
        """keypoints_model.py:129-138: + wd * l2_loss(k) for every 'weights'/'kernel' variable except depthwise."""
        if self._wd is None:
            ks = [k for k in self.vars if ("weights" in k or "kernel" in k) and "depthwise_weights" not in k]
            self._wd = ops.AxpyBatch([self.vars[k] for k in ks], [self.grads[k] for k in ks])
        self._wd.run(weight_decay)

    def add_weight_decay_loss(self, weight_decay):
        """The regularisation term itself: total_loss += sum_k wd * l2_loss(k) over the same variables
        (tf.losses.get_total_loss(add_regularization_losses=True), keypoints_model.py:24-27,79). Call after
        compute_losses; the per-term losses stay as they are."""
        total = self._last[0]["losses"][6:7]
        if self._l2 is None:
            self._l2 = ops.L2LossBatch([w for k, w in self.vars.items()
                                        if ("weights" in k or "kernel" in k) and "depthwise_weights" not in k])
        self._l2.run(weight_decay, total)

    # ------------------------------------------------------------------ optimizer
    def optimizer_step(self, initial_learning_rate, num_steps, grad_scale=1.0):
        """Cosine LR + clip(+-200) + TF-Adam over the flat arena, then refresh the packed weights."""
        ops.adam_prepare(self.global_step, self.hyper, initial_learning_rate, num_steps)
        ops.adam_step(self.theta, self.grad, self.adam_m, self.adam_v, self.hyper, grad_scale=grad_scale)
        self.mark_variables_changed()
        self.repack_weights()
