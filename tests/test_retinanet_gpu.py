"""GPU: the RetinaNet person-detector head (BASELINE config 4) against oracle/retinanet.py:
stride-2 gathers, anchor matching (bit-exact integers), losses and every gradient, one optimizer step, NMS."""
import numpy as np
import pytest
import torch

from oracle import network as onet
from oracle import retinanet as R
from util import dev, rnd

pytestmark = pytest.mark.gpu
HP = {"initial_learning_rate": 1e-3, "num_steps": 150000, "weight_decay": 5e-5, "localization_loss_weight": 1.0,
      "classification_loss_weight": 2.0, "gamma": 2.0, "alpha": 0.25, "depth_multiplier": 1.0}


def _ops():
    from multiposenet_amd import ops
    return ops


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,C", [(2, 14, 22, 128), (1, 7, 11, 64), (2, 8, 8, 32)])
def test_stride2_conv_as_patchify_plus_1x1(cuda, dtype, N, H, W, C):
    """conv2d_same(k=3, stride=2) (layer_utils.py:19-39) = mpn_patchify3x3s2 + the 1x1 kernel; its data gradient =
    1x1 data gradient + mpn_unpatchify3x3s2; affine + ReLU of the producer applied inside the gather."""
    ops = _ops()
    rs = np.random.RandomState(H + C)
    Cout = 128
    x = rnd(rs.randn(N, H, W, C), dtype)
    w = (rs.randn(3, 3, C, Cout) / np.sqrt(9 * C)).astype(np.float32)
    sc = torch.tensor(0.5 + rs.rand(C), dtype=torch.float32); sh = torch.tensor(rs.randn(C) * 0.5, dtype=torch.float32)
    a = rnd(torch.relu(x * sc + sh), dtype)
    xin = a.permute(0, 3, 1, 2).clone().requires_grad_(True)
    want = onet.conv2d_same(xin, rnd(w, dtype), stride=2)
    OH, OW = (H + 1) // 2, (W + 1) // 2
    assert tuple(want.shape) == (N, Cout, OH, OW)
    patches = torch.empty((N, OH, OW, 9 * C), device="cuda", dtype=dtype)
    ops.patchify3x3s2(dev(x, dtype), patches, ops.Affine(dev(sc), dev(sh), 1))
    pc = ops.PackedConv(dev(w).view(1, 1, 9 * C, Cout), dtype)
    y = ops.conv_fwd(patches, pc.fwd, Cout, 1)
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    np.testing.assert_allclose(y.float().cpu().numpy(), want.permute(0, 2, 3, 1).detach().numpy(), atol=tol * float(want.abs().max()), rtol=tol)
    dy = rnd(rs.randn(N, OH, OW, Cout), dtype)
    want.backward(dy.permute(0, 3, 1, 2))
    dp = ops.conv_fwd(dev(dy, dtype), pc.bwd, 9 * C, 1)
    dx = torch.empty((N, H, W, C), device="cuda", dtype=dtype)
    ops.unpatchify3x3s2(dp, dx)
    g = xin.grad.permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(dx.float().cpu().numpy(), g, atol=(2e-5 if dtype == torch.float32 else 3e-2) * float(np.abs(g).max()), rtol=tol)


def _random_groundtruth(rs, B, maxn):
    boxes = np.zeros((B, maxn, 4), np.float32)
    num = rs.randint(0, maxn + 1, B).astype(np.int32)
    for b in range(B):
        for n in range(maxn):
            cy, cx = rs.rand(2)
            h, w = 0.05 + 0.5 * rs.rand(2)
            boxes[b, n] = [max(cy - h / 2, 0), max(cx - w / 2, 0), min(cy + h / 2, 1), min(cx + w / 2, 1)]
    return boxes, num


def test_anchor_matching_is_bit_exact(cuda):
    """mpn_retina_match vs the numpy restatement of get_training_targets: matches identical (arg-max ties, forced matches,
    the 0.05 rule, images without boxes), targets to float32 rounding of logf."""
    from multiposenet_amd import _lib
    from multiposenet_amd.retinanet import generate_anchors
    rs = np.random.RandomState(3)
    for (H, W, B, maxn) in [(128, 256, 4, 7), (256, 384, 3, 20)]:
        anchors, _ = generate_anchors(H, W)
        boxes, num = _random_groundtruth(rs, B, maxn)
        num[0] = 0                                         # an image without people
        boxes[1, 1] = boxes[1, 0]                          # duplicated box: equal ious, the first index wins
        if B > 2:
            boxes[2, 0] = [0.0, 0.0, 0.004, 0.004]         # tiny box: best anchor has iou < 0.05 -> no forced match
            num[2] = max(num[2], 2)
        A = anchors.shape[0]
        d_an, d_bx, d_nb = dev(anchors), dev(boxes), torch.tensor(num).cuda()
        matches = torch.empty((B, A), dtype=torch.int32, device="cuda")
        targets = torch.empty((B, A, 4), dtype=torch.float32, device="cuda")
        nm = torch.zeros(1, dtype=torch.int32, device="cuda")
        ws = torch.empty(_lib.lib().mpn_retina_match_workspace_bytes(B, maxn), dtype=torch.uint8, device="cuda")
        for _ in range(2):                                 # twice: the workspace / counter reset themselves
            _lib.call("mpn_retina_match", _lib.ptr(d_an), _lib.ptr(d_bx), _lib.ptr(d_nb), B, A, maxn, 0.5, 0.5, _lib.ptr(matches),
                      _lib.ptr(targets), _lib.ptr(nm), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        got_m, got_t = matches.cpu().numpy(), targets.cpu().numpy()
        total = 0
        for b in range(B):
            wt, wm = R.get_training_targets(anchors, boxes[b, :num[b]])
            np.testing.assert_array_equal(got_m[b], wm, err_msg=f"image {b}")
            np.testing.assert_allclose(got_t[b], wt, rtol=2e-6, atol=2e-6)
            total += int((wm >= 0).sum())
        assert int(nm.item()) == total and total > 0


def _level_tensors(flat_boxes, flat_logits, shapes, dtype):
    """[B,A,4] / [B,A] in the reference's anchor order -> per-level NHWC tensors [B,h,w,24] / [B,h,w,8] (padding channels 0)."""
    B = flat_boxes.shape[0]
    bx, lg, off = [], [], 0
    for (h, w) in shapes:
        n = h * w * 6
        bx.append(dev(flat_boxes[:, off:off + n].reshape(B, h, w, 24), dtype))
        l8 = np.zeros((B, h, w, 8), np.float32)
        l8[..., :6] = flat_logits[:, off:off + n].reshape(B, h, w, 6)
        lg.append(dev(l8, dtype))
        off += n
    return bx, lg


@pytest.mark.parametrize("H,W,B,mean", [(128, 256, 3, -3.0), (256, 256, 2, 1.0)], ids=["lds-candidates", "global-candidates"])
def test_nms_equals_the_restatement(cuda, H, W, B, mean):
    """mpn_retina_nms (sigmoid, threshold, decode, clip, greedy NMS, zero padding) vs oracle.get_predictions: the same
    detections in the same order (scores / boxes to float32 rounding of expf), on crowded random predictions. The second
    case leaves ~6000 of 8184 anchors above the threshold: more candidates than the selection kernel keeps in LDS (4096),
    so it walks them in the workspace instead."""
    import ctypes
    from multiposenet_amd import _lib
    from multiposenet_amd.retinanet import generate_anchors
    rs = np.random.RandomState(11)
    anchors, shapes = generate_anchors(H, W)
    A = anchors.shape[0]
    enc = (rs.randn(B, A, 4) * 0.5).astype(np.float32)
    logit = (rs.randn(B, A) * 1.5 + mean).astype(np.float32)
    logit[B - 1] = -9.0                                           # an image with nothing above the threshold
    bias_c = (rs.randn(6) * 0.1).astype(np.float32); bias_b = (rs.randn(24) * 0.05).astype(np.float32)
    bx, lg = _level_tensors(enc, logit, shapes, torch.float32)
    d_bc, d_bb, d_an = dev(bias_c), dev(bias_b), dev(anchors)     # (kept alive: a temporary's memory is re-used at once)
    PA, IA = ctypes.c_void_p * 5, ctypes.c_int * 5
    for (thr, iou_thr, md) in [(0.05, 0.5, 25), (0.3, 0.6, 25), (0.05, 0.3, 7)]:
        ob = torch.full((B, md, 4), 7.0, device="cuda"); os_ = torch.full((B, md), 7.0, device="cuda")
        on = torch.zeros(B, dtype=torch.int32, device="cuda")
        ws = torch.empty(_lib.lib().mpn_retina_nms_workspace_bytes(B, A), dtype=torch.uint8, device="cuda")
        _lib.call("mpn_retina_nms", PA(*[_lib.ptr(t) for t in lg]), PA(*[_lib.ptr(t) for t in bx]), IA(*[s[0] for s in shapes]),
                  IA(*[s[1] for s in shapes]), _lib.MPN_F32, _lib.ptr(d_bc), _lib.ptr(d_bb), _lib.ptr(d_an), B,
                  thr, iou_thr, md, _lib.ptr(ob), _lib.ptr(os_), _lib.ptr(on), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        k_of = np.arange(A) % 6                                   # anchor a of a location = channel k (box_predictor.py:72-81)
        wb, wsx, wn = R.get_predictions(enc + bias_b.reshape(6, 4)[k_of][None], logit + bias_c[k_of][None], anchors, thr, iou_thr, md)
        np.testing.assert_array_equal(on.cpu().numpy(), wn)
        np.testing.assert_allclose(os_.cpu().numpy(), wsx, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ob.cpu().numpy(), wb, rtol=1e-4, atol=1e-5)
        assert wn[B - 1] == 0 and wn[0] > 3


def _setup(seed, B, H, W):
    rs = np.random.RandomState(seed)
    bb = onet.randomize_bn(onet.init_params(seed), seed + 1)
    hp = R.init_head_params(seed + 2)
    for k in hp:                                                   # non-trivial batch-norm parameters, livelier outputs
        if k.endswith("/gamma"):
            hp[k] = (0.7 + 0.6 * rs.rand(*hp[k].shape)).astype(np.float32)
        elif k.endswith("/beta"):
            hp[k] = (0.2 * rs.randn(*hp[k].shape)).astype(np.float32)
    hp["class_net/logits/kernel"] = (rs.randn(3, 3, 64, 6) * 0.05).astype(np.float32)
    hp["box_net/encoded_boxes/kernel"] = (rs.randn(3, 3, 64, 24) * 0.05).astype(np.float32)
    img = rs.rand(B, H, W, 3).astype(np.float32)
    boxes, num = _random_groundtruth(rs, B, 5)
    num = np.maximum(num, 1).astype(np.int32)
    return bb, hp, img, boxes, num


def test_detector_train_step_f32_vs_oracle(cuda):
    """Forward (raw predictions 1e-3), losses (2e-4), the gradient of EVERY head variable (autograd through the f64
    oracle), moving statistics and one TF-Adam step, f32 build. The backbone is frozen and runs on moving statistics."""
    from multiposenet_amd.retinanet import LEVELS, PersonDetectorNet, generate_anchors
    B, H, W = 2, 128, 256
    bb, hp, img, boxes, num = _setup(5, B, H, W)
    anchors, _ = generate_anchors(H, W)
    p64 = {k: torch.tensor(v, dtype=torch.float64, requires_grad=not (k.endswith("moving_mean") or k.endswith("moving_variance"))) for k, v in hp.items()}
    bb64 = {k: torch.tensor(v, dtype=torch.float64) for k, v in bb.items()}
    upd = {}
    enc, cls, _ = R.forward(torch.tensor(img, dtype=torch.float64), bb64, p64, True, updates=upd)
    tg = np.zeros((B, anchors.shape[0], 4), np.float32); mt = np.zeros((B, anchors.shape[0]), np.int32)
    for b in range(B):
        tg[b], mt[b] = R.get_training_targets(anchors, boxes[b, :num[b]])
    allv = dict(p64); allv.update({k: v for k, v in bb64.items() if k.startswith("MobilenetV1/")})
    total, ls = R.total_loss_fn(enc, cls, torch.tensor(tg, dtype=torch.float64), torch.tensor(mt), HP, allv)
    total.backward()

    net = PersonDetectorNet(backbone_values=bb, head_values=hp, dtype=torch.float32)
    bset = net.forward(torch.tensor(img).cuda(), True)
    raw = net.raw_predictions(bset)
    np.testing.assert_allclose(raw["encoded_boxes"].cpu().numpy(), enc.detach().numpy(), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(raw["class_predictions"].cpu().numpy(), cls.detach().numpy(), atol=1e-3, rtol=1e-3)
    net.create_targets({"boxes": torch.tensor(boxes).cuda(), "num_boxes": torch.tensor(num).cuda()})
    np.testing.assert_array_equal(bset["matches"].cpu().numpy(), mt)
    losses = net.compute_losses(HP).cpu().numpy()
    np.testing.assert_allclose(losses[0], float(ls["localization_loss"]), rtol=2e-4)
    np.testing.assert_allclose(losses[1], float(ls["classification_loss"]), rtol=2e-4)
    np.testing.assert_allclose(losses[3], float(total), rtol=2e-4)
    net.backward(HP["weight_decay"])
    bad = []
    for k, g in net.grads.items():
        want = p64[k].grad.numpy()
        err = float(np.abs(g.cpu().numpy() - want).max() / (np.abs(want).max() + 1e-30))
        if err > 2e-3:
            bad.append((k, err))
    assert not bad, sorted(bad, key=lambda kv: -kv[1])[:8]
    # moving statistics of the head (batch statistics of THIS step) and the optimizer step
    before = {k: v.clone() for k, v in net.vars.items()}
    grads = {k: v.cpu().numpy().copy() for k, v in net.grads.items()}
    net.optimizer_step(HP["initial_learning_rate"], HP["num_steps"])
    for k, v in upd.items():
        np.testing.assert_allclose(net.stats[k].cpu().numpy(), v.detach().numpy(), rtol=2e-4, atol=1e-5, err_msg=k)
    lr = onet.cosine_decay(HP["initial_learning_rate"], 0, HP["num_steps"])
    for k in ("fpn/p7/kernel", "class_net/logits/bias", "box_net/conv3x3_0/kernel", "p5_batch_norm/gamma"):
        pw = before[k].cpu().numpy().astype(np.float64)
        m, v = np.zeros_like(pw), np.zeros_like(pw)
        onet.adam_step(pw, grads[k].astype(np.float64), m, v, lr, 1, clip=np.inf)
        np.testing.assert_allclose(net.vars[k].cpu().numpy(), pw, rtol=1e-5, atol=1e-7, err_msg=k)
    assert int(net.global_step.item()) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_merged_first_tower_layer_equals_the_two_towers(cuda, monkeypatch, dtype):
    """The box and class towers' first convolutions read the same tensor (box_predictor.py:101-103 under both scopes): the build runs
    them as ONE 128 -> 128 convolution, one 128-channel batch-norm over adjacent variables, one weight gradient and ONE data gradient
    whose contraction over the merged channels is the sum the two towers send into p{l}_batch_norm (PersonDetectorNet.merge_tower0).
    Against the two towers run separately (MPN_RETINA_MERGE=0) on the same variables and batch: the same losses, predictions, moving
    statistics and gradients - f32 to summation order; bf16 to storage rounding (the sum of the two towers' gradients is formed in
    the accumulator instead of from two rounded tensors) - with and without the reductions fused into the data gradients."""
    from multiposenet_amd.retinanet import PersonDetectorNet
    B, H, W = 2, 128, 256
    bb, hp, img, boxes, num = _setup(21, B, H, W)
    gt = {"boxes": torch.tensor(boxes).cuda(), "num_boxes": torch.tensor(num).cuda()}

    def run(merge, fuse):
        monkeypatch.setenv("MPN_RETINA_MERGE", "1" if merge else "0")
        net = PersonDetectorNet(backbone_values=bb, head_values=hp, dtype=dtype)
        assert net.merge_tower0 == merge
        net.fuse_conv_bn = fuse
        bset = net.forward(torch.tensor(img).cuda(), True)
        raw = {k: v.float().cpu().numpy() for k, v in net.raw_predictions(bset).items()}
        net.create_targets(gt)
        losses = net.compute_losses(HP).cpu().numpy().copy()
        net.backward(HP["weight_decay"])
        return raw, losses, {k: v.cpu().numpy().copy() for k, v in net.grads.items()}, {k: v.cpu().numpy().copy() for k, v in net.stats.items()}

    ref = run(False, True)
    f32 = dtype == torch.float32
    for fuse in (True, False):
        raw, losses, grads, stats = run(True, fuse)
        for k in raw:
            np.testing.assert_allclose(raw[k], ref[0][k], atol=(1e-5 if f32 else 3e-2) * float(np.abs(ref[0][k]).max()), err_msg=k)
        np.testing.assert_allclose(losses, ref[1], rtol=1e-5 if f32 else 2e-2)
        for k, v in stats.items():
            np.testing.assert_allclose(v, ref[3][k], rtol=1e-4 if f32 else 2e-2, atol=1e-6 if f32 else 2e-3, err_msg=k)
        bad = []
        for k, g in grads.items():
            want = ref[2][k]
            err = float(np.linalg.norm((g - want).ravel()) / (np.linalg.norm(want.ravel()) + 1e-30))
            if err > (2e-5 if f32 else 0.12):
                bad.append((k, err))
        assert not bad, (fuse, sorted(bad, key=lambda kv: -kv[1])[:8])


def test_nms_replayed_from_a_hipgraph_with_every_anchor_a_candidate(cuda):
    """Round 5, a device fault: the candidate lists' counters were cleared by hipMemsetAsync; inside a REPLAYED hipGraph that memset
    node was seen to land after the first appends of the kernel behind it, so a list started at the previous call's count - with every
    anchor above the score threshold (count = A) it ran past the workspace. The counters are cleared by a launch now, an append
    past the list is refused AND reported in the workspace's overflow word, the selection never reads past a list. Eager call, capture,
    three replays, every anchor a candidate: identical detections each time, the overflow word clear."""
    from multiposenet_amd.retinanet import PersonDetectorNet
    B, H, W = 2, 128, 256
    bb, hp, img, _, _ = _setup(33, B, H, W)
    hp["class_net/logits/kernel"] = (np.random.RandomState(3).randn(3, 3, 64, 6) * 0.4).astype(np.float32)
    hp["class_net/logits/bias"] = np.full(6, -2.0, np.float32)
    net = PersonDetectorNet(backbone_values=bb, head_values=hp, dtype=torch.float32)
    b = net.forward(torch.tensor(img).cuda(), False)
    want = {k: v.cpu().numpy().copy() for k, v in net.nms(b, 0.0, 0.6, 25).items()}
    assert (want["num_boxes"] == 25).all()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = net.nms(b, 0.0, 0.6, 25)
    assert int(want["overflow"][0]) == 0
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        net.check_nms(out)                                  # the call's overflow word (mpn_retina_nms_overflow_offset): clear
        for k, v in want.items():
            np.testing.assert_array_equal(out[k].cpu().numpy(), v, err_msg=k)


def test_detector_inference_and_bf16_build(cuda):
    """get_predictions (is_training=False) of the f32 build vs the oracle's NMS on the oracle's raw predictions; the bf16 build
    trains (finite losses, loss close to the f32 build's)."""
    from multiposenet_amd.retinanet import PersonDetectorNet, generate_anchors
    B, H, W = 2, 128, 256
    bb, hp, img, boxes, num = _setup(9, B, H, W)
    hp["class_net/logits/bias"] = np.full(6, -1.0, np.float32)     # enough candidates above the threshold
    anchors, _ = generate_anchors(H, W)
    with torch.no_grad():
        enc, cls, _ = R.forward(torch.tensor(img), {k: torch.tensor(v) for k, v in bb.items()}, {k: torch.tensor(v) for k, v in hp.items()}, False)
    wb, ws, wn = R.get_predictions(enc.numpy(), cls.numpy(), anchors, 0.3, 0.6, 25)
    net = PersonDetectorNet(backbone_values=bb, head_values=hp, dtype=torch.float32)
    out = net.predict(torch.tensor(img).cuda(), 0.3, 0.6, 25)
    gn = out["num_boxes"].cpu().numpy()
    assert wn.sum() > 0
    # (the network's f32 logits differ from the oracle's in the last bits: compare the detections, not bit patterns)
    assert np.abs(gn - wn).max() <= 1
    k = int(min(gn[0], wn[0]))
    np.testing.assert_allclose(out["scores"].cpu().numpy()[0, :k], ws[0, :k], atol=2e-3)
    losses = {}
    for dt in (torch.float32, torch.bfloat16):
        n2 = PersonDetectorNet(backbone_values=bb, head_values=hp, dtype=dt)
        l = n2.train_step(torch.tensor(img).cuda(), {"boxes": torch.tensor(boxes).cuda(), "num_boxes": torch.tensor(num).cuda()}, HP)
        losses[dt] = l.cpu().numpy().copy()
        assert np.isfinite(losses[dt]).all() and np.isfinite(n2.theta.cpu().numpy()).all()
    np.testing.assert_allclose(losses[torch.bfloat16][:2], losses[torch.float32][:2], rtol=0.1)


def test_reference_surface_shims(cuda):
    """mobilenet_v1 -> RetinaNet(backbone_features, image_shape, is_training, params) with .anchors / .raw_predictions /
    .get_predictions() / .loss(), AnchorGenerator, get_training_targets and person_detector_model.model_fn keep the
    reference's names, arguments and result keys, and agree with the class they wrap."""
    from multiposenet_amd import person_detector_model as pdm, variables
    from multiposenet_amd.detector import AnchorGenerator, RetinaNet
    from multiposenet_amd.detector.backbones.mobilenet_v1 import mobilenet_v1
    from multiposenet_amd.detector.training_target_creation import get_training_targets
    from multiposenet_amd.keypoints_model import ModeKeys
    from multiposenet_amd.retinanet import PersonDetectorNet
    B, H, W = 1, 128, 256
    bb, hp, img, boxes, num = _setup(13, B, H, W)
    hp["class_net/logits/bias"] = np.full(6, -1.0, np.float32)
    net = variables.set_default_detector(PersonDetectorNet(backbone_values=bb, head_values=hp, dtype=torch.float32))
    images = torch.tensor(img).cuda()
    feats = mobilenet_v1(images, is_training=False, net=net.backbone)
    rn = RetinaNet(feats, images.shape, False, {"depth_multiplier": 1.0}, net=net)
    want = net.predict(images, 0.3, 0.6, 25)
    got = rn.get_predictions(score_threshold=0.3, iou_threshold=0.6, max_detections=25)
    for k in ("boxes", "scores", "num_boxes"):
        assert torch.equal(got[k], want[k]), k
    assert tuple(rn.raw_predictions["encoded_boxes"].shape) == (B, rn.anchors.shape[0], 4)
    anchors = AnchorGenerator()(H, W)
    np.testing.assert_array_equal(anchors, rn.anchors.cpu().numpy())
    gt = {"boxes": torch.tensor(boxes).cuda(), "num_boxes": torch.tensor(num).cuda()}
    ls = rn.loss(gt, {"gamma": 2.0, "alpha": 0.25})
    assert set(ls) == {"localization_loss", "classification_loss"} and all(np.isfinite(float(v)) for v in ls.values())
    # a second RetinaNet on the shared detector at ANOTHER image shape must not redirect the first object's loss
    img2 = torch.rand(1, 128, 128, 3, device="cuda")
    RetinaNet(mobilenet_v1(img2, is_training=False, net=net.backbone), img2.shape, False, {"depth_multiplier": 1.0}, net=net)
    ls2 = rn.loss(gt, {"gamma": 2.0, "alpha": 0.25})
    for k in ls:
        assert float(ls2[k]) == float(ls[k]), k
    t, m = get_training_targets(anchors, boxes[0, :num[0]], positives_threshold=0.5, negatives_threshold=0.5)
    wt, wm = R.get_training_targets(anchors, boxes[0, :num[0]])
    np.testing.assert_array_equal(m.cpu().numpy(), wm)
    np.testing.assert_allclose(t.cpu().numpy(), wt, rtol=2e-6, atol=2e-6)
    # model_fn: TRAIN applies a step, EVAL returns predictions + losses
    pdm.reset_registry()
    params = dict(HP, score_threshold=0.3, iou_threshold=0.6, max_boxes=25, dtype="f32", backbone_values=bb, head_values=hp, model_dir="t")
    s1 = pdm.model_fn({"images": img}, {"boxes": boxes, "num_boxes": num}, ModeKeys.TRAIN, params)
    assert s1.train_op is not None and np.isfinite(float(s1.loss)) and set(s1.losses) == {"localization_loss", "classification_loss", "regularization_loss", "total_loss"}
    s2 = pdm.model_fn({"images": img}, {"boxes": boxes, "num_boxes": num}, ModeKeys.EVAL, params)
    assert s2.train_op is None and set(s2.eval_metric_ops) == {"boxes", "scores", "num_boxes"}
    assert int(pdm.get_detector(params).global_step.item()) == 1
    with pytest.raises(AssertionError):
        pdm.model_fn({"images": img}, {"boxes": boxes, "num_boxes": num}, ModeKeys.PREDICT, params)


def test_detector_step_replays_from_a_hipgraph(cuda):
    """The whole TRAIN step (frozen backbone, head, matching, losses, backward, Adam) captured once and replayed: bit-identical
    to the eager step (no atomics on floats anywhere; the integer counters and ordered keys are order-free)."""
    from multiposenet_amd.retinanet import PersonDetectorNet
    B, H, W = 2, 128, 128
    bb, hp, img, boxes, num = _setup(17, B, H, W)
    images = torch.tensor(img).cuda()
    gt = {"boxes": torch.tensor(boxes).cuda(), "num_boxes": torch.tensor(num).cuda()}
    ref = PersonDetectorNet(backbone_values=bb, head_values=hp, dtype=torch.bfloat16)
    want = [ref.train_step(images, gt, HP).cpu().numpy().copy() for _ in range(3)]
    net = PersonDetectorNet(backbone_values=bb, head_values=hp, dtype=torch.bfloat16)
    first = net.train_step(images, gt, HP).cpu().numpy().copy()          # eager warm-up = step 1
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        losses = net.train_step(images, gt, HP)
    # (capture does not execute: steps 2 and 3 are the two replays)
    got = [first]
    for _ in range(2):
        g.replay()
        got.append(losses.cpu().numpy().copy())
    for a, b_ in zip(want, got):
        np.testing.assert_array_equal(a, b_)
    assert torch.equal(ref.theta, net.theta) and torch.equal(ref.moving, net.moving)


def test_p6_split_k_path_at_small_batch_equals_the_1x1_path(cuda):
    """fpn/p6 at inference with a few dozen output pixels runs as a split-K contraction (the PRN's trick) instead of ONE 128-pixel
    tile on one CU: the same p6 (bf16 rounding of an f32 sum in another order) and the same detections as the 1x1 path."""
    from multiposenet_amd.retinanet import PersonDetectorNet
    B, H, W = 1, 256, 384
    bb, hp, img, _, _ = _setup(23, B, H, W)
    hp["class_net/logits/kernel"] = (np.random.RandomState(8).randn(3, 3, 64, 6) * 0.4).astype(np.float32)
    hp["class_net/logits/bias"] = np.full(6, -2.0, np.float32)
    images = torch.tensor(img).cuda()
    out = {}
    for skinny in (True, False):
        net = PersonDetectorNet(backbone_values=bb, head_values=hp, dtype=torch.bfloat16)
        if not skinny:
            net._p6_skinny = lambda b: False
        else:
            assert net._p6_skinny(net._buffers(B, H, W))
        bset = net.forward(images, False)
        pred = net.nms(bset, 0.3, 0.6, 25)
        out[skinny] = (bset["p"][6].float().cpu().numpy().copy(), bset["p"][7].float().cpu().numpy().copy(),
                       int(pred["num_boxes"][0]), pred["scores"][0].cpu().numpy().copy())
    a, b_ = out[True], out[False]
    assert np.abs(a[0]).max() > 0
    np.testing.assert_allclose(a[0], b_[0], atol=1.2e-2 * np.abs(b_[0]).max(), rtol=1.2e-2)
    np.testing.assert_allclose(a[1], b_[1], atol=3e-2 * np.abs(b_[1]).max(), rtol=3e-2)
    assert abs(a[2] - b_[2]) <= 1
    k = min(a[2], b_[2])
    np.testing.assert_allclose(a[3][:k], b_[3][:k], atol=2e-2)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_detector_backward_pass_teacher_forced_against_the_oracle(cuda, dtype):
    """an ABSOLUTE bound on the gradients of the detector head's bf16 step (BASELINE config 4), as config 2 got in round
    5. What a plain comparison reads: round 5's last, uncommitted scratch run held the bf16 build's step against the f32 oracle at random
    initialisation on one 128 x 256 batch - "all rel-L2 0.31, every *_for_level_7 / p7 / pre_p7_bn tensor 1.04-1.28": batch statistics
    over 2 x 1 x 2 = 4 pixels (level 7) and ReLU masks move under bf16 storage of the FORWARD pass, and the backward pass amplifies that
    (tools/bf16_grad_bound.py measured the same on the keypoint net: the f64 oracle's own gradient moves 0.93 under bf16 storage). So
    the perturbation is taken out (tools/bf16_teacher_forced_detector.py): the build runs its forward pass, then every tensor its
    backward pass reads - the backbone features, the FPN sums, p3..p7, both stride-2 patch tensors, every tower layer's raw output, the
    raw box / class outputs, mean / invstd / scale / shift of all 46 batch-norm layers of the head - is overwritten with the emulating
    oracle's values (oracle/retinanet.py under onet.storage_emulation: exactly representable in bf16), and the build's matching, loss
    gradient and whole backward chain run from there, on head variables the f32 build trained for 40 steps and a batch they have not
    seen (2 @ 256 x 384: levels 3..7 = 32 x 48 .. 2 x 3 pixels). Every gradient tensor - the level-6 / level-7 ones included, no
    exception needed - within 10 % and cosine 0.99 of the oracle's (measured: 0.0-2.1 %, all 112 tensors together 0.37 %,
    profiles/r06_bf16_teacher_forced_detector.txt): a systematic error of 20 % in any backward kernel of the chain fails. The f32 build
    through the same machinery: 2e-3 (measured < 1e-5)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bf16_teacher_forced_detector as tf
    out = tf.run(steps=40, B=2, H=256, W=384, dtype=torch.bfloat16 if dtype == "bf16" else torch.float32, verbose=False)
    tol_rel, tol_cos = (0.10, 0.99) if dtype == "bf16" else (2e-3, 0.99999)
    np.testing.assert_allclose(out["loss"], out["oracle_loss"], rtol=1e-5 if dtype == "bf16" else 1e-6)
    assert len(out["rows"]) == 112                                     # every trainable head variable
    # (a tower level without a matched anchor has no box gradient at all: the oracle's is exactly zero and so must the build's be)
    bad = [(k, rel, cos) for k, _, norm, rel, cos in out["rows"] if not ((norm == 0.0 and rel == 0.0) or (rel <= tol_rel and cos >= tol_cos))]
    assert sum(1 for r in out["rows"] if r[2] > 0.0) >= 80              # (88 with this batch: three levels without a matched anchor)
    worst = sorted(out["rows"], key=lambda r: -r[3])[:4]
    print(f"\n[{dtype} detector backward, teacher-forced] all {len(out['rows'])} gradient tensors together: rel-L2 {out['all_rel']:.5f}, "
          f"cosine {out['all_cos']:.6f}; worst: {[(r[0], round(r[3], 4)) for r in worst]}")
    assert not bad, bad
    assert out["all_rel"] <= (0.02 if dtype == "bf16" else 5e-4)
    # ... and the bound is one a wrong kernel breaks: the same comparison with ONE tower layer's gradient scaled by 1.2 fails it
    k = "box_net/conv3x3_2/kernel"
    g, w_ = out["got"][k] * 1.2, out["want"][k]
    assert np.linalg.norm(g - w_) / np.linalg.norm(w_) > tol_rel
