"""GPU: the reference-shaped Python surface (mobilenet_v1, feature_pyramid_network, KeypointSubnet, Detector)."""
import numpy as np
import pytest
import torch

from oracle import network as onet

pytestmark = pytest.mark.gpu


def _lively_head(params, seed):
    """heatmaps/kernel ~ N(0, 1e-4) and bias -ln 99 at initialisation (keypoint_subnet.py:41-53) keep every heatmap near 0.01:
    below the 0.2 mask of create_pb.py:95-99, i.e. every crop zero and every PRN map a tie. A trained-like head spreads the
    logits over a few units so that the crops, the PRN and the arg-max downstream see real data."""
    rs = np.random.RandomState(seed)
    params["heatmaps/kernel"] = (rs.randn(1, 1, 64, 18) * 0.35).astype(np.float32)
    params["heatmaps/bias"] = np.concatenate([np.full(17, -1.0), [0.0]]).astype(np.float32)
    return params


def _net(dtype=torch.float32, seed=5, lively=False):
    from multiposenet_amd.net import KeypointNet
    params = onet.randomize_bn(onet.init_params(seed), seed + 1)
    if lively:
        _lively_head(params, seed)
    return KeypointNet(values=params, dtype=dtype), params


def test_mobilenet_fpn_subnet_functions_match_oracle(cuda):
    from multiposenet_amd.detector import KeypointSubnet, feature_pyramid_network
    from multiposenet_amd.detector.backbones import mobilenet_v1
    net, params = _net()
    img = np.random.RandomState(0).rand(1, 128, 256, 3).astype(np.float32)   # non-square, multiples of 128
    pt = {k: torch.tensor(v) for k, v in params.items()}
    with torch.no_grad():
        ref_feats = onet.mobilenet_v1(torch.tensor(img), pt, False)
        ref_fpn = onet.feature_pyramid_network(ref_feats, pt)
        ref_heat, ref_enr = onet.keypoint_subnet(ref_feats, pt, False)
    feats = mobilenet_v1(img, False, 1.0, net=net)
    assert sorted(feats) == ["c2", "c3", "c4", "c5"] and feats["c5"].shape == (1, 1024, 4, 8)
    for k in feats:
        np.testing.assert_allclose(feats[k].numpy(), ref_feats[k].numpy(), atol=2e-3, rtol=1e-3)
    fpn = feature_pyramid_network(feats, False, 128, min_level=2, add_coarse_features=False, scope="keypoint_fpn", net=net)
    for k in fpn:
        np.testing.assert_allclose(fpn[k].numpy(), ref_fpn[k].numpy(), atol=2e-3, rtol=1e-3)
    with pytest.raises(NotImplementedError):
        feature_pyramid_network(feats, False, 256, net=net)
    sub = KeypointSubnet(feats, False, {}, net=net)
    assert tuple(sub.heatmaps.shape) == (1, 32, 64, 18)
    np.testing.assert_allclose(sub.heatmaps.cpu().numpy(), ref_heat.numpy(), atol=2e-3, rtol=1e-3)
    np.testing.assert_allclose(sub.enriched_features["p3"].float().cpu().numpy(), ref_enr["p3"].numpy(), atol=2e-3, rtol=1e-3)


def test_detector_call_and_decode(cuda, tmp_path):
    from multiposenet_amd.inference import Detector, get_keypoints
    net, params = _net(seed=7)
    path = tmp_path / "weights.npz"
    np.savez(path, **net.state_dict())
    det = Detector(str(path), dtype=torch.float32)
    img = np.random.RandomState(1).randint(0, 256, (128, 128, 3)).astype(np.uint8)
    out = det(img)
    assert set(out) == {"boxes", "scores", "num_boxes", "keypoint_heatmaps", "segmentation_masks", "keypoint_scores",
                        "keypoint_positions"}
    assert out["keypoint_heatmaps"].shape == (32, 32, 17) and out["segmentation_masks"].shape == (32, 32)
    pt = {k: torch.tensor(v) for k, v in params.items()}
    with torch.no_grad():
        heat, _ = onet.forward(torch.tensor(img[None].astype(np.float32) * np.float32(1 / 255.0)), pt, False)
    np.testing.assert_allclose(out["keypoint_heatmaps"], torch.sigmoid(heat[0, ..., :17]).numpy(), atol=1e-3)
    np.testing.assert_allclose(out["segmentation_masks"], heat[0, ..., 17].numpy(), atol=2e-3, rtol=1e-3)
    # peak decode of the network's own heatmaps: bit-exact against the oracle decode of the SAME heatmaps
    from oracle import decode as odec
    box = np.array([0, 0, 128, 128])
    np.testing.assert_array_equal(get_keypoints(out["keypoint_heatmaps"], box, 0.005),
                                  odec.get_keypoints(out["keypoint_heatmaps"], box, 0.005))
    with pytest.raises(AssertionError):
        det(np.zeros((100, 128, 3), np.uint8))


def _decided_prn_positions(det, crops, logits_oracle):
    """Which (person, channel) arg-max positions of the PRN are DECIDED: the HIP PRN's logits on the same crops against the
    f64 oracle's - a channel is decided when the oracle's top-2 logit gap exceeds twice the largest logit difference (inside
    that margin the arg-max of the reference itself is not determined: VERDICT r3 "weak" 3 - a computed set, not a flat 10 %).
    Returns (mask [n,17], largest |logit difference|)."""
    net = det.assigner.net
    n = len(crops)
    x = np.zeros((net.valid,) + crops.shape[1:], np.float32)
    x[:n] = crops
    hip = net.predict(torch.tensor(x, device=net.device)).cpu().numpy()[:n]
    err = float(np.abs(hip - logits_oracle).max())
    flat = logits_oracle.reshape(n, -1, logits_oracle.shape[-1])
    part = np.partition(flat, flat.shape[1] - 2, axis=1)
    return (part[:, -1, :] - part[:, -2, :]) > 2 * err, err


def test_detector_with_prn_assigns_keypoints_to_given_boxes(cuda, tmp_path):
    """Detector(model_path, prn_path=...)(image, boxes=...) = create_pb.py:86-142 on caller-provided person boxes."""
    from multiposenet_amd.inference import Detector
    from multiposenet_amd.prn import initial_values
    from oracle import prn_post as opost, prn as oprn
    net, params = _net(seed=7, lively=True)
    wpath, ppath = tmp_path / "weights.npz", tmp_path / "prn.npz"
    np.savez(wpath, **net.state_dict())
    pvals = initial_values(seed=5)
    np.savez(ppath, **pvals)
    det = Detector(str(wpath), dtype=torch.float32, prn_path=str(ppath), max_boxes=8)
    img = np.random.RandomState(2).randint(0, 256, (128, 128, 3)).astype(np.uint8)
    boxes = np.array([[0.1, 0.1, 0.9, 0.6], [0.3, 0.4, 0.8, 0.95], [0.0, 0.0, 1.0, 1.0]], np.float32)
    scores = np.array([0.9, 0.01, 0.5], np.float32)
    out = det(img, score_threshold=0.05, boxes=boxes, scores=scores)
    # the 0.01 box is filtered (inference/detector.py:54-59); 'num_boxes' stays the graph's count, as in the reference
    assert out["num_boxes"] == 3 and out["boxes"].shape == (2, 4)
    assert out["keypoint_scores"].shape == (2, 17) and out["keypoint_positions"].shape == (2, 17, 2)
    # the same chain on the CPU restatements, from the detector's own heatmaps
    hm = out["keypoint_heatmaps"][None]
    norm, _, _ = opost.normalize_heatmaps(hm)
    crops = opost.crop_and_resize(norm, out["boxes"], np.zeros(2, np.int32), (56, 36))
    pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in pvals.items()}
    logits = oprn.prn(torch.tensor(crops, dtype=torch.float64), pt).numpy().astype(np.float32)
    ws, wp = opost.decode(logits)
    np.testing.assert_allclose(out["keypoint_scores"], ws, rtol=5e-3)
    decided, err = _decided_prn_positions(det, crops, logits)
    print(f"\n[PRN positions, given boxes] decided {int(decided.sum())} of {decided.size} channels, max |logit diff| {err:.2e}")
    assert err < 1e-3
    assert np.any(crops != 0) and decided.mean() >= 0.5
    assert np.all(out["keypoint_positions"] == wp, axis=-1)[decided].all()
    assert det(img)["keypoint_positions"].shape == (0, 17, 2)                # without boxes: empty, as before


def test_detector_joint_graph_matches_the_oracle_chain(cuda, tmp_path):
    """Detector(model_path, detector_path=, prn_path=) = the joint graph of create_pb.py:44-153: ONE backbone pass under the
    keypoint subnet and the RetinaNet head, NMS (0.3 / 0.6 / 25), crops of the min-max-normalised heatmaps, PRN, argmax_2d -
    all seven outputs against the chain of CPU restatements (oracle/network + retinanet + prn_post + prn) on one 256x384 image."""
    from multiposenet_amd.inference import Detector
    from multiposenet_amd.prn import initial_values
    from multiposenet_amd.retinanet import generate_anchors
    from oracle import prn as oprn, prn_post as opost, retinanet as R
    from test_retinanet_gpu import _setup
    H, W = 256, 384
    bb, hp, _, _, _ = _setup(31, 1, H, W)
    _lively_head(bb, 31)
    # lively class logits: enough candidates above the 0.3 threshold, scores spread out (random-init towers give nearly
    # equal scores, and NMS order among near-ties is not a property of the graph)
    hp["class_net/logits/kernel"] = (np.random.RandomState(8).randn(3, 3, 64, 6) * 0.4).astype(np.float32)
    hp["class_net/logits/bias"] = np.full(6, -2.0, np.float32)
    pvals = initial_values(seed=5)
    kpath, dpath, ppath = tmp_path / "keypoints.npz", tmp_path / "detector.npz", tmp_path / "prn.npz"
    np.savez(kpath, **bb); np.savez(dpath, **hp); np.savez(ppath, **pvals)
    det = Detector(str(kpath), dtype=torch.float32, detector_path=str(dpath), prn_path=str(ppath))
    img = np.random.RandomState(4).randint(0, 256, (H, W, 3)).astype(np.uint8)
    out = det(img, score_threshold=0.0)
    assert set(out) == {"boxes", "scores", "num_boxes", "keypoint_heatmaps", "segmentation_masks", "keypoint_scores", "keypoint_positions"}
    # ---- the oracle chain
    x = torch.tensor(img[None].astype(np.float32) * np.float32(1 / 255.0))
    with torch.no_grad():
        heat, _ = onet.forward(x, {k: torch.tensor(v) for k, v in bb.items()}, False)
        enc, cls, _ = R.forward(x, {k: torch.tensor(v) for k, v in bb.items()}, {k: torch.tensor(v) for k, v in hp.items()}, False)
    anchors, _ = generate_anchors(H, W)
    wb, ws, wn = R.get_predictions(enc.numpy(), cls.numpy(), anchors, 0.3, 0.6, 25)
    whm = torch.sigmoid(heat[0, ..., :17]).numpy()
    np.testing.assert_allclose(out["keypoint_heatmaps"], whm, atol=1e-3)
    np.testing.assert_allclose(out["segmentation_masks"], heat[0, ..., 17].numpy(), atol=2e-3, rtol=1e-3)
    n = int(wn[0])
    assert n >= 3 and int(out["num_boxes"]) == n
    # detections identical wherever the oracle's scores are not within 1e-3 of a neighbour (a near-tie may swap two slots)
    gaps = np.abs(np.diff(ws[0, :n]))
    decided = np.concatenate([[True], gaps > 1e-3]) & np.concatenate([gaps > 1e-3, [True]])
    assert decided.sum() >= 3, (decided, ws[0, :n])
    np.testing.assert_allclose(out["scores"][decided], ws[0, :n][decided], atol=1e-3)
    np.testing.assert_allclose(out["boxes"][decided], wb[0, :n][decided], atol=2e-3)
    # keypoint assignment: the restatement chain on the detector's own heatmaps and boxes
    norm, _, _ = opost.normalize_heatmaps(out["keypoint_heatmaps"][None])
    crops = opost.crop_and_resize(norm, out["boxes"], np.zeros(len(out["boxes"]), np.int32), (56, 36))
    pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in pvals.items()}
    wlogits = oprn.prn(torch.tensor(crops, dtype=torch.float64), pt).numpy().astype(np.float32)
    wsc, wpos = opost.decode(wlogits)
    assert out["keypoint_scores"].shape == (n, 17) and out["keypoint_positions"].shape == (n, 17, 2)
    np.testing.assert_allclose(out["keypoint_scores"], wsc, rtol=5e-3)
    decided, err = _decided_prn_positions(det, crops, wlogits)
    print(f"\n[PRN positions, joint graph] decided {int(decided.sum())} of {decided.size} channels, max |logit diff| {err:.2e}")
    assert err < 1e-3 and np.any(crops != 0) and decided.mean() >= 0.5
    assert np.all(out["keypoint_positions"] == wpos, axis=-1)[decided].all()
    # the score filter of inference/detector.py:54-59 on top of the graph's outputs
    thr = float(np.median(out["scores"]))
    flt = det(img, score_threshold=thr)
    keep = out["scores"] > thr
    assert int(flt["num_boxes"]) == n and len(flt["boxes"]) == int(keep.sum()) < n
    np.testing.assert_array_equal(flt["keypoint_positions"], out["keypoint_positions"][keep])
    # ONE backbone: the detector's head holds no MobileNet of its own
    assert det.retinanet.backbone is det.net
    # the device side replays from a hipGraph captured per image shape: the same outputs as the eager call, bit for bit, also
    # for a second image through the same graph
    eager = Detector(str(kpath), dtype=torch.float32, detector_path=str(dpath), prn_path=str(ppath))
    eager.use_graph = False
    img2 = np.random.RandomState(5).randint(0, 256, (H, W, 3)).astype(np.uint8)
    for im in (img, img2):
        a, b_ = det(im, score_threshold=0.0), eager(im, score_threshold=0.0)
        for k in a:
            np.testing.assert_array_equal(a[k], b_[k], err_msg=k)
    assert len(det._graphs) == 1 and not eager._graphs
    # bf16 build of the same graph: runs, finite, the same number of outputs
    o16 = Detector(str(kpath), dtype=torch.bfloat16, detector_path=str(dpath), prn_path=str(ppath))(img, score_threshold=0.0)
    assert np.isfinite(o16["keypoint_heatmaps"]).all() and len(o16["boxes"]) == int(o16["num_boxes"]) > 0
    assert o16["keypoint_positions"].shape == (len(o16["boxes"]), 17, 2)


def test_detector_graph_follows_reloaded_and_trained_variables(cuda, tmp_path):
    """the batch-norm inference affines and the p6 operand cast are host-cached and not part of the
    captured graph. After load_state_dict on either net - and after a replayed train step on the shared backbone - a
    replay must give what a freshly built eager Detector gives on the new variables, bit for bit."""
    from multiposenet_amd.inference import Detector
    from multiposenet_amd.prn import initial_values
    from multiposenet_amd.synthetic import synthetic_batch
    from multiposenet_amd.train import Trainer
    from test_retinanet_gpu import _setup
    H, W = 128, 256
    paths = {}
    for tag, seed in (("a", 31), ("b", 47)):
        bb, hp, _, _, _ = _setup(seed, 1, H, W)
        hp["class_net/logits/kernel"] = (np.random.RandomState(seed).randn(3, 3, 64, 6) * 0.4).astype(np.float32)
        hp["class_net/logits/bias"] = np.full(6, -2.0, np.float32)
        kp, dp = tmp_path / f"k{tag}.npz", tmp_path / f"d{tag}.npz"
        np.savez(kp, **bb); np.savez(dp, **hp)
        paths[tag] = (str(kp), str(dp), bb, hp)
    ppath = tmp_path / "prn.npz"
    np.savez(ppath, **initial_values(seed=5))
    img = np.random.RandomState(4).randint(0, 256, (H, W, 3)).astype(np.uint8)

    def eager_reference(kp, dp):
        e = Detector(kp, dtype=torch.float32, detector_path=dp, prn_path=str(ppath))
        e.use_graph = False
        return e(img, score_threshold=0.0)

    det = Detector(paths["a"][0], dtype=torch.float32, detector_path=paths["a"][1], prn_path=str(ppath))
    first = det(img, score_threshold=0.0)                                   # captures the graph on variables A
    for k, v in eager_reference(*paths["a"][:2]).items():
        np.testing.assert_array_equal(first[k], v, err_msg=k)
    # reload BOTH variable sets through the objects the graph was captured over
    det.net.load_state_dict(paths["b"][2])
    own = set(det.retinanet.vars) | set(det.retinanet.stats)
    det.retinanet.load_state_dict({k: v for k, v in paths["b"][3].items() if k in own})
    second = det(img, score_threshold=0.0)
    assert len(det._graphs) == 1                                            # the same graph, refreshed caches
    want = eager_reference(*paths["b"][:2])
    assert not np.array_equal(first["keypoint_heatmaps"], second["keypoint_heatmaps"])
    for k, v in want.items():
        np.testing.assert_array_equal(second[k], v, err_msg=k)
    # a train step replayed from the Trainer's hipGraph on the SHARED backbone: no Python of the net runs in a replay
    trainer = Trainer(det.net, {"initial_learning_rate": 1e-2, "num_steps": 1000, "weight_decay": 0.0}, use_graph=True)
    feats, labels = synthetic_batch(2, 128, 128, rank=0, device=det.net.device)
    for _ in range(3):
        trainer.step(feats, labels)
    third = det(img, score_threshold=0.0)
    kp3 = tmp_path / "k3.npz"
    np.savez(kp3, **det.net.state_dict())
    want3 = eager_reference(str(kp3), paths["b"][1])
    assert not np.array_equal(third["keypoint_heatmaps"], second["keypoint_heatmaps"])
    for k, v in want3.items():
        np.testing.assert_array_equal(third[k], v, err_msg=k)


def test_detector_head_inference_follows_replayed_train_steps(cuda):
    """ADVICE r3 case (b): eval, train steps replayed from a hipGraph, eval - the second eval scores with the trained
    variables' affines (the head's cache is opt-in and off here)."""
    from multiposenet_amd.retinanet import PersonDetectorNet
    from test_retinanet_gpu import _setup, HP
    B, H, W = 2, 128, 256
    bb, hp, img, boxes, num = _setup(5, B, H, W)
    net = PersonDetectorNet(backbone_values=bb, head_values=hp, dtype=torch.float32)
    x = torch.tensor(img, device="cuda")
    gt = {"boxes": torch.tensor(boxes, device="cuda"), "num_boxes": torch.tensor(num, device="cuda")}
    params = dict(HP, initial_learning_rate=1e-2, num_steps=1000)
    b0 = net.forward(x, False)
    before = net.raw_predictions(b0)["class_predictions"].clone()
    net.train_step(x, gt, params)                                            # eager warm-up
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        net.train_step(x, gt, params)
    for _ in range(3):
        g.replay()
    got = net.raw_predictions(net.forward(x, False))["class_predictions"].clone()
    fresh = PersonDetectorNet(backbone_values=bb, head_values={k: v for k, v in net.state_dict().items()}, dtype=torch.float32)
    want = fresh.raw_predictions(fresh.forward(x, False))["class_predictions"]
    assert not torch.equal(got, before)
    assert torch.equal(got, want)


@pytest.mark.parametrize("image_dtype", [torch.float32, torch.uint8], ids=["f32", "u8"])
def test_host_batch_feeder_equals_device_resident_steps(cuda, image_dtype):
    """HostBatchFeeder (pinned slots, copies on a side stream, the previous step still running): four steps on four
    different host batches give bit for bit the losses and variables of the same steps fed from device tensors."""
    from multiposenet_amd.input_feed import HostBatchFeeder
    from multiposenet_amd.net import KeypointNet
    from multiposenet_amd.train import Trainer
    rs = np.random.RandomState(21)
    B, H, W = 2, 128, 128
    hp = {"initial_learning_rate": 3e-4, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}

    def batch():
        img = rs.rand(B, H, W, 3).astype(np.float32)
        if image_dtype == torch.uint8:
            img = (img * 255).astype(np.uint8)
        heat = (rs.rand(B, H // 4, W // 4, 17) * 0.9).astype(np.float32)
        heat.reshape(B, -1)[:, rs.randint(0, heat[0].size, 8)] = 1.0
        lab = {"heatmaps": heat, "loss_masks": (rs.rand(B, H // 4, W // 4) < 0.9).astype(np.float32),
               "segmentation_masks": (rs.rand(B, H // 4, W // 4) < 0.3).astype(np.float32),
               "num_boxes": rs.randint(1, 5, size=B).astype(np.int64)}       # int64 as numpy makes it: cast by feed()
        return {"images": img}, lab

    batches = [batch() for _ in range(4)]
    a = Trainer(KeypointNet(dtype=torch.bfloat16, seed=3), hp, use_graph=True)
    b = Trainer(KeypointNet(dtype=torch.bfloat16, seed=3), hp, use_graph=True)
    want = []
    for f, l in batches:
        df = {"images": torch.from_numpy(f["images"]).cuda()}
        dl = {k: torch.from_numpy(v).cuda().to(torch.int32 if k == "num_boxes" else torch.float32) for k, v in l.items()}
        want.append(a.step(df, dl).cpu().numpy().copy())
    feeder = HostBatchFeeder(b, B, H, W, depth=2, image_dtype=image_dtype)
    got = []
    feeder.feed(*batches[0])
    for i in range(4):
        if i + 1 < 4:
            feeder.feed(*batches[i + 1])          # copy of batch i+1 runs beside step i
        assert feeder.pending() >= 1
        got.append(feeder.train_step().cpu().numpy().copy())
    with pytest.raises(RuntimeError):
        feeder.train_step()                        # nothing submitted
    for w_, g_ in zip(want, got):
        np.testing.assert_array_equal(w_, g_)
    sa, sb = a.net.state_dict(), b.net.state_dict()
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    # slots are filled in place too
    slot = feeder.acquire()
    arrs = feeder.slot_arrays(slot)
    assert arrs["images"].shape == (B, H, W, 3) and arrs["num_boxes"].dtype == np.int32
    feeder.submit(slot)
    with pytest.raises(RuntimeError):
        feeder.acquire(); feeder.acquire()         # depth 2: one more is free, the third is not
