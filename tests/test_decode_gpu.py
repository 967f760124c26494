"""GPU: HIP heatmap decode (through the C ABI) vs the reference goldens and the oracle - bit-exact."""
import numpy as np
import pytest

from decode_cases import cases
from oracle import decode as oracle_decode

pytestmark = pytest.mark.gpu

GOLD = np.load(__file__.replace("test_decode_gpu.py", "golden/decode_goldens.npz"))
CASES = list(cases())


@pytest.mark.parametrize("name,hm,box,thr", CASES, ids=[c[0] for c in CASES])
def test_get_keypoints_matches_reference_goldens(cuda, name, hm, box, thr):
    from multiposenet_amd.inference.utils import get_keypoints
    got = get_keypoints(hm, box, thr)
    assert got.dtype == np.int32 and got.shape == (17, 3)
    np.testing.assert_array_equal(got, GOLD[f"{name}/keypoints"])


@pytest.mark.parametrize("B,h,w", [(1, 128, 128), (3, 64, 96), (32, 128, 128), (5, 33, 7), (2, 256, 256), (300, 16, 16)])
def test_batch_vs_oracle(cuda, B, h, w):
    import torch
    from multiposenet_amd.inference.utils import get_keypoints_batch
    rs = np.random.RandomState(B * 1000 + h)
    lg = rs.randn(B, h, w, 17).astype(np.float32) * 1.5 - 4.6
    hm = (1.0 / (1.0 + np.exp(-lg))).astype(np.float32)
    # quantise so that ties across distant pixels are common
    hm = np.round(hm * 64) / 64
    boxes = np.stack([np.array([0, 0, 4 * h + i, 4 * w - i]) for i in range(B)])
    xyv, score = get_keypoints_batch(torch.from_numpy(hm).cuda(), boxes, 0.2, return_scores=True)
    want = oracle_decode.get_keypoints_batch(hm, boxes, 0.2)
    np.testing.assert_array_equal(xyv.cpu().numpy(), want)
    mx, _ = oracle_decode.scores_and_indices(hm)
    np.testing.assert_array_equal(score.cpu().numpy(), mx)


def test_indices_and_workspace_reuse(cuda):
    import torch
    from multiposenet_amd.inference.utils import KeypointDecoder
    rs = np.random.RandomState(7)
    B, h, w = 8, 128, 128
    dec = KeypointDecoder(B)
    box_hw = torch.tensor([[512.0, 512.0]] * B, dtype=torch.float64, device="cuda")
    for it in range(3):  # the kernel must leave its workspace zeroed
        hm = rs.rand(B, h, w, 17).astype(np.float32)
        xyv, score, index = dec(torch.from_numpy(hm).cuda(), box_hw, 0.2)
        mx, idx = oracle_decode.scores_and_indices(hm)
        np.testing.assert_array_equal(index.cpu().numpy(), idx)
        np.testing.assert_array_equal(score.cpu().numpy(), mx)
        assert int(dec.workspace.to(torch.int32).abs().sum()) == 0


def test_bf16_and_fp16_inputs(cuda):
    import torch
    from multiposenet_amd.inference.utils import get_keypoints_batch
    rs = np.random.RandomState(3)
    hm32 = rs.rand(4, 64, 64, 17).astype(np.float32)
    boxes = np.array([[0, 0, 256, 256]] * 4)
    for td in (torch.bfloat16, torch.float16):
        t = torch.from_numpy(hm32).to(td)
        ref = t.float().numpy()  # exact values the kernel sees
        got = get_keypoints_batch(t.cuda(), boxes, 0.5).cpu().numpy()
        thr = float(torch.tensor(0.5, dtype=td).float())
        np.testing.assert_array_equal(got, oracle_decode.get_keypoints_batch(ref, boxes, thr))


def test_rejects_bad_shapes(cuda):
    import torch
    from multiposenet_amd.inference.utils import get_keypoints
    with pytest.raises(ValueError):
        get_keypoints(np.zeros((8, 8, 16), np.float32), np.array([0, 0, 8, 8]), 0.1)
