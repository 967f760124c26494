"""Shared helpers for the GPU parity tests (oracle = torch-CPU restatement, see oracle/)."""
import numpy as np
import torch

DTYPES = [torch.float32, torch.bfloat16]


def rnd(x, dtype):
    """Round an f32 numpy array / tensor to the storage dtype and back to f32 (CPU tensor)."""
    t = torch.as_tensor(np.asarray(x), dtype=torch.float32) if not torch.is_tensor(x) else x.float().cpu()
    return t.to(dtype).float()


def dev(x, dtype=None):
    t = torch.as_tensor(np.asarray(x)) if not torch.is_tensor(x) else x
    if dtype is not None:
        t = t.to(dtype)
    elif t.dtype == torch.float64:
        t = t.float()
    return t.contiguous().cuda()


def nchw(x):
    return x.permute(0, 3, 1, 2)


def nhwc(x):
    return x.permute(0, 2, 3, 1)


def act_ref(x, act):
    from multiposenet_amd._lib import ACT_RELU, ACT_RELU6
    if act == ACT_RELU:
        return torch.relu(x)
    if act == ACT_RELU6:
        return torch.clamp(x, 0, 6)
    return x


def tol(dtype, k=1):
    """(rtol, atol) for a result that is a length-k f32-accumulated sum stored in `dtype`."""
    if dtype == torch.float32:
        return 2e-5, 2e-5 * max(1.0, k ** 0.5)
    if dtype == torch.float16:
        return 1.5e-3, 1.5e-3  # fp16 storage: 2^-11 relative on outputs of O(1)
    return 1.2e-2, 1.2e-2  # bf16 storage: 2^-8 relative on outputs of O(1)


def assert_close(got, want, dtype, k=1, scale=None):
    got = got.float().cpu()
    want = want.float().cpu()
    rtol, atol = tol(dtype, k)
    if scale is None:
        scale = float(want.abs().max()) or 1.0
    err = (got - want).abs()
    bound = atol * scale + rtol * want.abs()
    bad = err > bound
    assert not bool(bad.any()), (
        f"max err {float(err.max()):.3e} (scale {scale:.3e}, {int(bad.sum())}/{bad.numel()} outside tol) "
        f"at {np.unravel_index(int(err.argmax()), tuple(err.shape))}")


import os as _os

RENDER_GOLD = np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "render_goldens.npz"))


def render_golden(name):
    """Dense float32 [h,w,17] heatmaps the reference produced for a case of tests/golden/render_cases.py."""
    shape = tuple(int(s) for s in RENDER_GOLD[f"{name}/shape"])
    out = np.zeros(int(np.prod(shape)), np.float32)
    out[RENDER_GOLD[f"{name}/index"]] = RENDER_GOLD[f"{name}/value"]
    return out.reshape(shape)
