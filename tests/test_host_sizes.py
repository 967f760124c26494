"""CPU: the launchers' size formulas at the full sizes of BASELINE configs 2, 4 and 5 (no GPU, no compute call). These are the
`long long` / `size_t` / `int` products that only the GPU tests exercised before; tools/sanitize_host.sh runs this file (with
test_abi.py and test_host_logic.py) against an ASan + UBSan build of the library's host half."""
import ctypes

import numpy as np
import pytest

from multiposenet_amd import _lib

L = _lib.lib
BF16, F32, F16 = _lib.MPN_BF16, _lib.MPN_F32, _lib.MPN_F16
ANCHORS = 157542           # BASELINE config 4: 896 x 1408, levels 3..7, 9 anchors per cell (tests/test_retinanet_oracle.py)


def _sz(name, *args, restype=ctypes.c_size_t):
    f = getattr(L(), name)
    f.restype = restype
    return f(*args)


def test_config4_detector_workspaces_at_157542_anchors():
    lib = L()
    lib.mpn_retina_match_workspace_bytes.restype = ctypes.c_size_t
    lib.mpn_retina_nms_workspace_bytes.restype = ctypes.c_size_t
    m16 = lib.mpn_retina_match_workspace_bytes(16, 100)
    assert m16 >= 16 * 100 * 8 and m16 < (1 << 34)
    n16 = lib.mpn_retina_nms_workspace_bytes(16, ANCHORS)
    assert n16 >= 16 * ANCHORS * 4 and n16 < (1 << 36)
    # the formulas are monotone in the batch and do not wrap a 32-bit product: 512 images x 157 542 anchors x 4 bytes > 2^28
    assert lib.mpn_retina_nms_workspace_bytes(512, ANCHORS) >= 32 * n16 - 4096
    # the overflow word (ABI 600): inside the workspace, behind the candidate lists and the per-image counters, 4-byte aligned - at any batch
    lib.mpn_retina_nms_overflow_offset.restype = ctypes.c_size_t
    for b in (1, 2, 16, 17, 512):
        off, size = lib.mpn_retina_nms_overflow_offset(b, ANCHORS), lib.mpn_retina_nms_workspace_bytes(b, ANCHORS)
        assert off == b * ANCHORS * 32 + b * 4 and off % 4 == 0 and off + 4 <= size and size % 16 == 0
    assert lib.mpn_retina_match_workspace_bytes(4096, 1000) >= 4096 * 1000 * 8
    parts = lib.mpn_retina_loss_num_parts(16, ANCHORS)
    assert 0 < parts <= 16 * ANCHORS
    assert lib.mpn_retina_loss_num_parts(512, ANCHORS) >= parts


def test_config2_partial_row_counts_at_batch_32_512x512():
    lib = L()
    # the stem output is 32 x 256 x 256 x 32, the first pointwise layer leaves 16 384 partial rows
    assert lib.mpn_conv_num_parts(32, 256, 256, 1) == 16384
    assert lib.mpn_conv_num_parts(32, 128, 128, 3) == 32 * 16 * 8
    for (h, c, s) in [(256, 32, 1), (256, 64, 2), (128, 128, 1), (128, 128, 2), (64, 256, 1), (64, 256, 2), (32, 512, 1), (32, 512, 2), (16, 1024, 1)]:
        for f in ("mpn_dwconv_num_parts", "mpn_dwconv_bwd_data_bn_num_parts", "mpn_dwconv_wgrad_num_parts"):
            n = getattr(lib, f)(32, h, h, c, s, BF16)
            assert 0 < n < (1 << 24), (f, h, c, s, n)
    lib.mpn_bn_stats_num_parts.argtypes = [ctypes.c_longlong]
    assert 0 < lib.mpn_bn_stats_num_parts(32 * 256 * 256) <= 32 * 256 * 256
    assert lib.mpn_bn_stats_num_parts(1 << 40) > 0                      # a row count beyond 32 bits stays positive
    lib.mpn_heatmap_head_bwd_num_parts.argtypes = [ctypes.c_longlong]
    assert 0 < lib.mpn_heatmap_head_bwd_num_parts(32 * 128 * 128) < (1 << 24)
    assert 0 < lib.mpn_keypoint_loss_num_parts(32, 128, 128) < (1 << 24)
    assert 0 < lib.mpn_stem_conv_fwd_num_parts(32, 512, 512, 32, BF16) < (1 << 24)
    assert 0 < lib.mpn_stem_conv_wgrad_num_parts(32, 512, 512) < (1 << 24)


def test_weight_gradient_slabs_and_packed_weights_do_not_wrap():
    lib = L()
    lib.mpn_conv_packed_bytes.restype = ctypes.c_size_t
    for (cin, cout, k) in [(128, 128, 3), (512, 64, 3), (64, 512, 3), (1024, 1024, 1), (512, 512, 1), (256, 256, 3)]:
        for dt, es in ((BF16, 2), (F32, 4)):
            b = lib.mpn_conv_packed_bytes(cin, cout, k, 0, dt)
            assert b >= k * k * cin * cout * es, (cin, cout, k, dt, b)
    # split-K parts x the slab of one part stays below 2^31 floats for every layer of the step (the batched reduction indexes in int)
    for (h, cin, cout, k) in [(128, 128, 128, 3), (128, 512, 64, 3), (256, 32, 64, 1), (32, 512, 512, 1), (16, 1024, 1024, 1)]:
        n = lib.mpn_conv_wgrad_num_parts(32, h, h, cin, cout, k, BF16)
        assert 0 < n and n * k * k * cin * cout < (1 << 31), (h, cin, cout, k, n)
    # grouped: four pyramid levels of a subnet stage
    H = (ctypes.c_int * 4)(128, 64, 32, 16)
    nparts = (ctypes.c_int * 4)()
    assert lib.mpn_conv_wgrad_grouped_num_parts(4, 32, H, H, 128, 128, 3, BF16, nparts) == 0
    assert all(0 < n for n in nparts) and sum(nparts) <= 1024, list(nparts)
    assert lib.mpn_conv_wgrad_grouped_num_parts(4, 32, H, H, 128, 128, 3, BF16, None) != 0      # a NULL table is refused, not written


def test_decode_render_and_l2_workspaces():
    lib = L()
    lib.mpn_heatmap_decode_workspace_bytes.restype = ctypes.c_size_t
    lib.mpn_heatmap_render_workspace_bytes.restype = ctypes.c_size_t
    lib.mpn_l2_loss_batched_workspace_bytes.restype = ctypes.c_size_t
    assert lib.mpn_heatmap_decode_workspace_bytes(1 << 20) >= (1 << 20) * 17 * 8
    assert lib.mpn_heatmap_render_workspace_bytes(1 << 16) >= (1 << 16) * 17 * 3 * 4
    n = (ctypes.c_longlong * 3)(5521492, 1 << 33, 7)
    assert lib.mpn_l2_loss_batched_workspace_bytes(3, n) > 0
    assert lib.mpn_gemm_nt_num_parts(56 * 36 * 17) > 0
