"""CPU: host-side logic that needs no GPU - variable inventory, arena layout, bucketing, sharding, 2-rank gloo all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import network as onet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_variable_inventory_matches_oracle_and_reference_names():
    from multiposenet_amd import net
    a, b = net.variable_shapes(1.0), onet.param_shapes(1.0)
    assert list(a.items()) == list(b.items())
    assert sum(int(np.prod(s)) for k, s in a.items() if net.is_trainable(k)) == 5521490
    assert "MobilenetV1/Conv2d_13_pointwise/BatchNorm/moving_variance" in a and "keypoint_fpn/lateral2/kernel" in a
    init = net.initial_values(0)
    assert abs(init["heatmaps/bias"][0] + np.log(99.0)) < 1e-6 and init["heatmaps/bias"][17] == 0.0


def test_arena_layout_is_16_byte_aligned_and_head_adjacent():
    from multiposenet_amd import net
    shapes = {k: v for k, v in net.variable_shapes(1.0).items() if net.is_trainable(k)}
    arena = net._Arena(shapes, "cpu")
    flat = arena.new()
    views = arena.views(flat)
    for k, (off, n, shape) in arena.offsets.items():
        assert off % 4 == 0 and tuple(views[k].shape) == tuple(shape)
    ok, nk, _ = arena.offsets["heatmaps/kernel"]
    ob, _, _ = arena.offsets["heatmaps/bias"]
    assert ob == ok + nk
    assert arena.size % 4 == 0 and arena.size >= 5521490


def test_bucket_bounds_and_shards():
    from multiposenet_amd.parallel import bucket_bounds, shard_range
    b = bucket_bounds(5521700, 2 << 20)
    assert b[0][1] == 5521700 and b[-1][0] == 0
    assert all(x[0] == y[1] for x, y in zip(b[:-1], b[1:]))     # contiguous, last bucket first
    assert all((e - s) % 4 == 0 for s, e in b[:-1])
    assert sum(e - s for s, e in b) == 5521700
    assert shard_range(256, 3, 8) == (96, 128)
    with pytest.raises(ValueError):
        shard_range(250, 0, 8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from multiposenet_amd.parallel import GradientAllReducer, init_distributed
    r, _, w = init_distributed("gloo")
    torch.manual_seed(100 + r)
    flat = torch.randn(100003 * 4)           # rank-dependent "gradients"
    mine = flat.clone()
    red = GradientAllReducer(flat, bucket_bytes=64 << 10)
    # the three-phase exchange of train.Trainer: the head end of the arena first (while the backbone's backward would
    # still run), then the deep backbone blocks (while the shallow ones' backward runs), then the rest; every element
    # exactly once
    split, deep = 123456, 40000
    red.start(split, None)
    red.start(deep, split)
    red.start(0, deep)
    red.finish()
    out[rank] = (mine.numpy(), flat.numpy().copy(), red.grad_scale, len(red.bounds))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g0, s0, scale, nb = out[0]
    g1, s1, _, _ = out[1]
    assert nb > 1 and scale == 0.5
    np.testing.assert_array_equal(s0, s1)                           # every rank holds the same sum
    np.testing.assert_allclose(s0, g0 + g1, rtol=1e-6, atol=1e-6)   # = sum of the per-rank gradients


def test_the_three_exchange_ranges_tile_the_arena_and_follow_the_backward_order():
    """Data-parallel exchange: [head end | deep backbone blocks 7..13 | blocks 1..6 + stem] are contiguous
    ranges of the flat arena in the order the three backward phases complete them; the deep range holds most of the backbone."""
    from multiposenet_amd import net as mnet
    shapes = {k: v for k, v in mnet.variable_shapes(1.0).items() if mnet.is_trainable(k)}
    arena = mnet._Arena(shapes, "cpu")
    end, deep = mnet.backbone_grad_end_of(arena), mnet.backbone_deep_begin_of(arena)
    total = arena.new().numel()
    assert 0 < deep < end < total and deep % 4 == 0 and end % 4 == 0
    for k, (o, n, _) in arena.offsets.items():
        if not k.startswith("MobilenetV1/"):
            assert o >= end, k
        else:
            blk = int(k.split("Conv2d_")[1].split("_")[0].split("/")[0])
            assert (o >= deep) == (blk >= mnet.DP_DEEP_FROM_BLOCK), k
    assert (end - deep) > 8 * deep                      # 2.9 M of 3.2 M backbone parameters travel under the shallow blocks
    assert deep * 4 < 1.5e6                             # what stays exposed: ~1.3 MB
