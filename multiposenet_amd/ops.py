"""Thin host wrappers over the C ABI (include/mpn.h): shape checks, buffer allocation, launch.

Tensors are torch CUDA tensors used purely as device-memory owners; activations are NHWC
(`[N, H, W, C]`, contiguous) in float32 or bfloat16. Nothing here computes on the host and
nothing falls back to PyTorch ops: every function ends in a `_lib.call` into libmpn_hip.so.
"""
import ctypes
from collections import namedtuple

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_RELU6, call, ptr, stream_ptr

BN_MOMENTUM = 0.95   # detector/utils/layer_utils.py:5, detector/backbones/mobilenet_v1.py:7
BN_EPSILON = 1e-3    # layer_utils.py:6, mobilenet_v1.py:8

# Test switch (tests/test_fullsize_gpu.py): True launches every job of every grouped / batched entry point ALONE - the same entry point
# with one job per call - so that a step with shared grids can be compared with one where no block ever walks from one job into the
# next. Read when a launch is made (grouped calls) or a device table is built (the batched finalizes / reductions): set it before the
# net's first step of a shape. Not a product path: nothing in the package sets it.
LAUNCH_JOBS_ALONE = False

# Producer-side batch-norm record that consumers apply on load.
Affine = namedtuple("Affine", ["scale", "shift", "act"])


def _f32(n, dev):
    return torch.empty(n, dtype=torch.float32, device=dev)


def _check_nhwc(x, name="x"):
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError(f"{name} must be a contiguous NHWC tensor, got shape {tuple(x.shape)}")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError(f"{name}: unsupported dtype {x.dtype}")


def _aff(a):
    if a is None:
        return None, None, ACT_NONE
    return ptr(a.scale), ptr(a.shift), int(a.act)


# ----------------------------------------------------------------------------- dense conv (MFMA)
class PackedConv:
    """Weights of one dense conv packed for the MFMA kernel (forward and, lazily, data-gradient)."""

    def __init__(self, w_hwio, dtype):
        k, k2, cin, cout = w_hwio.shape
        assert k == k2 and k in (1, 3)
        if w_hwio.dtype != torch.float32 or not w_hwio.is_contiguous():
            raise ValueError("conv weights must be a contiguous float32 HWIO tensor")
        self.ksize, self.cin, self.cout, self.dtype = k, cin, cout, dtype
        self.w = w_hwio  # f32 master view [k,k,Cin,Cout] (HWIO, the reference's variable layout)
        dc = _lib.dtype_code(dtype)
        self.fwd = torch.empty(_lib.lib().mpn_conv_packed_bytes(cin, cout, k, 0, dc), dtype=torch.uint8, device=w_hwio.device)
        self.bwd = torch.empty(_lib.lib().mpn_conv_packed_bytes(cin, cout, k, 1, dc), dtype=torch.uint8, device=w_hwio.device)
        self.repack()

    def repack(self, with_bwd=True):
        dc = _lib.dtype_code(self.dtype)
        call("mpn_conv_pack_weights", ptr(self.w), self.cin, self.cout, self.ksize, 0, dc, ptr(self.fwd), stream_ptr())
        if with_bwd:
            call("mpn_conv_pack_weights", ptr(self.w), self.cin, self.cout, self.ksize, 1, dc, ptr(self.bwd), stream_ptr())


def conv_num_parts(N, H, W, ksize):
    """Rows to SIZE a convolution's statistics slab with (one per tile: an upper bound of what any kernel writes)."""
    return _lib.lib().mpn_conv_num_parts(N, H, W, ksize)


def conv_stats_rows(N, H, W, cin, cout, ksize, dtype):
    """Rows of its statistics slab a convolution [N,H,W,cin] -> cout WRITES (conv_fwd[_grouped]; conv_bwd_data_bn[_grouped] with
    cin = the gradient's channels, cout = the fed batch-norm's): the nparts of the finalize behind it. The persistent 3x3 kernel
    writes one row per block, every other kernel conv_num_parts rows (mpn_conv_stats_rows)."""
    rows = _lib.lib().mpn_conv_stats_rows(int(N), int(H), int(W), int(cin), int(cout), int(ksize), _lib.dtype_code(dtype))
    if rows <= 0:
        raise _lib.MpnError(f"mpn_conv_stats_rows({N}, {H}, {W}, {cin}, {cout}, {ksize}) = {rows}: {_lib.last_error()}")
    return rows


def _slice_stride(t, channels):
    """Pixel stride (elements) of an NHWC tensor or of a channel slice `t[..., a:a+channels]` of a wider contiguous one."""
    if t.dim() != 4 or t.shape[3] != channels or t.stride(3) != 1:
        raise ValueError(f"expected an NHWC tensor (or channel slice) with {channels} channels, got {tuple(t.shape)}")
    ps = t.stride(2)
    if t.stride(1) != t.shape[2] * ps or t.stride(0) != t.shape[1] * t.shape[2] * ps or ps < channels:
        raise ValueError("only channel slices of contiguous NHWC tensors are supported")
    return ps


def _row_stride(t, channels):
    """Row stride (elements) of a [..., C] tensor viewed as rows of C channels: dense, or a channel slice of a wider
    contiguous tensor (every leading dimension must collapse onto one uniform row stride)."""
    if t.shape[-1] != channels or t.stride(-1) != 1 or t.dim() < 2:
        raise ValueError(f"expected [..., {channels}] with contiguous channels, got {tuple(t.shape)}")
    rs = t.stride(-2)
    for d in range(t.dim() - 2, 0, -1):
        if t.shape[d - 1] != 1 and t.stride(d - 1) != t.shape[d] * t.stride(d):
            raise ValueError("only channel slices of contiguous tensors are supported")
    if rs < channels:
        raise ValueError("overlapping rows")
    return rs


def conv_fwd(x, packed, cout, ksize, affine=None, out=None, stats_part=None, up_res=None):
    """y = conv(act(bn(x))) [+ nearest2x(up_res)]; optional per-tile BN partial sums. x and out may be channel slices
    (`t[..., a:b]`) of wider contiguous NHWC tensors: the kernel takes their pixel strides."""
    if x.dim() != 4 or x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError(f"x must be an NHWC f32/bf16/f16 tensor, got {tuple(x.shape)} {x.dtype}")
    N, H, W, cin = x.shape
    xs = _slice_stride(x, cin)
    if out is None:
        out = torch.empty((N, H, W, cout), dtype=x.dtype, device=x.device)
    ys = _slice_stride(out, cout)
    sc, sh, act = _aff(affine)
    call("mpn_conv_fwd", ptr(x), ptr(packed), ptr(out), N, H, W, cin, cout, xs, ys, ksize, _lib.dtype_code(x.dtype),
         sc, sh, act, ptr(stats_part), ptr(up_res), stream_ptr())
    return out


def conv_fwd_grouped(xs, packeds, cout, ksize, affines, outs, stats_parts):
    """Several independent convolutions of one channel geometry in one grid (mpn_conv_fwd_grouped): lists per job;
    affines / stats_parts entries may be None. All inputs [N,H_j,W_j,Cin] of one dtype, same activation code."""
    import ctypes
    n = len(xs)
    if LAUNCH_JOBS_ALONE and n > 1:
        for j in range(n):
            conv_fwd_grouped([xs[j]], [packeds[j]], cout, ksize, [affines[j]], [outs[j]], [stats_parts[j]])
        return outs
    N, _, _, cin = xs[0].shape
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    sc, sh, act = [], [], ACT_NONE
    for a in affines:
        s_, h_, a_ = _aff(a)
        sc.append(s_); sh.append(h_)
        if a is not None:
            act = a_
    call("mpn_conv_fwd_grouped", n, PA(*[ptr(x) for x in xs]), PA(*[ptr(p) for p in packeds]), PA(*[ptr(o) for o in outs]), N,
         IA(*[x.shape[1] for x in xs]), IA(*[x.shape[2] for x in xs]), cin, cout, IA(*[_slice_stride(x, cin) for x in xs]),
         IA(*[_slice_stride(o, cout) for o in outs]), ksize, _lib.dtype_code(xs[0].dtype), PA(*sc), PA(*sh), int(act), PA(*[ptr(t) for t in stats_parts]), stream_ptr())
    return outs


def conv_bwd_data_bn_supported(k, c, ksize, dtype):
    return bool(_lib.lib().mpn_conv_bwd_data_bn_supported(int(k), int(c), int(ksize), _lib.dtype_code(dtype)))


def conv_bwd_data_bn(dy, packed_t, c, ksize, bn, x_bn, out, part):
    """Data gradient of one convolution (3x3, or a deep 1x1 layer) that also reduces for the batch-norm layer `bn` it feeds
    (raw tensor x_bn): out <- masked gradient, part <- partial sums of g and g * x (raw x: bn_backward(..., reduced_parts=rows,
    raw=True)). Returns the rows the slab holds."""
    N, H, W, k = dy.shape
    call("mpn_conv_bwd_data_bn", ptr(dy), ptr(packed_t), ptr(out), N, H, W, k, int(c), _slice_stride(dy, k), _slice_stride(out, c), int(ksize),
         _lib.dtype_code(dy.dtype), ptr(x_bn), _slice_stride(x_bn, c), ptr(bn.scale), ptr(bn.shift), int(bn.act), ptr(part), stream_ptr())
    return conv_stats_rows(N, H, W, k, c, ksize, dy.dtype)


def conv_bwd_data_bn_grouped(dys, packeds_t, c, bns, xs_bn, outs, parts):
    """Data gradients of several independent 3x3 convolutions in one grid that also reduce for the batch-norm layers `bns`
    they feed (raw tensors xs_bn): outs[j] <- masked gradient, parts[j] <- partial sums of g and g * x (RAW x: finalize with a
    BnBwdFinalizeBatch job marked raw). Returns the rows each slab holds (conv_stats_rows)."""
    import ctypes
    n = len(dys)
    if LAUNCH_JOBS_ALONE and n > 1:
        return [conv_bwd_data_bn_grouped([dys[j]], [packeds_t[j]], c, [bns[j]], [xs_bn[j]], [outs[j]], [parts[j]])[0] for j in range(n)]
    N, _, _, k = dys[0].shape
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    call("mpn_conv_bwd_data_bn_grouped", n, PA(*[ptr(t) for t in dys]), PA(*[ptr(p) for p in packeds_t]), PA(*[ptr(o) for o in outs]), N,
         IA(*[t.shape[1] for t in dys]), IA(*[t.shape[2] for t in dys]), k, int(c), IA(*[_slice_stride(t, k) for t in dys]),
         IA(*[_slice_stride(o, c) for o in outs]), _lib.dtype_code(dys[0].dtype), PA(*[ptr(x) for x in xs_bn]),
         IA(*[_slice_stride(x, c) for x in xs_bn]), PA(*[ptr(b.scale) for b in bns]), PA(*[ptr(b.shift) for b in bns]), int(bns[0].act),
         PA(*[ptr(t) for t in parts]), stream_ptr())
    return [conv_stats_rows(N, t.shape[1], t.shape[2], k, c, 3, t.dtype) for t in dys]


def conv1x1_bwd_fused_supported(cin, cout, dtype):
    """True when conv1x1_bwd_fused takes this pointwise layer (bf16 storage, Cin <= 128, Cout <= 128)."""
    return _lib.lib().mpn_conv1x1_bwd_fused_supported(int(cin), int(cout), _lib.dtype_code(dtype)) == 1


def conv1x1_bwd_fused_apply_supported(cin, cout, dtype):
    """True when conv1x1_bwd_fused(apply_bn=...) takes this layer (Cin <= 64, Cout <= 128, bf16)."""
    return _lib.lib().mpn_conv1x1_bwd_fused_apply_supported(int(cin), int(cout), _lib.dtype_code(dtype)) == 1


def conv1x1_bwd_fused(x, dy, w, bn, out, wpart, bn_part, apply_bn=None, y_raw=None):
    """A thin 1x1 convolution's backward in one pass over x and dy: x = the layer's RAW input (raw output of the layer with batch-norm
    state `bn`), dy = gradient w.r.t. the layer's output, w = its f32 kernel [1,1,Cin,Cout]. out <- the data gradient masked by bn's
    activation, wpart <- the weight gradient's split-K slab (conv_wgrad_num_parts rows; reduce later), bn_part <- partial sums of g
    and g * x (raw x). Returns the rows both slabs hold - pass them to bn_backward(..., reduced_parts=rows, raw=True).
    bn_part=None: no reduction, out is the plain (unmasked) data gradient; `bn` then only supplies the affine of x (an ops.Affine does)."""
    N, H, W, cin = x.shape
    cout = dy.shape[3]
    rows = conv_wgrad_num_parts(N, H, W, cin, cout, 1, x.dtype)
    if wpart.numel() < rows * cin * cout or (bn_part is not None and bn_part.numel() < rows * 2 * cin):
        raise ValueError("conv1x1_bwd_fused: partial slab too small")
    if apply_bn is not None:
        # dy is the gradient w.r.t. the ACTIVATED output of the layer's own batch-norm `apply_bn` (finalized: k1 / k2 set), y_raw the
        # layer's raw output: that batch-norm's apply pass happens while dY is staged
        a = apply_bn
        call("mpn_conv1x1_bwd_fused_apply", ptr(x), ptr(dy), ptr(y_raw), ptr(w), ptr(out), ptr(wpart), ptr(bn_part), N, H, W, cin, cout,
             _slice_stride(x, cin), _slice_stride(dy, cout), _slice_stride(y_raw, cout), _slice_stride(out, cin), _lib.dtype_code(x.dtype),
             ptr(bn.scale), ptr(bn.shift), int(bn.act), ptr(a.scale), ptr(a.shift), ptr(a.mean), ptr(a.invstd), ptr(a.k1), ptr(a.k2),
             int(a.act), stream_ptr())
        return rows
    call("mpn_conv1x1_bwd_fused", ptr(x), ptr(dy), ptr(w), ptr(out), ptr(wpart), ptr(bn_part), N, H, W, cin, cout, _slice_stride(x, cin),
         _slice_stride(dy, cout), _slice_stride(out, cin), _lib.dtype_code(x.dtype), ptr(bn.scale), ptr(bn.shift), int(bn.act), stream_ptr())
    return rows


def conv_wgrad_num_parts(N, H, W, cin, cout, ksize, dtype):
    return _lib.lib().mpn_conv_wgrad_num_parts(N, H, W, cin, cout, ksize, _lib.dtype_code(dtype))


def conv_bwd_weight(x, dy, ksize, affine, dw_out, part=None, reduce=True):
    """dw_out (f32 HWIO view, [k,k,Cin,Cout]) <- sum over pixels of act(bn(x)) (x) dy.
    reduce=False leaves the split-K slab in `part` for a later batched reduction (SlabReducer)."""
    N, H, W, cin = x.shape
    cout = dy.shape[3]
    nparts = conv_wgrad_num_parts(N, H, W, cin, cout, ksize, x.dtype)
    n = ksize * ksize * cin * cout
    if part is None:
        part = _f32(nparts * n, x.device)
    sc, sh, act = _aff(affine)
    call("mpn_conv_bwd_weight", ptr(x), ptr(dy), ptr(part), N, H, W, cin, cout, _slice_stride(x, cin), _slice_stride(dy, cout),
         ksize, _lib.dtype_code(x.dtype), sc, sh, act, stream_ptr())
    if reduce:
        call("mpn_reduce_partials", ptr(part), nparts, n, ptr(dw_out), 0, 1.0, stream_ptr())
    return dw_out


class SlabReducer:
    """All weight-gradient slab reductions of a step in one launch (mpn_reduce_partials_batched).
    `jobs`: list of (part tensor, nparts, n, out tensor); the device table is built once."""

    def __init__(self, jobs, device):
        import ctypes
        self._alone = [SlabReducer([j], device) for j in jobs] if LAUNCH_JOBS_ALONE and len(jobs) > 1 else None
        lib = _lib.lib()
        nb = lib.mpn_reduce_desc_bytes()
        host = (ctypes.c_ubyte * (nb * len(jobs)))()
        begin = 0
        for j, (part, nparts, n, out) in enumerate(jobs):
            blocks = lib.mpn_reduce_desc_fill(ctypes.byref(host, j * nb), ptr(part), int(nparts), int(n), ptr(out), 1.0, begin)
            if blocks <= 0:
                raise ValueError("bad slab reduction job")
            begin += blocks
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)
        self.njobs, self.blocks = len(jobs), begin
        self._keep = jobs   # the tensors the table points at

    def run(self):
        if self._alone is not None:
            for r in self._alone:
                r.run()
            return
        call("mpn_reduce_partials_batched", ptr(self.table), self.njobs, self.blocks, stream_ptr())


def conv_wgrad_grouped_num_parts(N, hws, cin, cout, ksize, dtype):
    """Slab counts of conv_bwd_weight_grouped for jobs of sizes hws = [(H, W), ...]."""
    import ctypes
    n = len(hws)
    if LAUNCH_JOBS_ALONE and n > 1:
        return [conv_wgrad_grouped_num_parts(N, [hw], cin, cout, ksize, dtype)[0] for hw in hws]
    IA = ctypes.c_int * n
    out = IA()
    call("mpn_conv_wgrad_grouped_num_parts", n, N, IA(*[h for h, _ in hws]), IA(*[w for _, w in hws]), cin, cout, ksize,
         _lib.dtype_code(dtype), out)
    return list(out)


def conv_bwd_weight_grouped(xs, dys, ksize, affines, parts):
    """conv_bwd_weight of several independent layers of one channel geometry in one grid (the pyramid levels of a subnet
    stage); the slabs stay in `parts` (sized by conv_wgrad_grouped_num_parts) for the batched reduction."""
    import ctypes
    n = len(xs)
    if LAUNCH_JOBS_ALONE and n > 1:
        for j in range(n):
            conv_bwd_weight_grouped([xs[j]], [dys[j]], ksize, [affines[j]], [parts[j]])
        return
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    N, cin, cout = xs[0].shape[0], xs[0].shape[3], dys[0].shape[3]
    sc, sh, act = zip(*[_aff(a) for a in affines])
    if len(set(act)) != 1:
        raise ValueError("grouped jobs must share the activation")
    call("mpn_conv_bwd_weight_grouped", n, PA(*[ptr(t) for t in xs]), PA(*[ptr(t) for t in dys]), PA(*[ptr(t) for t in parts]), N,
         IA(*[t.shape[1] for t in xs]), IA(*[t.shape[2] for t in xs]), cin, cout, IA(*[_slice_stride(t, cin) for t in xs]),
         IA(*[_slice_stride(t, cout) for t in dys]), ksize, _lib.dtype_code(xs[0].dtype), PA(*sc), PA(*sh), int(act[0]), stream_ptr())


# ----------------------------------------------------------------------------- batch norm
class BNState:
    """Device state of one batch-norm layer: views into the parameter arenas + per-step buffers."""

    def __init__(self, gamma, beta, moving_mean, moving_var, act):
        C = gamma.numel()
        dev = gamma.device
        self.C, self.act = C, act
        self.gamma, self.beta, self.moving_mean, self.moving_var = gamma, beta, moving_mean, moving_var
        self.scale, self.shift = _f32(C, dev), _f32(C, dev)
        self.mean, self.invstd = _f32(C, dev), _f32(C, dev)
        self.k1, self.k2 = _f32(C, dev), _f32(C, dev)
        self.dgamma = self.dbeta = None  # views into the gradient arena, set by the owner

    @property
    def affine(self):
        return Affine(self.scale, self.shift, self.act)

    def channel_slice(self, lo, hi):
        """The layer of channels lo..hi-1 of a MERGED layer (two batch-norms whose variables lie side by side in the arenas and
        whose inputs are channel slices of one tensor): every tensor of the result is a view of this state's."""
        s = object.__new__(BNState)
        s.C, s.act = hi - lo, self.act
        for k in ("gamma", "beta", "moving_mean", "moving_var", "scale", "shift", "mean", "invstd", "k1", "k2"):
            setattr(s, k, getattr(self, k)[lo:hi])
        s.dgamma = self.dgamma[lo:hi] if self.dgamma is not None else None
        s.dbeta = self.dbeta[lo:hi] if self.dbeta is not None else None
        return s


def bn_stats(x, part=None):
    M, C = x.numel() // x.shape[-1], x.shape[-1]
    nparts = _lib.lib().mpn_bn_stats_num_parts(M)
    if part is None:
        part = _f32(nparts * 2 * C, x.device)
    call("mpn_bn_stats", ptr(x), M, C, _lib.dtype_code(x.dtype), ptr(part), stream_ptr())
    return part, nparts


def bn_finalize(bn, part, nparts, count, training=True):
    call("mpn_bn_finalize", ptr(part), nparts, bn.C, count, ptr(bn.gamma), ptr(bn.beta),
         ptr(bn.moving_mean) if training else None, ptr(bn.moving_var) if training else None,
         BN_MOMENTUM, BN_EPSILON, ptr(bn.scale), ptr(bn.shift), ptr(bn.mean), ptr(bn.invstd), stream_ptr())


class BnFinalizeBatch:
    """Forward finalizes of several independent layers in one launch (mpn_bn_finalize_batched).
    jobs: list of (bn, part tensor, nparts, count); the device table is built once (pointers are static)."""

    def __init__(self, jobs, device):
        import ctypes
        self._alone = [BnFinalizeBatch([j], device) for j in jobs] if LAUNCH_JOBS_ALONE and len(jobs) > 1 else None
        lib = _lib.lib()
        nb = lib.mpn_bn_fin_desc_bytes()
        host = (ctypes.c_ubyte * (nb * len(jobs)))()
        begin = 0
        for j, (bn, part, nparts, count) in enumerate(jobs):
            blocks = lib.mpn_bn_fin_desc_fill(ctypes.byref(host, j * nb), ptr(part), int(nparts), bn.C, int(count), ptr(bn.gamma),
                                              ptr(bn.beta), ptr(bn.moving_mean), ptr(bn.moving_var), ptr(bn.scale), ptr(bn.shift),
                                              ptr(bn.mean), ptr(bn.invstd), begin)
            if blocks <= 0:
                raise ValueError("bad batch-norm finalize job")
            begin += blocks
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)
        self.njobs, self.blocks, self._keep = len(jobs), begin, jobs

    def run(self):
        if self._alone is not None:
            for r in self._alone:
                r.run()
            return
        call("mpn_bn_finalize_batched", ptr(self.table), self.njobs, self.blocks, BN_MOMENTUM, BN_EPSILON, stream_ptr())


class BnBwdFinalizeBatch:
    """Backward finalizes (dgamma, dbeta, k1, k2) of several independent layers in one launch."""

    def __init__(self, jobs, device):
        import ctypes
        self._alone = [BnBwdFinalizeBatch([j], device) for j in jobs] if LAUNCH_JOBS_ALONE and len(jobs) > 1 else None
        lib = _lib.lib()
        nb = lib.mpn_bn_bwd_fin_desc_bytes()
        host = (ctypes.c_ubyte * (nb * len(jobs)))()
        begin = 0
        for j, job in enumerate(jobs):
            bn, part, nparts, count = job[:4]
            if len(job) > 4 and job[4]:      # raw: the slab holds sum g * x (written by conv_bwd_data_bn_grouped)
                blocks = lib.mpn_bn_bwd_fin_desc_fill_raw(ctypes.byref(host, j * nb), ptr(part), int(nparts), bn.C, int(count),
                                                          ptr(bn.dgamma), ptr(bn.dbeta), ptr(bn.k1), ptr(bn.k2), ptr(bn.mean),
                                                          ptr(bn.invstd), begin)
            else:
                blocks = lib.mpn_bn_bwd_fin_desc_fill(ctypes.byref(host, j * nb), ptr(part), int(nparts), bn.C, int(count),
                                                      ptr(bn.dgamma), ptr(bn.dbeta), ptr(bn.k1), ptr(bn.k2), begin)
            if blocks <= 0:
                raise ValueError("bad batch-norm backward finalize job")
            begin += blocks
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)
        self.njobs, self.blocks, self._keep = len(jobs), begin, jobs

    def run(self):
        if self._alone is not None:
            for r in self._alone:
                r.run()
            return
        call("mpn_bn_bwd_finalize_batched", ptr(self.table), self.njobs, self.blocks, stream_ptr())


def bn_bwd_reduce(bn, dA, x, part):
    """First pass of bn_backward alone: partial sums of g and g * xhat into `part` (mpn_bn_stats_num_parts(M) rows)."""
    M, C = x.numel() // x.shape[-1], x.shape[-1]
    call("mpn_bn_bwd_reduce", ptr(dA), ptr(x), M, C, _lib.dtype_code(x.dtype), ptr(bn.scale), ptr(bn.shift), ptr(bn.mean),
         ptr(bn.invstd), int(bn.act), ptr(part), stream_ptr())
    return _lib.lib().mpn_bn_stats_num_parts(M)


def bn_bwd_apply(bn, dA, x, add_ch0=None):
    """Last pass of bn_backward alone (after a finalize filled bn.k1 / bn.k2)."""
    M, C = x.numel() // x.shape[-1], x.shape[-1]
    call("mpn_bn_bwd_apply", ptr(dA), ptr(x), M, C, _lib.dtype_code(x.dtype), ptr(bn.scale), ptr(bn.shift), ptr(bn.mean),
         ptr(bn.invstd), ptr(bn.k1), ptr(bn.k2), int(bn.act), ptr(add_ch0), stream_ptr())
    return dA


def _bn_group_args(bns, dAs, xs):
    import ctypes
    n = len(bns)
    PA, LA, IA = ctypes.c_void_p * n, ctypes.c_longlong * n, ctypes.c_int * n
    C = xs[0].shape[-1]
    Ms = [x.numel() // x.shape[-1] for x in xs]
    strides = (IA(*[_row_stride(t, C) for t in dAs]), IA(*[_row_stride(t, C) for t in xs]))   # channel slices allowed
    return n, PA, PA(*[ptr(t) for t in dAs]), PA(*[ptr(t) for t in xs]), LA(*Ms), C, _lib.dtype_code(xs[0].dtype), strides


def bn_bwd_reduce_grouped(bns, dAs, xs, parts):
    """bn_bwd_reduce of several independent layers (same channel count, dtype, activation) in one grid."""
    if LAUNCH_JOBS_ALONE and len(bns) > 1:
        for j in range(len(bns)):
            bn_bwd_reduce_grouped([bns[j]], [dAs[j]], [xs[j]], [parts[j]])
        return
    n, PA, pd, px, Ms, C, dc, (sd, sx) = _bn_group_args(bns, dAs, xs)
    call("mpn_bn_bwd_reduce_grouped", n, pd, px, Ms, C, dc, PA(*[ptr(b.scale) for b in bns]), PA(*[ptr(b.shift) for b in bns]),
         PA(*[ptr(b.mean) for b in bns]), PA(*[ptr(b.invstd) for b in bns]), int(bns[0].act), PA(*[ptr(p) for p in parts]),
         sd, sx, stream_ptr())


def bn_bwd_apply_grouped(bns, dAs, xs, add_ch0s=None):
    """bn_bwd_apply of several independent layers in one grid (after their finalizes)."""
    if LAUNCH_JOBS_ALONE and len(bns) > 1:
        for j in range(len(bns)):
            bn_bwd_apply_grouped([bns[j]], [dAs[j]], [xs[j]], None if add_ch0s is None else [add_ch0s[j]])
        return
    n, PA, pd, px, Ms, C, dc, (sd, sx) = _bn_group_args(bns, dAs, xs)
    add = add_ch0s if add_ch0s is not None else [None] * n
    call("mpn_bn_bwd_apply_grouped", n, pd, px, Ms, C, dc, PA(*[ptr(b.scale) for b in bns]), PA(*[ptr(b.shift) for b in bns]),
         PA(*[ptr(b.mean) for b in bns]), PA(*[ptr(b.invstd) for b in bns]), PA(*[ptr(b.k1) for b in bns]),
         PA(*[ptr(b.k2) for b in bns]), int(bns[0].act), PA(*[ptr(t) for t in add]), sd, sx, stream_ptr())


def bn_inference_affine(bn):
    call("mpn_bn_inference_affine", bn.C, ptr(bn.gamma), ptr(bn.beta), ptr(bn.moving_mean), ptr(bn.moving_var),
         BN_EPSILON, ptr(bn.scale), ptr(bn.shift), stream_ptr())


def bn_act_apply(x, affine, out=None):
    M, C = x.numel() // x.shape[-1], x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    call("mpn_bn_act_apply", ptr(x), ptr(out), M, C, _lib.dtype_code(x.dtype), ptr(affine.scale), ptr(affine.shift),
         int(affine.act), stream_ptr())
    return out


def bn_backward(bn, dA, x, part, add_ch0=None, reduced_parts=0, raw=False, apply=True):
    """In place: dA (gradient w.r.t. act(bn(x))) -> gradient w.r.t. the raw conv output x.
    Writes bn.dgamma / bn.dbeta. `part` must hold mpn_bn_stats_num_parts(M)*2*C floats.
    reduced_parts > 0: the producer of dA already wrote that many partial rows into `part` (dwconv_bwd_data(..., bn=...);
    raw: conv_bwd_data_bn, whose slab holds sum g * x with the raw x)."""
    M, C = x.numel() // x.shape[-1], x.shape[-1]
    dc = _lib.dtype_code(x.dtype)
    nparts = _lib.lib().mpn_bn_stats_num_parts(M)
    if reduced_parts and raw:
        call("mpn_bn_bwd_finalize_raw", ptr(part), int(reduced_parts), C, M, ptr(bn.dgamma), ptr(bn.dbeta), ptr(bn.k1), ptr(bn.k2),
             ptr(bn.mean), ptr(bn.invstd), stream_ptr())
    elif reduced_parts:
        call("mpn_bn_bwd_finalize", ptr(part), int(reduced_parts), C, M, ptr(bn.dgamma), ptr(bn.dbeta), ptr(bn.k1), ptr(bn.k2), stream_ptr())
    else:
        call("mpn_bn_bwd_reduce", ptr(dA), ptr(x), M, C, dc, ptr(bn.scale), ptr(bn.shift), ptr(bn.mean), ptr(bn.invstd),
             int(bn.act), ptr(part), stream_ptr())
        call("mpn_bn_bwd_finalize", ptr(part), nparts, C, M, ptr(bn.dgamma), ptr(bn.dbeta), ptr(bn.k1), ptr(bn.k2), stream_ptr())
    if apply:       # (apply=False: reduce + finalize only - the consumer applies on load, conv1x1_bwd_fused(apply_bn=...))
        call("mpn_bn_bwd_apply", ptr(dA), ptr(x), M, C, dc, ptr(bn.scale), ptr(bn.shift), ptr(bn.mean), ptr(bn.invstd),
             ptr(bn.k1), ptr(bn.k2), int(bn.act), ptr(add_ch0), stream_ptr())
    return dA


# ----------------------------------------------------------------------------- depthwise
def dwconv_out_hw(H, W, stride):
    l = _lib.lib()
    return l.mpn_dwconv_out_size(H, stride), l.mpn_dwconv_out_size(W, stride)


def dwconv_num_parts(N, H, W, C, stride, dtype):
    return _lib.lib().mpn_dwconv_num_parts(N, H, W, C, stride, _lib.dtype_code(dtype))


def dwconv_fwd(x, w, stride, affine=None, out=None, stats_part=None):
    _check_nhwc(x)
    N, H, W, C = x.shape
    OH, OW = dwconv_out_hw(H, W, stride)
    if out is None:
        out = torch.empty((N, OH, OW, C), dtype=x.dtype, device=x.device)
    sc, sh, act = _aff(affine)
    call("mpn_dwconv_fwd", ptr(x), ptr(w), ptr(out), N, H, W, C, stride, _lib.dtype_code(x.dtype), sc, sh, act, 0,
         ptr(stats_part), stream_ptr())
    return out


def dwconv_bwd_data_bn_num_parts(N, H, W, C, stride, dtype):
    """Partial rows of the fused data gradient + batch-norm reduction; 0 = not available for this shape."""
    return _lib.lib().mpn_dwconv_bwd_data_bn_num_parts(N, H, W, C, stride, _lib.dtype_code(dtype))


def dwconv_bwd_data_add_supported(N, H, W, C, stride, dtype):
    """True when dwconv_bwd_data(addend=...) takes this shape (stride 2, even H and W)."""
    return _lib.lib().mpn_dwconv_bwd_data_add_supported(N, H, W, C, stride, _lib.dtype_code(dtype)) == 1


def dwconv_bwd_data(dy, w, in_hw, stride, out=None, bn=None, x_bn=None, part=None, addend=None):
    """bn / x_bn / part: fuse the batch-norm backward reduction of the layer that the result feeds (x_bn = that layer's raw
    conv output, same shape as the result); returns (out, rows) then - pass rows to bn_backward(reduced_parts=rows).
    addend: a tensor of the result's shape added to the gradient before the store (and before the reduction)."""
    N, OH, OW, C = dy.shape
    H, W = in_hw
    if out is None:
        out = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    dc = _lib.dtype_code(dy.dtype)
    if addend is not None:
        if addend.shape != out.shape or addend.dtype != out.dtype or not addend.is_contiguous():
            raise ValueError("dwconv_bwd_data: addend must have the result's shape and dtype")
        rows = 0
        if bn is not None:
            rows = dwconv_bwd_data_bn_num_parts(N, H, W, C, stride, dy.dtype)
            if rows <= 0:
                raise ValueError("dwconv_bwd_data: fused batch-norm reduction not available for this shape")
            if part is None:
                part = _f32(rows * 2 * C, dy.device)
            if part.numel() < rows * 2 * C:
                raise ValueError("dwconv_bwd_data: partial slab too small")
        call("mpn_dwconv_bwd_data_add", ptr(dy), ptr(w), ptr(out), N, H, W, C, stride, dc, ptr(addend),
             ptr(x_bn) if bn is not None else None, *((ptr(bn.scale), ptr(bn.shift), ptr(bn.mean), ptr(bn.invstd), int(bn.act),
                                                      ptr(part)) if bn is not None else (None, None, None, None, 0, None)),
             stream_ptr())
        return (out, rows) if bn is not None else out
    if bn is not None:
        rows = dwconv_bwd_data_bn_num_parts(N, H, W, C, stride, dy.dtype)
        if rows <= 0:
            raise ValueError("dwconv_bwd_data: fused batch-norm reduction not available for this shape")
        if part is None:
            part = _f32(rows * 2 * C, dy.device)
        if part.numel() < rows * 2 * C:
            raise ValueError("dwconv_bwd_data: partial slab too small")
        call("mpn_dwconv_bwd_data_bn", ptr(dy), ptr(w), ptr(out), N, H, W, C, stride, dc, ptr(x_bn), ptr(bn.scale), ptr(bn.shift),
             ptr(bn.mean), ptr(bn.invstd), int(bn.act), ptr(part), stream_ptr())
        return out, rows
    call("mpn_dwconv_bwd_data", ptr(dy), ptr(w), ptr(out), N, H, W, C, stride, dc, stream_ptr())
    return out


def dwconv_wgrad_num_parts(N, H, W, C, stride, dtype):
    return _lib.lib().mpn_dwconv_wgrad_num_parts(N, H, W, C, stride, _lib.dtype_code(dtype))


def dwconv_bwd_weight(x, dy, stride, affine, dw_out, part=None, reduce=True):
    N, H, W, C = x.shape
    dc = _lib.dtype_code(x.dtype)
    nparts = _lib.lib().mpn_dwconv_wgrad_num_parts(N, H, W, C, stride, dc)
    if part is None:
        part = _f32(nparts * 9 * C, x.device)
    sc, sh, act = _aff(affine)
    call("mpn_dwconv_bwd_weight", ptr(x), ptr(dy), ptr(part), N, H, W, C, stride, dc, sc, sh, act, stream_ptr())
    if reduce:
        call("mpn_reduce_partials", ptr(part), nparts, 9 * C, ptr(dw_out), 0, 1.0, stream_ptr())
    return dw_out


def dwconv_bwd_fused_supported(N, H, W, C, stride, dtype):
    """True when dwconv_bwd_fused takes this layer (stride 1; stride 2 with even H and W)."""
    return _lib.lib().mpn_dwconv_bwd_fused_supported(N, H, W, C, stride, _lib.dtype_code(dtype)) == 1


def dwconv_bwd_fused(x, dy, w, bn, dw_out, out=None, wpart=None, bn_part=None, reduce=True, reduce_bn=True, stride=1, addend=None):
    """Stride-1 depthwise backward in one pass: x = the conv's RAW input (the raw output of the layer with batch-norm state `bn`,
    whose affine + activation the forward applied on load), dy = gradient w.r.t. the conv's output. Writes the data gradient to
    `out`, the weight-gradient partials to `wpart` (reduced into dw_out when reduce) and, when reduce_bn, the partial sums of
    that batch-norm's backward reduction to `bn_part`; returns (out, rows) - pass rows to bn_backward(reduced_parts=rows)."""
    N, H, W, C = x.shape
    dc = _lib.dtype_code(x.dtype)
    if not dwconv_bwd_fused_supported(N, H, W, C, stride, x.dtype):
        raise ValueError("dwconv_bwd_fused: shape not supported")
    if addend is not None and (stride != 2 or addend.shape != x.shape or addend.dtype != x.dtype or not addend.is_contiguous()):
        raise ValueError("dwconv_bwd_fused: addend needs stride 2 and the result's shape and dtype")
    if out is None:
        out = torch.empty_like(x)
    rows = _lib.lib().mpn_dwconv_wgrad_num_parts(N, H, W, C, stride, dc)
    if wpart is None:
        wpart = _f32(rows * 9 * C, x.device)
    if reduce_bn and bn_part is None:
        bn_part = _f32(rows * 2 * C, x.device)
    if wpart.numel() < rows * 9 * C or (reduce_bn and bn_part.numel() < rows * 2 * C):
        raise ValueError("dwconv_bwd_fused: partial slab too small")
    if stride == 2:
        call("mpn_dwconv_bwd_fused_s2", ptr(x), ptr(dy), ptr(w), ptr(out), ptr(wpart), N, H, W, C, dc, ptr(bn.scale), ptr(bn.shift),
             int(bn.act), ptr(bn.mean), ptr(bn.invstd), ptr(bn_part) if reduce_bn else None, ptr(addend), stream_ptr())
    else:
        call("mpn_dwconv_bwd_fused", ptr(x), ptr(dy), ptr(w), ptr(out), ptr(wpart), N, H, W, C, dc, ptr(bn.scale), ptr(bn.shift),
             int(bn.act), ptr(bn.mean), ptr(bn.invstd), ptr(bn_part) if reduce_bn else None, stream_ptr())
    if reduce:
        call("mpn_reduce_partials", ptr(wpart), rows, 9 * C, ptr(dw_out), 0, 1.0, stream_ptr())
    return out, (rows if reduce_bn else 0)


# ----------------------------------------------------------------------------- stem
def stem_conv_fwd_num_parts(N, H, W, c0, dtype):
    """Rows of the statistics slab stem_conv_fwd(stats_part=...) writes; 0 = fused statistics not available for this c0."""
    return _lib.lib().mpn_stem_conv_fwd_num_parts(N, H, W, c0, _lib.dtype_code(dtype))


def stem_conv_fwd(images, w, c0, dtype, out=None, stats_part=None):
    """stats_part: f32 slab of stem_conv_fwd_num_parts(...) * 2 * c0 floats - the kernel also writes the batch-norm partial
    sums of its output there (finish with bn_finalize(bn, stats_part, rows, count))."""
    if images.dim() != 4 or images.shape[3] != 3 or not images.is_contiguous():
        raise ValueError("images must be contiguous [N,H,W,3]")
    u8 = images.dtype == torch.uint8
    if not u8 and images.dtype != torch.float32:
        raise ValueError("images must be float32 in [0,1] or uint8")
    N, H, W, _ = images.shape
    if out is None:
        out = torch.empty((N, (H + 1) // 2, (W + 1) // 2, c0), dtype=dtype, device=images.device)
    if stats_part is not None:
        rows = stem_conv_fwd_num_parts(N, H, W, c0, dtype)
        if rows <= 0 or stats_part.numel() < rows * 2 * c0:
            raise ValueError("stem_conv_fwd: fused statistics not available for this shape, or slab too small")
        call("mpn_stem_conv_fwd_stats", ptr(images), int(u8), ptr(w), ptr(out), N, H, W, c0, _lib.dtype_code(dtype),
             ptr(stats_part), stream_ptr())
        return out
    call("mpn_stem_conv_fwd", ptr(images), int(u8), ptr(w), ptr(out), N, H, W, c0, _lib.dtype_code(dtype), stream_ptr())
    return out


def stem_conv_bwd_weight(images, dy, dw_out, part=None, reduce=True):
    N, H, W, _ = images.shape
    c0 = dy.shape[3]
    nparts = _lib.lib().mpn_stem_conv_wgrad_num_parts(N, H, W)
    if part is None:
        part = _f32(nparts * 27 * c0, dy.device)
    call("mpn_stem_conv_bwd_weight", ptr(images), int(images.dtype == torch.uint8), ptr(dy), ptr(part), N, H, W, c0,
         _lib.dtype_code(dy.dtype), stream_ptr())
    if reduce:
        call("mpn_reduce_partials", ptr(part), nparts, 27 * c0, ptr(dw_out), 0, 1.0, stream_ptr())
    return dw_out


# ----------------------------------------------------------------------------- resizes
def bilinear_up_fwd(x, upsample, out, channel_offset, affine=None):
    N, h, w, C = x.shape
    sc, sh, act = _aff(affine)
    call("mpn_bilinear_up_fwd", ptr(x), ptr(out), N, h, w, C, upsample, channel_offset, out.shape[3],
         _lib.dtype_code(x.dtype), sc, sh, act, stream_ptr())
    return out


def bilinear_up_bwd(dy, upsample, channel_offset, C, out=None):
    N, OH, OW, ctot = dy.shape
    h, w = OH // upsample, OW // upsample
    if out is None:
        out = torch.empty((N, h, w, C), dtype=dy.dtype, device=dy.device)
    call("mpn_bilinear_up_bwd", ptr(dy), ptr(out), N, h, w, C, upsample, channel_offset, ctot, _lib.dtype_code(dy.dtype),
         stream_ptr())
    return out


def sumpool2x2(src, dst=None, accumulate=False):
    N, H2, W2, C = src.shape
    if dst is None:
        dst = torch.empty((N, H2 // 2, W2 // 2, C), dtype=src.dtype, device=src.device)
    call("mpn_sumpool2x2", ptr(src), ptr(dst), N, H2 // 2, W2 // 2, C, int(accumulate), _lib.dtype_code(src.dtype), stream_ptr())
    return dst


def add_inplace(dst, src):
    if dst.shape != src.shape or dst.dtype != src.dtype:
        raise ValueError("add_inplace: shape/dtype mismatch")
    call("mpn_add_inplace", ptr(dst), ptr(src), dst.numel(), _lib.dtype_code(dst.dtype), stream_ptr())
    return dst


# ----------------------------------------------------------------------------- head, loss, optimizer
def heatmap_head_fwd(x, w, bias, affine, inference=False, out=None, out_seg=None):
    N, H, W, cin = x.shape
    M = N * H * W
    sc, sh, act = _aff(affine)
    if inference:
        if out is None:
            out = torch.empty((N, H, W, 17), dtype=torch.float32, device=x.device)
        if out_seg is None:
            out_seg = torch.empty((N, H, W), dtype=torch.float32, device=x.device)
    elif out is None:
        out = torch.empty((N, H, W, 18), dtype=torch.float32, device=x.device)
    call("mpn_heatmap_head_fwd", ptr(x), ptr(w), ptr(bias), M, cin, _lib.dtype_code(x.dtype), sc, sh, act,
         int(inference), ptr(out), ptr(out_seg), stream_ptr())
    return (out, out_seg) if inference else out


def heatmap_head_bwd_bn_supported(cin, dtype):
    return bool(_lib.lib().mpn_heatmap_head_bwd_bn_supported(int(cin), _lib.dtype_code(dtype)))


def heatmap_head_bwd(x, dlogits, w, affine, dA, dw_db_out, part=None, reduce=True, bn_part=None):
    """dA <- gradient w.r.t. the activated input; dw_db_out: flat f32 view [Cin*18 + 18].
    bn_part (f32, heatmap_head_bwd_num_parts(M) * 2 * Cin): also run the reduction of the batch-norm behind `affine` - dA
    comes out masked by its activation and the slab takes the per-block sums of g and g * x (raw x); returns the number
    of slab rows (bn_backward(..., reduced_parts=rows, raw=True) finishes)."""
    N, H, W, cin = x.shape
    M = N * H * W
    nparts = _lib.lib().mpn_heatmap_head_bwd_num_parts(M)
    nout = cin * 18 + 18
    if part is None:
        part = _f32(nparts * nout, x.device)
    sc, sh, act = _aff(affine)
    if bn_part is not None:
        if bn_part.numel() < nparts * 2 * cin:
            raise ValueError("heatmap_head_bwd: bn_part too small")
        call("mpn_heatmap_head_bwd_bn", ptr(x), ptr(dlogits), ptr(w), M, cin, _lib.dtype_code(x.dtype), sc, sh, act, ptr(dA),
             ptr(part), ptr(bn_part), stream_ptr())
    else:
        call("mpn_heatmap_head_bwd", ptr(x), ptr(dlogits), ptr(w), M, cin, _lib.dtype_code(x.dtype), sc, sh, act, ptr(dA),
             ptr(part), stream_ptr())
    if reduce:
        call("mpn_reduce_partials", ptr(part), nparts, nout, ptr(dw_db_out), 0, 1.0, stream_ptr())
    return nparts if bn_part is not None else dA


LOSS_NAMES = ["focal_loss", "regression_loss", "segmentation_loss_at_level_2", "segmentation_loss_at_level_3",
              "segmentation_loss_at_level_4", "segmentation_loss_at_level_5", "total_loss", "per_pixel_reg_loss"]


def keypoint_loss(logits, labels, ps, dlogits=None, daux=None, part=None, losses_out=None):
    """labels: dict of device tensors (heatmaps f32 [B,h,w,17], loss_masks, segmentation_masks f32 [B,h,w],
    num_boxes int32 [B]); ps: [p2,p3,p4,p5] raw FPN outputs (NHWC). Returns f32[8] device tensor (LOSS_NAMES)."""
    B, h, w, _ = logits.shape
    dev = logits.device
    nparts = _lib.lib().mpn_keypoint_loss_num_parts(B, h, w)
    if part is None:
        part = _f32(nparts * 8, dev)
    if losses_out is None:
        losses_out = _f32(8, dev)
    da = daux if daux is not None else [None] * 4
    call("mpn_keypoint_loss", ptr(logits), ptr(labels["heatmaps"]), ptr(labels["loss_masks"]),
         ptr(labels["segmentation_masks"]), ptr(labels["num_boxes"]), ptr(ps[0]), ptr(ps[1]), ptr(ps[2]), ptr(ps[3]),
         ps[0].shape[3], _lib.dtype_code(ps[0].dtype), ptr(dlogits), ptr(da[0]), ptr(da[1]), ptr(da[2]), ptr(da[3]),
         ptr(part), ptr(losses_out), B, h, w, stream_ptr())
    return losses_out


def adam_prepare(step, hyper, initial_learning_rate, num_steps, alpha=1e-4, beta1=0.9, beta2=0.999):
    call("mpn_adam_prepare", ptr(step), ptr(hyper), float(initial_learning_rate), float(num_steps), float(alpha),
         float(beta1), float(beta2), stream_ptr())


def adam_step(params, grads, m, v, hyper, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-8, clip=200.0):
    call("mpn_adam_step", ptr(params), ptr(grads), ptr(m), ptr(v), params.numel(), ptr(hyper), beta1, beta2, eps, clip,
         float(grad_scale), stream_ptr())


def gemm_nt_num_parts(k):
    return _lib.lib().mpn_gemm_nt_num_parts(int(k))


def gemm_nt(a, b, out, slab):
    """out[M,N] (f32) = a[M,K] @ b[N,K]^T, a / b 16-bit with K contiguous (mpn_gemm_nt + the fixed-order slab reduction).
    slab: f32, gemm_nt_num_parts(K) * M * N elements."""
    (M, K), (N, K2) = a.shape, b.shape
    if K != K2 or a.dtype != b.dtype or not (a.is_contiguous() and b.is_contiguous()):
        raise ValueError("gemm_nt: a [M,K] and b [N,K] must be contiguous, of one 16-bit dtype")
    parts = gemm_nt_num_parts(K)
    if slab.numel() < parts * M * N:
        raise ValueError("gemm_nt: slab too small")
    call("mpn_gemm_nt", ptr(a), ptr(b), ptr(slab), M, N, K, _lib.dtype_code(a.dtype), stream_ptr())
    call("mpn_reduce_partials", ptr(slab), parts, M * N, ptr(out), 0, 1.0, stream_ptr())
    return out


class AdamCastJobs:
    """Ranges of a flat f32 arena whose updated values adam_step also writes as 16-bit operand copies (mpn_adam_step_cast):
    jobs = [(offset, count, dst tensor of dtype float16 / bfloat16 with `count` elements)]."""

    def __init__(self, jobs):
        n = len(jobs)
        assert 1 <= n <= 4 and len({d.dtype for _, _, d in jobs}) == 1
        assert all(d.numel() == c and d.is_contiguous() for _, c, d in jobs)
        self._keep = [d for _, _, d in jobs]
        self.n = n
        self.begin = (ctypes.c_longlong * n)(*[int(o) for o, _, _ in jobs])
        self.count = (ctypes.c_longlong * n)(*[int(c) for _, c, _ in jobs])
        self.dst = (ctypes.c_void_p * n)(*[d.data_ptr() for _, _, d in jobs])
        self.dtype = _lib.dtype_code(jobs[0][2].dtype)


def adam_step_cast(params, grads, m, v, hyper, jobs, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-8, clip=200.0):
    call("mpn_adam_step_cast", ptr(params), ptr(grads), ptr(m), ptr(v), params.numel(), ptr(hyper), beta1, beta2, eps, clip,
         float(grad_scale), jobs.n, jobs.begin, jobs.count, jobs.dst, jobs.dtype, stream_ptr())


def reduce_partials(part, nparts, n, out, accumulate=False, scale=1.0):
    call("mpn_reduce_partials", ptr(part), nparts, n, ptr(out), int(accumulate), float(scale), stream_ptr())


def axpy(a, x, y):
    call("mpn_axpy", x.numel(), float(a), ptr(x), ptr(y), stream_ptr())


class AxpyBatch:
    """ys[t] += a * xs[t] for a fixed list of float32 tensor pairs in one launch (mpn_axpy_batched)."""

    def __init__(self, xs, ys):
        n = len(xs)
        assert n == len(ys) and all(x.numel() == y.numel() for x, y in zip(xs, ys))
        self._keep = (list(xs), list(ys))
        self.n = n
        self.xp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in xs])
        self.yp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ys])
        self.counts = (ctypes.c_longlong * n)(*[t.numel() for t in xs])

    def run(self, a):
        if LAUNCH_JOBS_ALONE:
            for x, y in zip(*self._keep):
                call("mpn_axpy_batched", 1, (ctypes.c_void_p * 1)(x.data_ptr()), (ctypes.c_void_p * 1)(y.data_ptr()),
                     (ctypes.c_longlong * 1)(x.numel()), float(a), stream_ptr())
            return
        call("mpn_axpy_batched", self.n, self.xp, self.yp, self.counts, float(a), stream_ptr())


class L2LossBatch:
    """acc[0] += scale * sum of tf.nn.l2_loss over a fixed list of tensors, two launches (mpn_l2_loss_batched)."""

    def __init__(self, tensors):
        n = len(tensors)
        self._keep = list(tensors)
        self.n = n
        self.ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
        self.counts = (ctypes.c_longlong * n)(*[t.numel() for t in tensors])
        nbytes = _lib.lib().mpn_l2_loss_batched_workspace_bytes(n, self.counts)
        self.ws = torch.empty(max(8, nbytes), dtype=torch.uint8, device=tensors[0].device)

    def run(self, scale, acc):
        call("mpn_l2_loss_batched", self.n, self.ptrs, self.counts, float(scale), ptr(acc), ptr(self.ws), self.ws.numel(), stream_ptr())


def l2_loss_accumulate(w, scale, acc):
    """acc[0] += scale * sum(w^2)/2 (acc: f32 device tensor view of one element)."""
    call("mpn_l2_loss_accumulate", w.numel(), ptr(w), float(scale), ptr(acc), stream_ptr())


# ----------------------------------------------------------------------------- RetinaNet head pieces
def patchify3x3s2(x, out, affine=None):
    """x [N,H,W,C] -> out [N,ceil(H/2),ceil(W/2),9*C]: the gather behind conv2d_same(k=3, stride=2) (mpn_patchify3x3s2)."""
    _check_nhwc(x)
    N, H, W, C = x.shape
    if tuple(out.shape) != (N, (H + 1) // 2, (W + 1) // 2, 9 * C) or out.dtype != x.dtype or not out.is_contiguous():
        raise ValueError(f"patchify3x3s2: out must be contiguous {(N, (H + 1) // 2, (W + 1) // 2, 9 * C)} {x.dtype}")
    sc, sh, act = _aff(affine)
    call("mpn_patchify3x3s2", ptr(x), ptr(out), N, H, W, C, _lib.dtype_code(x.dtype), sc, sh, act, stream_ptr())
    return out


def unpatchify3x3s2(dpatches, dx):
    """Transpose of patchify3x3s2: dpatches [N,ceil(H/2),ceil(W/2),9*C] -> dx [N,H,W,C]."""
    N, H, W, C = dx.shape
    if tuple(dpatches.shape) != (N, (H + 1) // 2, (W + 1) // 2, 9 * C) or dpatches.dtype != dx.dtype:
        raise ValueError("unpatchify3x3s2: shape / dtype mismatch")
    call("mpn_unpatchify3x3s2", ptr(dpatches), ptr(dx), N, H, W, C, _lib.dtype_code(dx.dtype), stream_ptr())
    return dx
