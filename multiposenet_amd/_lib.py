"""ctypes binding of libmpn_hip.so (the C ABI declared in include/mpn.h).

There is NO fallback: if the shared library is missing or a call fails, the product
path raises. (The CPU restatement under oracle/ is test infrastructure only.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPN_LIB: an alternative build of the same library (diagnostic A/B of compile-time variants, tools/build_variant.sh)
LIB_PATH = os.environ.get("MPN_LIB") or os.path.join(_HERE, "libmpn_hip.so")

MPN_VERSION = 600     # the ABI revision this binding was written against (include/mpn.h); lib() refuses another
MPN_F32, MPN_BF16, MPN_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_F = _c.c_float
_D = _c.c_double
_Z = _c.c_size_t
_L = _c.c_longlong

# name -> (restype, argtypes); filled in below, checked against include/mpn.h by the tests
SIGNATURES = {
    "mpn_version": (_I, []),
    "mpn_last_error": (_I, [_c.c_char_p, _Z]),
    "mpn_heatmap_decode_workspace_bytes": (_Z, [_I]),
    "mpn_heatmap_decode": (_I, [_P, _I, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P, _Z, _P]),
    "mpn_heatmap_render_workspace_bytes": (_Z, [_I]),
    "mpn_heatmap_render": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "mpn_conv_packed_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "mpn_conv_pack_weights": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "mpn_conv_pack_desc_bytes": (_Z, []),
    "mpn_conv_pack_desc_fill": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I]),
    "mpn_conv_pack_weights_batched": (_I, [_P, _I, _I, _I, _P]),
    "mpn_conv_num_parts": (_I, [_I, _I, _I, _I]),
    "mpn_conv_stats_rows": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "mpn_conv_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "mpn_conv_fwd_grouped": (_I, [_I, _P, _P, _P, _I, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _I, _P, _P]),
    "mpn_conv_wgrad_num_parts": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "mpn_conv_bwd_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "mpn_conv1x1_bwd_fused_supported": (_I, [_I, _I, _I]),
    "mpn_conv1x1_bwd_fused": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "mpn_conv1x1_bwd_fused_apply_supported": (_I, [_I, _I, _I]),
    "mpn_conv1x1_bwd_fused_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "mpn_conv_wgrad_grouped_num_parts": (_I, [_I, _I, _P, _P, _I, _I, _I, _I, _P]),
    "mpn_conv_bwd_weight_grouped": (_I, [_I, _P, _P, _P, _I, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _I, _P]),
    "mpn_bn_stats_num_parts": (_I, [_L]),
    "mpn_bn_stats": (_I, [_P, _L, _I, _I, _P, _P]),
    "mpn_bn_finalize": (_I, [_P, _I, _I, _L, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "mpn_bn_inference_affine": (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P]),
    "mpn_bn_act_apply": (_I, [_P, _P, _L, _I, _I, _P, _P, _I, _P]),
    "mpn_bn_bwd_reduce": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "mpn_bn_bwd_finalize": (_I, [_P, _I, _I, _L, _P, _P, _P, _P, _P]),
    "mpn_bn_bwd_apply": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "mpn_dwconv_out_size": (_I, [_I, _I]),
    "mpn_dwconv_num_parts": (_I, [_I, _I, _I, _I, _I, _I]),
    "mpn_dwconv_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P]),
    "mpn_dwconv_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mpn_dwconv_bwd_data_bn_num_parts": (_I, [_I, _I, _I, _I, _I, _I]),
    "mpn_dwconv_bwd_data_bn": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "mpn_dwconv_bwd_data_add_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "mpn_dwconv_bwd_data_add": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "mpn_dwconv_wgrad_num_parts": (_I, [_I, _I, _I, _I, _I, _I]),
    "mpn_dwconv_bwd_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "mpn_dwconv_bwd_fused_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "mpn_dwconv_bwd_fused": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "mpn_dwconv_bwd_fused_s2": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P]),
    "mpn_stem_conv_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mpn_stem_conv_fwd_num_parts": (_I, [_I, _I, _I, _I, _I]),
    "mpn_stem_conv_fwd_stats": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mpn_stem_conv_wgrad_num_parts": (_I, [_I, _I, _I]),
    "mpn_stem_conv_bwd_weight": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mpn_bilinear_up_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "mpn_bilinear_up_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mpn_sumpool2x2": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mpn_add_inplace": (_I, [_P, _P, _L, _I, _P]),
    "mpn_heatmap_head_fwd": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _I, _I, _P, _P, _P]),
    "mpn_heatmap_head_bwd_num_parts": (_I, [_L]),
    "mpn_heatmap_head_bwd": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _I, _P, _P, _P]),
    "mpn_heatmap_head_bwd_bn_supported": (_I, [_I, _I]),
    "mpn_heatmap_head_bwd_bn": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "mpn_keypoint_loss_num_parts": (_I, [_I, _I, _I]),
    "mpn_keypoint_loss": (_I, [_P] * 9 + [_I, _I] + [_P] * 7 + [_I, _I, _I, _P]),
    "mpn_adam_prepare": (_I, [_P, _P, _D, _D, _D, _D, _D, _P]),
    "mpn_adam_step": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _P]),
    "mpn_adam_step_cast": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _I, _P, _P, _P, _I, _P]),
    "mpn_reduce_partials": (_I, [_P, _I, _L, _P, _I, _F, _P]),
    "mpn_reduce_desc_bytes": (_Z, []),
    "mpn_reduce_desc_fill": (_I, [_P, _P, _I, _L, _P, _F, _I]),
    "mpn_reduce_partials_batched": (_I, [_P, _I, _I, _P]),
    "mpn_gemm_nt_num_parts": (_I, [_I]),
    "mpn_gemm_nt": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "mpn_transpose_cast": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "mpn_cast": (_I, [_P, _I, _P, _I, _L, _P]),
    "mpn_bias_relu_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _P]),
    "mpn_bias_relu_bwd": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _P]),
    "mpn_prn_loss": (_I, [_P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _F, _P]),
    "mpn_bn_bwd_reduce_grouped": (_I, [_I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mpn_bn_bwd_apply_grouped": (_I, [_I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mpn_bn_fin_desc_bytes": (_Z, []),
    "mpn_bn_bwd_fin_desc_bytes": (_Z, []),
    "mpn_bn_fin_desc_fill": (_I, [_P, _P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I]),
    "mpn_bn_bwd_fin_desc_fill": (_I, [_P, _P, _I, _I, _L, _P, _P, _P, _P, _I]),
    "mpn_bn_finalize_batched": (_I, [_P, _I, _I, _F, _F, _P]),
    "mpn_bn_bwd_finalize_batched": (_I, [_P, _I, _I, _P]),
    "mpn_heatmap_minmax": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "mpn_prn_crop": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "mpn_prn_decode": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "mpn_conv_bwd_data_bn_supported": (_I, [_I, _I, _I, _I]),
    "mpn_conv_bwd_data_bn": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P]),
    "mpn_bn_bwd_finalize_raw": (_I, [_P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P]),
    "mpn_conv_bwd_data_bn_grouped": (_I, [_I, _P, _P, _P, _I, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P]),
    "mpn_bn_bwd_fin_desc_fill_raw": (_I, [_P, _P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _I]),
    "mpn_prn_crop_slots": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "mpn_prn_residual": (_I, [_P, _P, _I, _L, _P, _P]),
    "mpn_retina_loss_finalize": (_I, [_P, _P, _F, _F, _P, _P, _P, _P]),
    "mpn_axpy": (_I, [_L, _F, _P, _P, _P]),
    "mpn_axpy_batched": (_I, [_I, _P, _P, _P, _F, _P]),
    "mpn_patchify3x3s2": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "mpn_unpatchify3x3s2": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "mpn_retina_match_workspace_bytes": (_Z, [_I, _I]),
    "mpn_retina_match": (_I, [_P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _Z, _P]),
    "mpn_retina_loss_num_parts": (_I, [_I, _I]),
    "mpn_retina_loss": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _P, _P]),
    "mpn_retina_nms_workspace_bytes": (_Z, [_I, _I]),
    "mpn_retina_nms_overflow_offset": (_Z, [_I, _I]),
    "mpn_retina_nms": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _F, _F, _I, _P, _P, _P, _P, _Z, _P]),
    "mpn_l2_loss_accumulate": (_I, [_L, _P, _F, _P, _P]),
    "mpn_l2_loss_batched_workspace_bytes": (_Z, [_I, _P]),
    "mpn_l2_loss_batched": (_I, [_I, _P, _P, _F, _P, _P, _Z, _P]),
}

_lib = None


class MpnError(RuntimeError):
    pass


def lib():
    """Returns the loaded CDLL; raises if libmpn_hip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MpnError(
                f"{LIB_PATH} not found: build it with `python -m multiposenet_amd.build` "
                "(there is no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (same soname as
        # /opt/rocm's). Load torch's copy first so that libmpn_hip.so binds to it and shares
        # torch's device context, streams and allocations.
        import torch
        hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(hip_rt):
            ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
        l = ctypes.CDLL(LIB_PATH)
        l.mpn_version.restype = _I
        if l.mpn_version() != MPN_VERSION:
            # exported signatures change between revisions (r3 -> r4: scratch slabs became writable, batch-norm finalizes
            # moved into consumer kernels): a stale build would be handed the wrong argument lists
            raise MpnError(f"{LIB_PATH} is ABI revision {l.mpn_version()}, this binding needs {MPN_VERSION}: "
                           "rebuild with `python -m multiposenet_amd.build --force`")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib().mpn_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


_ERR_NAMES = {-1: "BAD_SHAPE", -2: "BAD_DTYPE", -3: "BAD_ALIGN", -4: "HIP", -5: "BAD_ARG", -6: "WORKSPACE"}


def check(rc):
    if rc != 0:
        msg = f"MPN_ERR_{_ERR_NAMES.get(rc, rc)}: {last_error()}"
        if rc in (-1, -2, -3, -5):
            raise ValueError(msg)
        raise MpnError(msg)


# bench.py's in-step family timing: while PROFILE is a list every launch is bracketed by two events on the launch stream and
# recorded as (tag, entry point, start, end); `tagged` names the call site's family where the entry point alone does not (a 1x1
# convolution of the backbone and an FPN lateral are the same entry point). None (the default): one attribute test per call.
PROFILE = None
_TAG = ""


class tagged:
    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        global _TAG
        self.prev, _TAG = _TAG, self.tag

    def __exit__(self, *a):
        global _TAG
        _TAG = self.prev


# MPN_TRACE_CALLS=1 (debugging a device fault): every entry point is named on stderr BEFORE it launches and the device is
# synchronised behind it, so the last name printed is the launch that faulted. Not for captured streams.
TRACE_CALLS = os.environ.get("MPN_TRACE_CALLS", "0") == "1"


def call(name, *args):
    if TRACE_CALLS:
        import sys
        import torch
        print(f"[mpn] {name}", file=sys.stderr, flush=True)
        check(getattr(lib(), name)(*args))
        torch.cuda.synchronize()
        return
    if PROFILE is None:
        check(getattr(lib(), name)(*args))
        return
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(getattr(lib(), name)(*args))
    e1.record()
    PROFILE.append((_TAG, name, e0, e1))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else _P(t.data_ptr())


def stream_ptr():
    import torch
    return _P(torch.cuda.current_stream().cuda_stream)


def device_guarded(*names):
    """Class decorator: the named methods run with `self.device` as the CURRENT device. Every launch takes
    torch.cuda.current_stream() of the current device and HIP function attributes are per device, so an object built
    for cuda:1 must not launch while cuda:0 is current (kernels would run on device 0's stream against device-1 pointers)."""
    import functools

    def wrap(fn):
        @functools.wraps(fn)
        def inner(self, *a, **k):
            import torch
            dev = torch.device(self.device)
            if dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
                return fn(self, *a, **k)
            with torch.cuda.device(dev):
                return fn(self, *a, **k)
        return inner

    def deco(cls):
        for n in names:
            setattr(cls, n, wrap(getattr(cls, n)))
        return cls
    return deco


def dtype_code(torch_dtype):
    import torch
    try:
        return {torch.float32: MPN_F32, torch.bfloat16: MPN_BF16, torch.float16: MPN_F16}[torch_dtype]
    except KeyError:
        raise ValueError(f"unsupported dtype {torch_dtype}")


def current_device():
    """The CUDA device of this process (one process per GPU: the rank's torch.cuda.set_device), for shims that receive host
    data and have no network to take the device from - never a hard-coded "cuda:0", which is rank 0's card on every rank."""
    import torch
    return torch.device("cuda", torch.cuda.current_device())


def to_device_f32(x):
    """numpy array or tensor -> contiguous f32 tensor on the process's device (a CUDA tensor stays on its own device)."""
    import numpy as np
    import torch
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    dev = x.device if x.is_cuda else current_device()
    return x.to(dev, torch.float32).contiguous()
