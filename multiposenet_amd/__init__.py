"""multiposenet_amd: MI355X-native (gfx950) keypoint hot path of MultiPoseNet.

Host code mirrors the reference's Python surface (module and function names follow
TropComplique/MultiPoseNet); all device arithmetic runs in hand-written HIP kernels
behind the C ABI of include/mpn.h (libmpn_hip.so).
"""
__version__ = "0.1.0"
