// Internal interface of pointwise.hip (the deep 1x1 layers as a plain bf16 GEMM); called by conv_mfma.hip only.
#pragma once
#include <hip/hip_runtime.h>

// does mpn_conv_fwd route a (K = GEMM depth, N = output channels) layer of this geometry to the GEMM kernel? The weight
// packer must agree: such layers are packed as a plain [N][K] bf16 matrix instead of the tiled LDS image.
__attribute__((visibility("hidden"))) bool pw_gemm_eligible(int K, int N, int taps, int es);
__attribute__((visibility("hidden"))) int pw_gemm_launch(const void* x, const void* w_nk, void* y, long long M, int K, int N,
                                                         int x_stride, int y_stride, const float* in_scale,
                                                         const float* in_shift, int in_act, float* stats_part, hipStream_t st,
                                                         const void* bnr_x = nullptr, int bnr_xs = 0, const float* bnr_scale = nullptr,
                                                         const float* bnr_shift = nullptr, int bnr_act = 0);
