// K11: the `heatmaps` head: 1x1 conv Cin(64) -> 18 with bias, NHWC f32 logits out.
// Replaces tf.layers.conv2d(x, 18, 1, bias) + the NCHW->NHWC transpose at
// detector/keypoint_subnet.py:49-58 and, for inference, the post-ops of create_pb.py:73-76
// (sigmoid on channels 0..16, channel 17 raw). Forward applies final_bn + ReLU on load.
// HBM-bound (19 MMAC/image): one thread per pixel, weights broadcast from LDS.
#include "common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kOut = 18;
constexpr int kMaxCin = 128;

template <typename T>
__global__ __launch_bounds__(kThreads) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, long long M, int Cin,
                                                            const float* __restrict__ sc, const float* __restrict__ sh,
                                                            int act, int mode, float* __restrict__ out,
                                                            float* __restrict__ out_seg) {
    constexpr int VE = Vec16<T>::N;
    __shared__ float wl[kMaxCin * kOut];
    __shared__ float scl[kMaxCin], shl[kMaxCin];
    __shared__ __attribute__((aligned(16))) float ot[kThreads * kOut];
    for (int i = threadIdx.x; i < Cin * kOut; i += kThreads) wl[i] = w[i];
    for (int i = threadIdx.x; i < Cin; i += kThreads) { scl[i] = sc ? sc[i] : 1.f; shl[i] = sc ? sh[i] : 0.f; }
    __syncthreads();
    const long long m0 = (long long)blockIdx.x * kThreads;
    const long long m = m0 + threadIdx.x;
    float acc[kOut];
#pragma unroll
    for (int k = 0; k < kOut; ++k) acc[k] = bias[k];
    if (m < M) {
        for (int c0 = 0; c0 < Cin; c0 += VE) {
            Vec16<T> v;
            v.load(x + m * Cin + c0);
            float f[VE];
            v.unpack(f);
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                float t = f[j] * scl[c0 + j] + shl[c0 + j];
                if (act != MPN_ACT_NONE) t = fmaxf(t, 0.f);
                if (act == MPN_ACT_RELU6) t = fminf(t, 6.f);
#pragma unroll
                for (int k = 0; k < kOut; ++k) acc[k] += t * wl[(c0 + j) * kOut + k];
            }
        }
    }
    if (mode == 1) {  // inference: sigmoid(keypoint logits) [M][17] + raw segmentation [M]
#pragma unroll
        for (int k = 0; k < kOut - 1; ++k) ot[threadIdx.x * (kOut - 1) + k] = 1.0f / (1.0f + expf(-acc[k]));
        if (m < M) out_seg[m] = acc[kOut - 1];
        __syncthreads();
        const long long nvalid = (M - m0 < kThreads ? M - m0 : kThreads) * (kOut - 1);
        for (int i = threadIdx.x; i < nvalid; i += kThreads) out[m0 * (kOut - 1) + i] = ot[i];
    } else {
#pragma unroll
        for (int k = 0; k < kOut; ++k) ot[threadIdx.x * kOut + k] = acc[k];
        __syncthreads();
        const long long nvalid = (M - m0 < kThreads ? M - m0 : kThreads) * kOut;
        float* dst = out + m0 * kOut;  // m0*18*4 bytes is 16-byte aligned
        for (int i = threadIdx.x * 4; i < nvalid; i += kThreads * 4) {
            if (i + 4 <= nvalid) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(ot + i);
            else for (int j = i; j < nvalid; ++j) dst[j] = ot[j];
        }
    }
}

// backward: dA[m][c] = sum_k dl[m][k] W[c][k];  dW[c][k] = sum_m a[m][c] dl[m][k];  db[k] = sum_m dl[m][k]
constexpr int kBwdPix = 128;
template <typename T>
__global__ __launch_bounds__(kThreads) void head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dl,
                                                            const float* __restrict__ w, long long M, int Cin,
                                                            const float* __restrict__ sc, const float* __restrict__ sh,
                                                            int act, T* __restrict__ dA, float* __restrict__ part) {
    constexpr int VE = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int as = Cin + 1;
    float* at = smem;                        // [128][Cin+1] activated input
    float* dlt = at + kBwdPix * as;          // [128][19]
    float* wl = dlt + kBwdPix * (kOut + 1);  // [Cin][18]
    constexpr int kOPT = (kMaxCin * kOut + kOut + kThreads - 1) / kThreads;  // outputs per thread (<= 10)
    float acc[kOPT];
#pragma unroll
    for (int k = 0; k < kOPT; ++k) acc[k] = 0.f;
    const int nout = Cin * kOut + kOut;
    for (int i = threadIdx.x; i < Cin * kOut; i += kThreads) wl[i] = w[i];
    const int cvec = Cin / VE;
    const long long ntiles = (M + kBwdPix - 1) / kBwdPix;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long m0 = t * kBwdPix;
        __syncthreads();
        for (int i = threadIdx.x; i < kBwdPix * cvec; i += kThreads) {
            const int px = i / cvec, vg = i % cvec;
            float f[VE];
#pragma unroll
            for (int j = 0; j < VE; ++j) f[j] = 0.f;
            if (m0 + px < M) {
                Vec16<T> v;
                v.load(x + (m0 + px) * Cin + vg * VE);
                v.unpack(f);
#pragma unroll
                for (int j = 0; j < VE; ++j) {
                    float q = sc ? f[j] * sc[vg * VE + j] + sh[vg * VE + j] : f[j];
                    if (act != MPN_ACT_NONE) q = fmaxf(q, 0.f);
                    if (act == MPN_ACT_RELU6) q = fminf(q, 6.f);
                    f[j] = q;
                }
            }
#pragma unroll
            for (int j = 0; j < VE; ++j) at[px * as + vg * VE + j] = f[j];
        }
        for (int i = threadIdx.x; i < kBwdPix * kOut; i += kThreads) {
            const int px = i / kOut, k = i % kOut;
            dlt[px * (kOut + 1) + k] = (m0 + px < M) ? dl[(m0 + px) * kOut + k] : 0.f;
        }
        __syncthreads();
        // (1) data gradient: each thread = (pixel, channel vector) pairs
        for (int i = threadIdx.x; i < kBwdPix * cvec; i += kThreads) {
            const int px = i / cvec, vg = i % cvec;
            if (m0 + px >= M) continue;
            float o[VE];
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < kOut; ++k) s += dlt[px * (kOut + 1) + k] * wl[(vg * VE + j) * kOut + k];
                o[j] = s;
            }
            Vec16<T> ov;
            ov.pack(o);
            ov.store(dA + (m0 + px) * Cin + vg * VE);
        }
        // (2) weight / bias gradient
#pragma unroll
        for (int k = 0; k < kOPT; ++k) {
            const int o = threadIdx.x + k * kThreads;
            if (o < nout) {
                float s = 0.f;
                if (o < Cin * kOut) {
                    const int c = o / kOut, kk = o % kOut;
                    for (int px = 0; px < kBwdPix; ++px) s += at[px * as + c] * dlt[px * (kOut + 1) + kk];
                } else {
                    const int kk = o - Cin * kOut;
                    for (int px = 0; px < kBwdPix; ++px) s += dlt[px * (kOut + 1) + kk];
                }
                acc[k] += s;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kOPT; ++k) {
        const int o = threadIdx.x + k * kThreads;
        if (o < nout) part[(long long)blockIdx.x * nout + o] = acc[k];
    }
}

int check(long long M, int Cin, int dtype) {
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "head: dtype %d", dtype);
    const int ve = dtype == MPN_F32 ? 4 : 8;
    MPN_REQUIRE(M > 0 && Cin > 0 && Cin <= kMaxCin && Cin % ve == 0, MPN_ERR_BAD_SHAPE,
                "head: Cin (%d) must be <= %d and a multiple of %d", Cin, kMaxCin, ve);
    return MPN_OK;
}
}  // namespace

/* mode 0: out = logits [M][18] f32.  mode 1 (inference, create_pb.py:73-76): out = sigmoid(logits[:, :17]) [M][17],
 * out_seg = logits[:, 17] [M]. */
extern "C" int mpn_heatmap_head_fwd(const void* x, const float* w, const float* bias, long long M, int Cin, int dtype,
                                    const float* in_scale, const float* in_shift, int in_act, int mode, float* out,
                                    float* out_seg, mpn_stream_t stream) {
    if (int rc = check(M, Cin, dtype)) return rc;
    MPN_REQUIRE(x && w && bias && out && (mode == 0 || out_seg), MPN_ERR_BAD_ARG, "head_fwd: null pointer");
    MPN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), MPN_ERR_BAD_ARG, "head_fwd: scale/shift mismatch");
    const int grid = (int)((M + kThreads - 1) / kThreads);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (head_fwd_kernel<T><<<grid, kThreads, 0, st>>>((const T*)x, w, bias, M, Cin, in_scale, in_shift,
                                                                            in_act, mode, out, out_seg)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_heatmap_head_bwd_num_parts(long long M) {
    const long long ntiles = (M + kBwdPix - 1) / kBwdPix;
    return (int)(ntiles < 512 ? ntiles : 512);
}

/* dA [M][Cin] (storage dtype); part [num_parts][Cin*18 + 18] f32: dW then db partials */
extern "C" int mpn_heatmap_head_bwd(const void* x, const float* dlogits, const float* w, long long M, int Cin, int dtype,
                                    const float* in_scale, const float* in_shift, int in_act, void* dA, float* part,
                                    mpn_stream_t stream) {
    if (int rc = check(M, Cin, dtype)) return rc;
    MPN_REQUIRE(x && dlogits && w && dA && part, MPN_ERR_BAD_ARG, "head_bwd: null pointer");
    const int grid = mpn_heatmap_head_bwd_num_parts(M);
    const size_t sm = (size_t)(kBwdPix * (Cin + 1) + kBwdPix * (kOut + 1) + Cin * kOut) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, {
        if (sm > 48 * 1024)
            MPN_HIP(hipFuncSetAttribute((const void*)head_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
        head_bwd_kernel<T><<<grid, kThreads, sm, st>>>((const T*)x, dlogits, w, M, Cin, in_scale, in_shift, in_act, (T*)dA, part);
    });
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
