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
    // weight rows padded to 20 floats: the 18 weights of a channel are five 16-byte LDS reads (broadcast) instead of 18
    // dword reads - one read per FMA made the LDS instruction rate the bound (1152 reads per pixel)
    constexpr int kWl = 20;
    __shared__ __attribute__((aligned(16))) float wl[kMaxCin * kWl];
    __shared__ float scl[kMaxCin], shl[kMaxCin];
    __shared__ __attribute__((aligned(16))) float ot[kThreads * kOut];
    for (int i = threadIdx.x; i < Cin * kWl; i += kThreads) {
        const int c = i / kWl, k = i - c * kWl;
        wl[i] = k < kOut ? w[c * kOut + k] : 0.f;
    }
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
                const float4* wr = reinterpret_cast<const float4*>(&wl[(c0 + j) * kWl]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w4 = wr[q];
                    acc[4 * q] += t * w4.x; acc[4 * q + 1] += t * w4.y; acc[4 * q + 2] += t * w4.z; acc[4 * q + 3] += t * w4.w;
                }
                const float4 w5 = wr[4];
                acc[16] += t * w5.x; acc[17] += t * w5.y;
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

__device__ __forceinline__ void store4(float* p, const float (&f)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&f)[4]) {
    const bf16_t a = (bf16_t)f[0], b = (bf16_t)f[1], c = (bf16_t)f[2], d = (bf16_t)f[3];
    uint2 q;
    q.x = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
    q.y = (unsigned)__builtin_bit_cast(unsigned short, c) | ((unsigned)__builtin_bit_cast(unsigned short, d) << 16);
    *reinterpret_cast<uint2*>(p) = q;
}

// backward: dA[m][c] = sum_k dl[m][k] W[c][k];  dW[c][k] = sum_m a[m][c] dl[m][k];  db[k] = sum_m dl[m][k]
// Blocks walk 128-pixel tiles staged in LDS (activated input f32 [128][Cin+4], dlogits [128][20]).
//   data gradient:   a thread keeps the 4 x 18 weights of ITS 4-channel group in registers and walks the tile's pixels:
//                    18 LDS reads (the pixel's dlogits, broadcast) per 72 FMAs;
//   weight gradient: 3 pixel sets x (Cin/4 channel groups x 5 groups of 4 outputs): a thread accumulates a 4 x 4 block
//                    of dW over every third pixel from two 16-byte LDS reads per 16 FMAs (one read per FMA operand made
//                    the LDS the bottleneck: 2 b32 reads per FMA = 150 us of LDS time per launch); the three sets are
//                    summed through LDS once per block; the bias gradient is 18 lanes summing dlogits columns.
constexpr int kBwdPix = 128;
constexpr int kDls = 20;                     // dlogits row stride in LDS (16-byte aligned groups of 4)
template <typename T>
__global__ __launch_bounds__(kThreads) void head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dl,
                                                            const float* __restrict__ w, long long M, int Cin,
                                                            const float* __restrict__ sc, const float* __restrict__ sh,
                                                            int act, T* __restrict__ dA, float* __restrict__ part) {
    constexpr int VE = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int as = Cin + 4;
    float* at = smem;                        // [128][Cin+4] activated input
    float* dlt = at + kBwdPix * as;          // [128][20]: 18 dlogits + 2 zeros
    float* scl = dlt + kBwdPix * kDls;       // [Cin] batch-norm scale, [Cin] shift of the input
    float* shl = scl + Cin;
    const int nout = Cin * kOut + kOut;
    for (int i = threadIdx.x; i < Cin; i += kThreads) { scl[i] = sc ? sc[i] : 1.f; shl[i] = sc ? sh[i] : 0.f; }
    const float lo = (sc && act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (sc && act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    const int cvec = Cin / VE;
    const int ncg = Cin / 4;                 // 4-channel groups per pixel (<= 32)
    // data-gradient role: channel group + first pixel
    const int d_cg = threadIdx.x % ncg, d_px0 = threadIdx.x / ncg, d_step = kThreads / ncg;
    float wreg[4][kOut];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < kOut; ++k) wreg[j][k] = w[(d_cg * 4 + j) * kOut + k];
    // weight-gradient role: (pixel set, k group, channel group); threads beyond 3 sets idle in that phase
    const int per_set = ncg * 5;
    const int nsets = kThreads / per_set < 1 ? 1 : kThreads / per_set;     // 3 for Cin = 64
    const int w_set = threadIdx.x / per_set, w_r = threadIdx.x % per_set;
    const int w_kg = w_r / ncg, w_cg = w_r % ncg;
    const bool w_on = w_set < nsets;
    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
    float accb = 0.f;

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
                // scale / shift from their LDS copies: per-element predicated global loads (`sc ? f*sc[c]+sh[c] : f`)
                // are issued as 2*VE strided dword loads and bound the kernel on the texture unit
#pragma unroll
                for (int j = 0; j < VE; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * scl[vg * VE + j] + shl[vg * VE + j], lo, hi);
            }
#pragma unroll
            for (int j = 0; j < VE; j += 4)
                *reinterpret_cast<float4*>(at + px * as + vg * VE + j) = make_float4(f[j], f[j + 1], f[j + 2], f[j + 3]);
        }
        for (int i = threadIdx.x; i < kBwdPix * kDls; i += kThreads) {
            const int px = i / kDls, k = i % kDls;
            dlt[i] = (k < kOut && m0 + px < M) ? dl[(m0 + px) * kOut + k] : 0.f;
        }
        __syncthreads();
        // (1) data gradient
        for (int px = d_px0; px < kBwdPix; px += d_step) {
            if (m0 + px >= M) break;
            float g[kDls];
#pragma unroll
            for (int k = 0; k < kDls; k += 4) {
                const float4 q = *reinterpret_cast<const float4*>(dlt + px * kDls + k);
                g[k] = q.x; g[k + 1] = q.y; g[k + 2] = q.z; g[k + 3] = q.w;
            }
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < kOut; ++k) sum += g[k] * wreg[j][k];
                o[j] = sum;
            }
            store4(dA + (m0 + px) * Cin + d_cg * 4, o);
        }
        // (2) weight gradient: 4 channels x 4 outputs per thread over every nsets-th pixel (zero rows beyond M)
        if (w_on) {
            for (int px = w_set; px < kBwdPix; px += nsets) {
                const float4 a4 = *reinterpret_cast<const float4*>(at + px * as + w_cg * 4);
                const float4 d4 = *reinterpret_cast<const float4*>(dlt + px * kDls + w_kg * 4);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[j][i] += av[j] * dv[i];
            }
        }
        // bias gradient: lane k sums column k
        if (threadIdx.x < kOut)
            for (int px = 0; px < kBwdPix; ++px) accb += dlt[px * kDls + threadIdx.x];
    }
    // ---- sum the pixel sets through LDS and write this block's partial row [Cin*18 dW | 18 db]
    __syncthreads();
    float* red = smem;                       // [nsets][per_set][16]  (<= 3 * 160 * 16 floats, fits in `at`)
    if (w_on) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(red + (w_set * per_set + w_r) * 16 + j * 4) =
                make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    }
    __syncthreads();
    float* dst = part + (long long)blockIdx.x * nout;
    for (int o = threadIdx.x; o < Cin * kOut; o += kThreads) {
        const int c = o / kOut, kk = o % kOut;
        const int r = (kk >> 2) * ncg + (c >> 2), e = (c & 3) * 4 + (kk & 3);
        float sum = 0.f;
        for (int st = 0; st < nsets; ++st) sum += red[(st * per_set + r) * 16 + e];
        dst[o] = sum;
    }
    if (threadIdx.x < kOut) dst[Cin * kOut + threadIdx.x] = accb;
}

int check(long long M, int Cin, int dtype) {
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "head: dtype %d", dtype);
    const int ve = dtype == MPN_F32 ? 4 : 8;
    MPN_REQUIRE(M > 0 && Cin > 0 && Cin <= kMaxCin && Cin % ve == 0 && (kThreads % (Cin / 4)) == 0, MPN_ERR_BAD_SHAPE,
                "head: Cin (%d) must be <= %d, a multiple of %d, and Cin/4 must divide %d", Cin, kMaxCin, ve, kThreads);
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
    const int cap = 768;   // three resident blocks per CU (45 KB of LDS each): one more to hide the staging of the others
    return (int)(ntiles < cap ? ntiles : cap);
}

/* dA [M][Cin] (storage dtype); part [num_parts][Cin*18 + 18] f32: dW then db partials */
extern "C" int mpn_heatmap_head_bwd(const void* x, const float* dlogits, const float* w, long long M, int Cin, int dtype,
                                    const float* in_scale, const float* in_shift, int in_act, void* dA, float* part,
                                    mpn_stream_t stream) {
    if (int rc = check(M, Cin, dtype)) return rc;
    MPN_REQUIRE(x && dlogits && w && dA && part, MPN_ERR_BAD_ARG, "head_bwd: null pointer");
    const int grid = mpn_heatmap_head_bwd_num_parts(M);
    const size_t sm = (size_t)(kBwdPix * (Cin + 4) + kBwdPix * kDls + 2 * Cin) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, {
        if (sm > 48 * 1024)
            MPN_HIP(hipFuncSetAttribute((const void*)head_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
        head_bwd_kernel<T><<<grid, kThreads, sm, st>>>((const T*)x, dlogits, w, M, Cin, in_scale, in_shift, in_act, (T*)dA, part);
    });
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
