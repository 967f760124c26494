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

// bf16 build, training logits (mode 0) on the matrix cores: logits^T[k][px] = W^T[k][c] a[c][px]. No LDS and no barriers: a lane
// loads its 8 consecutive channels of one pixel straight from HBM (a wave instruction covers 16 whole pixel rows), applies
// final_bn + ReLU, splits into bf16 hi + lo (f32 accuracy, three MFMAs per product) and feeds the B operand; the 18 outputs
// are two 16-row blocks of the A operand (W^T, hi / lo fragments in registers). A wave owns 64 consecutive pixels.
template <int CB>
__global__ __launch_bounds__(kThreads) void head_fwd_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, long long M,
                                                                 const float* __restrict__ sc, const float* __restrict__ sh,
                                                                 int act, float* __restrict__ out) {
    typedef H16<bf16_t> HT;
    typedef HT::x8 x8;
    typedef HT::acc_t acc_t;
    constexpr int Cin = 16 * CB;
    constexpr int KS = (Cin + 31) / 32;                                      // 32-channel steps (Cin = 16: the upper half is zero)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, kg = lane >> 4;
    const float lo = (sc && act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (sc && act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    // this lane's channels 32 * ks + 8 * kg .. + 7: scale / shift, and the weight fragments (row = output ob * 16 + n)
    float scv[KS][8], shv[KS][8];
    x8 w_hi[2][KS], w_lo[2][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = 32 * ks + 8 * kg + i;
            const bool cin = c < Cin;
            scv[ks][i] = (sc && cin) ? sc[c] : 1.f;
            shv[ks][i] = (sc && cin) ? sh[c] : 0.f;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const int k = ob * 16 + n;
                const float wk = (cin && k < kOut) ? w[c * kOut + k] : 0.f;
                const bf16_t h = (bf16_t)wk;
                w_hi[ob][ks][i] = h;
                w_lo[ob][ks][i] = (bf16_t)(wk - (float)h);
            }
        }
    float bv[2][4];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = ob * 16 + 4 * kg + e;
            bv[ob][e] = k < kOut ? bias[k] : 0.f;
        }
    const long long m0 = (long long)blockIdx.x * kThreads + wv * 64;
    uint4 xv[4][KS];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const long long m = m0 + 16 * g + n;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool ok = m < M && 32 * ks + 8 * kg < Cin;
            xv[g][ks] = *reinterpret_cast<const uint4*>(x + (ok ? m * Cin + 32 * ks + 8 * kg : 0));
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const long long m = m0 + 16 * g + n;
        acc_t d[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) d[ob] = (acc_t){bv[ob][0], bv[ob][1], bv[ob][2], bv[ob][3]};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Vec16<bf16_t> v;
            v.raw = xv[g][ks];
            float f[8];
            v.unpack(f);
            x8 b_hi, b_lo;
            const bool cin = 32 * ks + 8 * kg < Cin;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float a = cin ? __builtin_amdgcn_fmed3f(f[i] * scv[ks][i] + shv[ks][i], lo, hi) : 0.f;
                const bf16_t h = (bf16_t)a;
                b_hi[i] = h;
                b_lo[i] = (bf16_t)(a - (float)h);
            }
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                d[ob] = HT::mfma(w_lo[ob][ks], b_hi, d[ob]);
                d[ob] = HT::mfma(w_hi[ob][ks], b_lo, d[ob]);
                d[ob] = HT::mfma(w_hi[ob][ks], b_hi, d[ob]);
            }
        }
        if (m < M) {
            float* dst = out + m * kOut + 4 * kg;                            // (rows of 72 B: 8-byte aligned pieces)
            *reinterpret_cast<float2*>(dst) = make_float2(d[0][0], d[0][1]);
            *reinterpret_cast<float2*>(dst + 2) = make_float2(d[0][2], d[0][3]);
            if (kg == 0) *reinterpret_cast<float2*>(dst + 16) = make_float2(d[1][0], d[1][1]);
        }
    }
}

__device__ __forceinline__ void store4(float* p, const float (&f)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&f)[4]) {
    uint2 q;
    q.x = pack_bf16x2(f[0], f[1]);
    q.y = pack_bf16x2(f[2], f[3]);
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

// bf16 build: the backward pass on the matrix cores, persistent blocks over 128-pixel tiles.
//   images in LDS, pixel-major: a = act(bn(x)) and dlogits, each as bf16 hi + lo parts (v = hi + lo to 2^-17: the sums keep
//   f32 accuracy, three MFMAs per product: hi*hi + hi*lo + lo*hi); dlogits rows padded to 32 columns of zeros;
//   data gradient   dA^T[c][px] = W[c][k] dl^T[k][px]: K = 18 padded to one 32-deep step, the weights (hi / lo fragments in
//                   registers) as the A operand so that a lane ends up with 4 consecutive channels of one pixel; the dl
//                   fragment is a 16-byte row read of the images;
//   weight gradient dW[c][k] = sum over pixels a[px][c] dl[px][k]: the pixel index is K (a wave owns 32 of the tile's
//                   pixels), both operands through the transposing LDS read; the bias gradient is the same product with a
//                   row of ones;
//   BNR             (optional) the reduction of the batch-norm the head reads through: the data gradient is masked by the
//                   activation (lo < x * scale + shift < hi) before it is stored, and its per-channel sums sum g and
//                   sum g * x (raw x) go to one slab row per block - finish with mpn_bn_bwd_finalize_raw.
// The next tile's x vectors and dlogits are prefetched into registers under the current tile's products.
constexpr int kHT = 128;                     // pixels per tile
constexpr int kHD = 32;                      // dlogits columns in LDS (18 + zero padding)
template <int CB, bool BNR>
__global__ __launch_bounds__(kThreads, 2) void head_bwd_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ dl,
                                                                 const float* __restrict__ w, long long M,
                                                                 const float* __restrict__ sc, const float* __restrict__ sh,
                                                                 int act, bf16_t* __restrict__ dA, float* __restrict__ part,
                                                                 float* __restrict__ bn_part) {
    typedef H16<bf16_t> HT;
    typedef HT::x8 x8;
    typedef HT::acc_t acc_t;
    constexpr int Cin = 16 * CB;
    constexpr int CV = Cin / 8;                                              // 16-byte vectors per pixel row of x
    constexpr int ARS = Cin * 2 + 16;                                        // bytes per pixel row of the a images
    constexpr int DRS = kHD * 2 + 16;                                        // ... of the dlogits images
    constexpr int XV = kHT * CV / kThreads;                                  // x vectors per thread and tile (4 for Cin = 64)
    constexpr int NP = kHT * (kOut / 2);                                     // float2 pieces of a tile's dlogits (1152)
    constexpr int DP = (NP + kThreads - 1) / kThreads;                       // 5 per thread
    static_assert(kThreads % CV == 0 && XV >= 1, "head_bwd_mfma: channel vectors must divide the block");
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    unsigned char* a_hi = hsm;
    unsigned char* a_lo = a_hi + kHT * ARS;
    unsigned char* d_hi = a_lo + kHT * ARS;
    unsigned char* d_lo = d_hi + kHT * DRS;
    float* scl = reinterpret_cast<float*>(d_lo + kHT * DRS);                 // [Cin] scale, [Cin] shift
    float* shl = scl + Cin;
    unsigned char* x_img = reinterpret_cast<unsigned char*>(shl + Cin);      // BNR: the raw x tile (a global re-read in the
                                                                             // epilogue exposed its latency 8 times per tile)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, kg = lane >> 4, q = n >> 2, pq = n & 3;
    const float lo = (sc && act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (sc && act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    for (int i = threadIdx.x; i < Cin; i += kThreads) { scl[i] = sc ? sc[i] : 1.f; shl[i] = sc ? sh[i] : 0.f; }
    for (int i = threadIdx.x; i < kHT * DRS / 4; i += kThreads) {            // the padding columns stay zero for good
        reinterpret_cast<unsigned*>(d_hi)[i] = 0u;
        reinterpret_cast<unsigned*>(d_lo)[i] = 0u;
    }
    // weight fragments of the data gradient: row = channel cb * 16 + n, k = 8 * kg .. + 7
    x8 w_hi[CB], w_lo[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 8 * kg + i;
            const float wk = k < kOut ? w[(cb * 16 + n) * kOut + k] : 0.f;
            const bf16_t h = (bf16_t)wk;
            w_hi[cb][i] = h;
            w_lo[cb][i] = (bf16_t)(wk - (float)h);
        }
    x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;
    // staging role: x vector (pixel tid / CV + k * (256 / CV), channels 8 * vg ..), the same channel vector in every load
    const int vg = threadIdx.x % CV;
    acc_t acc[CB][2], accb[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        accb[kb] = (acc_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb][kb] = (acc_t){0.f, 0.f, 0.f, 0.f};
    }
    float s1[CB][4], s2[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[cb][e] = 0.f; s2[cb][e] = 0.f; }

    uint4 xv[XV];
    float2 dv[DP];
    auto load_tile = [&](long long m0) __attribute__((always_inline)) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));   // (keeps the index arithmetic out of the tile loop's live registers)
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            const long long m = m0 + tid / CV + k * (kThreads / CV);
            xv[k] = *reinterpret_cast<const uint4*>(x + (m < M ? m : 0) * Cin + (tid % CV) * 8);
        }
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            const int j = tid + k * kThreads;                                 // piece j: pixel j / 9, columns 2 * (j % 9) ..
            const long long e = (m0 * kOut + 2LL * j);
            dv[k] = *reinterpret_cast<const float2*>(dl + ((j < NP && e + 1 < M * kOut) ? e : 0));
        }
    };
    auto commit_tile = [&](long long m0) __attribute__((always_inline)) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            const int px = tid / CV + k * (kThreads / CV);
            Vec16<bf16_t> v;
            v.raw = xv[k];
            float f[8];
            v.unpack(f);
            x8 h8, l8;
            const bool live = m0 + px < M;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = live ? __builtin_amdgcn_fmed3f(f[j] * scl[vg * 8 + j] + shl[vg * 8 + j], lo, hi) : 0.f;
                const bf16_t h = (bf16_t)a;
                h8[j] = h;
                l8[j] = (bf16_t)(a - (float)h);
            }
            *reinterpret_cast<x8*>(a_hi + px * ARS + vg * 16) = h8;
            *reinterpret_cast<x8*>(a_lo + px * ARS + vg * 16) = l8;
            if (BNR) *reinterpret_cast<uint4*>(x_img + px * ARS + vg * 16) = xv[k];
        }
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            const int j = tid + k * kThreads;
            if (j < NP) {
                const int px = j / (kOut / 2), pr = j - px * (kOut / 2);
                const bool live = m0 + px < M;
                const float g0 = live ? dv[k].x : 0.f, g1 = live ? dv[k].y : 0.f;
                const bf16_t h0 = (bf16_t)g0, h1 = (bf16_t)g1;
                *reinterpret_cast<unsigned*>(d_hi + px * DRS + pr * 4) = pack_bf16x2(g0, g1);
                *reinterpret_cast<unsigned*>(d_lo + px * DRS + pr * 4) = pack_bf16x2(g0 - (float)h0, g1 - (float)h1);
            }
        }
    };

    const long long ntiles = (M + kHT - 1) / kHT;
    long long t = blockIdx.x;
    __syncthreads();                                                         // scale / shift and the zeroed images are in place
    if (t < ntiles) {
        load_tile(t * kHT);
        commit_tile(t * kHT);
    }
    for (; t < ntiles; t += gridDim.x) {
        const long long m0 = t * kHT;
        const bool more = t + gridDim.x < ntiles;
        if (more) load_tile((t + gridDim.x) * kHT);
        __syncthreads();                                                     // this tile's images are complete
        // ---- (1) data gradient of this wave's 32 pixels, two groups of 16
#pragma unroll 1
        for (int g = 0; g < 2; ++g) {   // (rolled: the unrolled form hoists both groups' loads - 280 VGPRs with the fused reduction)
            const int px = 32 * wv + 16 * g + n;
            const x8 b_hi = *reinterpret_cast<const x8*>(d_hi + px * DRS + kg * 16);
            const x8 b_lo = *reinterpret_cast<const x8*>(d_lo + px * DRS + kg * 16);
            const bool live = m0 + px < M;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                acc_t d = {0.f, 0.f, 0.f, 0.f};
                d = HT::mfma(w_lo[cb], b_hi, d);
                d = HT::mfma(w_hi[cb], b_lo, d);
                d = HT::mfma(w_hi[cb], b_hi, d);
                float o[4] = {d[0], d[1], d[2], d[3]};
                const int c0 = cb * 16 + 4 * kg;
                if (BNR) {
                    const uint2 raw = *reinterpret_cast<const uint2*>(x_img + px * ARS + c0 * 2);
                    const float xr[4] = {__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u),
                                         __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xffff0000u)};
                    const float4 s4 = *reinterpret_cast<const float4*>(scl + c0), h4 = *reinterpret_cast<const float4*>(shl + c0);
                    const float ss[4] = {s4.x, s4.y, s4.z, s4.w}, hh[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pre = xr[e] * ss[e] + hh[e];
                        const float gm = (live && pre > lo && pre < hi) ? o[e] : 0.f;
                        // the sums are over the ROUNDED gradient, the tensor mpn_bn_bwd_apply reads
                        const float gr = (float)(bf16_t)gm;
                        o[e] = gm;
                        s1[cb][e] += gr;
                        s2[cb][e] += gr * xr[e];
                    }
                }
                if (live) store4(dA + (m0 + px) * Cin + c0, o);
            }
        }
        // ---- (2) weight gradient: K = this wave's 32 pixels
        {
            const int row = 32 * wv + 8 * kg + q;
            x8 dh[2], dlw[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const unsigned char* ph = d_hi + row * DRS + (kb * 16 + 4 * pq) * 2;
                const unsigned char* pl = d_lo + row * DRS + (kb * 16 + 4 * pq) * 2;
                dh[kb] = __builtin_shufflevector(HT::tr_read(ph), HT::tr_read(ph + 4 * DRS), 0, 1, 2, 3, 4, 5, 6, 7);
                dlw[kb] = __builtin_shufflevector(HT::tr_read(pl), HT::tr_read(pl + 4 * DRS), 0, 1, 2, 3, 4, 5, 6, 7);
                accb[kb] = HT::mfma(ones, dlw[kb], accb[kb]);
                accb[kb] = HT::mfma(ones, dh[kb], accb[kb]);
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const unsigned char* ph = a_hi + row * ARS + (cb * 16 + 4 * pq) * 2;
                const unsigned char* pl = a_lo + row * ARS + (cb * 16 + 4 * pq) * 2;
                const x8 ah = __builtin_shufflevector(HT::tr_read(ph), HT::tr_read(ph + 4 * ARS), 0, 1, 2, 3, 4, 5, 6, 7);
                const x8 al = __builtin_shufflevector(HT::tr_read(pl), HT::tr_read(pl + 4 * ARS), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    acc[cb][kb] = HT::mfma(al, dh[kb], acc[cb][kb]);
                    acc[cb][kb] = HT::mfma(ah, dlw[kb], acc[cb][kb]);
                    acc[cb][kb] = HT::mfma(ah, dh[kb], acc[cb][kb]);
                }
            }
        }
        __syncthreads();                                                     // everybody is done reading this tile
        if (more) commit_tile((t + gridDim.x) * kHT);
    }
    // ---- the four waves' sums through LDS (fixed order): lane (n, kg) of block (cb, kb) holds dW[cb*16 + 4*kg + e][kb*16 + n]
    __syncthreads();
    float* red = reinterpret_cast<float*>(hsm);                              // [4][CB * 2 + 2][64][4]
    constexpr int NB = CB * 2 + 2;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
            *reinterpret_cast<float4*>(red + ((wv * NB + cb * 2 + kb) * 64 + lane) * 4) =
                make_float4(acc[cb][kb][0], acc[cb][kb][1], acc[cb][kb][2], acc[cb][kb][3]);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
        *reinterpret_cast<float4*>(red + ((wv * NB + CB * 2 + kb) * 64 + lane) * 4) =
            make_float4(accb[kb][0], accb[kb][1], accb[kb][2], accb[kb][3]);
    __syncthreads();
    constexpr int nout = Cin * kOut + kOut;
    float* dst = part + (long long)blockIdx.x * nout;
    for (int o = threadIdx.x; o < nout; o += kThreads) {
        float sum = 0.f;
        if (o < Cin * kOut) {
            const int c = o / kOut, k = o - c * kOut;
            const int blk = (c >> 4) * 2 + (k >> 4), ln = ((c & 15) >> 2) * 16 + (k & 15), e = c & 3;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) sum += red[((w4 * NB + blk) * 64 + ln) * 4 + e];
        } else {
            const int k = o - Cin * kOut;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) sum += red[((w4 * NB + CB * 2 + (k >> 4)) * 64 + (k & 15)) * 4];
        }
        dst[o] = sum;
    }
    if (BNR) {
        // lanes of one kg hold the same channels for 16 different pixels: butterfly over n, then the four waves through LDS
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = s1[cb][e], b = s2[cb][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                s1[cb][e] = a; s2[cb][e] = b;
            }
        __syncthreads();
        float* r2 = reinterpret_cast<float*>(hsm);                           // [4][2][Cin]
        if (n == 0) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r2[(wv * 2 + 0) * Cin + cb * 16 + 4 * kg + e] = s1[cb][e];
                    r2[(wv * 2 + 1) * Cin + cb * 16 + 4 * kg + e] = s2[cb][e];
                }
        }
        __syncthreads();
        for (int o = threadIdx.x; o < 2 * Cin; o += kThreads) {
            float sum = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) sum += r2[w4 * 2 * Cin + o];
            bn_part[(long long)blockIdx.x * 2 * Cin + o] = sum;
        }
    }
}

template <int CB, bool BNR>
int launch_head_bwd_mfma(int grid, hipStream_t st, const void* x, const float* dl, const float* w, long long M, const float* sc,
                         const float* sh, int act, void* dA, float* part, float* bn_part) {
    constexpr int Cin = 16 * CB;
    constexpr int sm = (BNR ? 3 : 2) * kHT * (Cin * 2 + 16) + 2 * kHT * (kHD * 2 + 16) + 2 * Cin * (int)sizeof(float);
    static_assert(sm >= 4 * (CB * 2 + 2) * 64 * 16, "head_bwd_mfma: the final reduction must fit in the tile images");
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)head_bwd_mfma_kernel<CB, BNR>, sm, &attr_mask));
    head_bwd_mfma_kernel<CB, BNR><<<grid, kThreads, sm, st>>>((const bf16_t*)x, dl, w, M, sc, sh, act, (bf16_t*)dA, part, bn_part);
    return MPN_OK;
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
    if (mode == 0 && dtype == MPN_BF16 && (Cin == 16 || Cin == 32 || Cin == 64)) {      // matrix-core kernel
        if (Cin == 16) head_fwd_mfma_kernel<1><<<grid, kThreads, 0, st>>>((const bf16_t*)x, w, bias, M, in_scale, in_shift, in_act, out);
        else if (Cin == 32) head_fwd_mfma_kernel<2><<<grid, kThreads, 0, st>>>((const bf16_t*)x, w, bias, M, in_scale, in_shift, in_act, out);
        else head_fwd_mfma_kernel<4><<<grid, kThreads, 0, st>>>((const bf16_t*)x, w, bias, M, in_scale, in_shift, in_act, out);
        MPN_LAUNCH_CHECK();
        return MPN_OK;
    }
    MPN_DISPATCH_DTYPE(dtype, (head_fwd_kernel<T><<<grid, kThreads, 0, st>>>((const T*)x, w, bias, M, Cin, in_scale, in_shift,
                                                                            in_act, mode, out, out_seg)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_heatmap_head_bwd_num_parts(long long M) {
    const long long ntiles = (M + kBwdPix - 1) / kBwdPix;
    const int cap = 512;   // two resident blocks per CU (57 KB of LDS each in the matrix-core kernel): one balanced round
    return (int)(ntiles < cap ? ntiles : cap);
}

namespace {
bool head_mfma_ok(int Cin, int dtype) { return dtype == MPN_BF16 && (Cin == 16 || Cin == 32 || Cin == 64); }
template <bool BNR>
int head_bwd_mfma_dispatch(int grid, hipStream_t st, const void* x, const float* dl, const float* w, long long M, int Cin,
                           const float* sc, const float* sh, int act, void* dA, float* part, float* bn_part) {
    switch (Cin / 16) {
        case 1: return launch_head_bwd_mfma<1, BNR>(grid, st, x, dl, w, M, sc, sh, act, dA, part, bn_part);
        case 2: return launch_head_bwd_mfma<2, BNR>(grid, st, x, dl, w, M, sc, sh, act, dA, part, bn_part);
        default: return launch_head_bwd_mfma<4, BNR>(grid, st, x, dl, w, M, sc, sh, act, dA, part, bn_part);
    }
}
}  // namespace

/* Can mpn_heatmap_head_bwd_bn run this geometry (the matrix-core kernel: bf16, Cin = 16, 32 or 64)? */
extern "C" int mpn_heatmap_head_bwd_bn_supported(int Cin, int dtype) { return head_mfma_ok(Cin, dtype) ? 1 : 0; }

/* mpn_heatmap_head_bwd + the reduction pass of the batch-norm the head reads its input through (final_bn,
 * keypoint_subnet.py:42-47): dA is the gradient w.r.t. that batch-norm's OUTPUT already masked by the activation
 * (lo < x * scale + shift < hi), and bn_part [mpn_heatmap_head_bwd_num_parts(M)][2][Cin] receives per-block sums of g and
 * g * x (raw x) - finish with mpn_bn_bwd_finalize_raw, then mpn_bn_bwd_apply. in_scale / in_shift are required. */
extern "C" int mpn_heatmap_head_bwd_bn(const void* x, const float* dlogits, const float* w, long long M, int Cin, int dtype,
                                       const float* in_scale, const float* in_shift, int in_act, void* dA, float* part,
                                       float* bn_part, mpn_stream_t stream) {
    if (int rc = check(M, Cin, dtype)) return rc;
    MPN_REQUIRE(head_mfma_ok(Cin, dtype), MPN_ERR_BAD_SHAPE, "head_bwd_bn: needs bf16 and Cin in {16, 32, 64} (got %d)", Cin);
    MPN_REQUIRE(x && dlogits && w && dA && part && bn_part && in_scale && in_shift, MPN_ERR_BAD_ARG, "head_bwd_bn: null pointer");
    if (int rc = head_bwd_mfma_dispatch<true>(mpn_heatmap_head_bwd_num_parts(M), (hipStream_t)stream, x, dlogits, w, M, Cin, in_scale,
                                              in_shift, in_act, dA, part, bn_part)) return rc;
    MPN_LAUNCH_CHECK();
    return MPN_OK;
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
    if (head_mfma_ok(Cin, dtype)) {
        if (int rc = head_bwd_mfma_dispatch<false>(grid, st, x, dlogits, w, M, Cin, in_scale, in_shift, in_act, dA, part, nullptr)) return rc;
        MPN_LAUNCH_CHECK();
        return MPN_OK;
    }
    MPN_DISPATCH_DTYPE(dtype, {
        if (sm > 48 * 1024)
            MPN_HIP(hipFuncSetAttribute((const void*)head_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
        head_bwd_kernel<T><<<grid, kThreads, sm, st>>>((const T*)x, dlogits, w, M, Cin, in_scale, in_shift, in_act, (T*)dA, part);
    });
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
