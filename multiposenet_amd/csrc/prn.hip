// PRN (pose residual network) helpers - SURVEY 8(f) rank 2, BASELINE config 5.
// Replaces detector/prn.py:5-25 (`prn`: flatten -> fc1 34272->1024 + ReLU -> fc2 1024->34272 + ReLU -> residual) and the
// loss of prn_model.py:16-30 (softmax over the h*w axis per keypoint channel + tf.losses.log_loss, mean).
// The four GEMMs run on the MFMA kernels that already exist (the two with K = 34272 as split-K "weight gradients" of a
// 1x1 convolution whose pixel axis is K, the other two as a 1x1 convolution / its weight gradient); this file holds what
// is left: transposes / casts that put an operand K-major, bias + ReLU forward / backward, and the loss with its gradient.
#include "common.h"

namespace {
constexpr int kThreads = 256;

template <typename T> __device__ __forceinline__ float ld_f32(const T* p, long long i) { return to_f32(p[i]); }

// out[c][r] = (TO) in[r][c]   (in: [R][C], out: [C][R]); 32x32 tiles through LDS, coalesced both ways
template <typename TI, typename TO>
__global__ __launch_bounds__(kThreads) void transpose_cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int r = r0 + ty + k, c = c0 + tx;
        tile[ty + k][tx] = (r < R && c < C) ? to_f32(in[(long long)r * C + c]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int c = c0 + ty + k, r = r0 + tx;
        if (c < C && r < R) out[(long long)c * R + r] = from_f32<TO>(tile[tx][ty + k]);
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(kThreads) void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads)
        out[i] = from_f32<TO>(to_f32(in[i]));
}

// y[r][c] = relu(pre[r][c] + bias[c])      (pre: f32 partial-reduced GEMM output or storage type)
template <typename TI, typename TO>
__global__ __launch_bounds__(kThreads) void bias_relu_fwd_kernel(const TI* __restrict__ pre, const float* __restrict__ bias,
                                                                TO* __restrict__ y, long long n, int C) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const float v = to_f32(pre[i]) + bias[i % C];
        y[i] = from_f32<TO>(fmaxf(v, 0.f));
    }
}

// the same, 8 elements (one or two 16-byte vectors) per thread; C % 8 == 0 so a vector never straddles a row
template <typename TI, typename TO>
__global__ __launch_bounds__(kThreads) void bias_relu_fwd_vec_kernel(const TI* __restrict__ pre, const float* __restrict__ bias,
                                                                    TO* __restrict__ y, long long nvec, int cvec) {
    for (long long v = (long long)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += (long long)gridDim.x * kThreads) {
        const int c0 = (int)(v % cvec) * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float f[8];
        if constexpr (sizeof(TI) == 4) {
            const float4 p0 = *reinterpret_cast<const float4*>(pre + v * 8), p1 = *reinterpret_cast<const float4*>(pre + v * 8 + 4);
            f[0] = p0.x; f[1] = p0.y; f[2] = p0.z; f[3] = p0.w; f[4] = p1.x; f[5] = p1.y; f[6] = p1.z; f[7] = p1.w;
        } else {
            Vec16<TI> q;
            q.load(pre + v * 8);
            q.unpack(f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] + bb[j], 0.f);
        if constexpr (sizeof(TO) == 4) {
            *reinterpret_cast<float4*>(y + v * 8) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4*>(y + v * 8 + 4) = make_float4(f[4], f[5], f[6], f[7]);
        } else {
            Vec16<TO> q;
            q.pack(f);
            q.store(y + v * 8);
        }
    }
}

// dpre[r][c] = y[r][c] > 0 ? dy[r][c] : 0;  dbias[c] = sum_r dpre[r][c]   (one thread per column, R rows: tiny R)
template <typename TY, typename TD>
__global__ __launch_bounds__(kThreads) void bias_relu_bwd_kernel(const TY* __restrict__ y, const float* __restrict__ dy,
                                                                TD* __restrict__ dpre, float* __restrict__ dbias, int R, int C) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) {
        const long long i = (long long)r * C + c;
        const float g = to_f32(y[i]) > 0.f ? dy[i] : 0.f;
        dpre[i] = from_f32<TD>(g);
        s += g;   // dbias from the unrounded gradient
    }
    dbias[c] = s;
}

// the same with 4 row groups per column quad: thread (quad, rg) walks rows rg, rg + 4, ... of columns 4 quad .. 4 quad + 3 (16-byte
// f32 loads, 4 rows in flight); the four row groups of a quad add their column sums through LDS in a fixed order. C % 4 == 0.
template <typename TY, typename TD>
__global__ __launch_bounds__(kThreads) void bias_relu_bwd_vec_kernel(const TY* __restrict__ y, const float* __restrict__ dy,
                                                                    TD* __restrict__ dpre, float* __restrict__ dbias, int R, int C) {
    __shared__ float red[4][64][4];
    const int ql = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + ql) * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        for (int r0 = rg; r0 < R; r0 += 16) {
            float4 g[4];
            float yv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + 4 * u < R ? r0 + 4 * u : rg;        // (unconditional loads from a valid row)
                const long long i = (long long)r * C + c;
                g[u] = *reinterpret_cast<const float4*>(dy + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) yv[u][j] = to_f32(y[i + j]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r0 + 4 * u >= R) break;
                const long long i = (long long)(r0 + 4 * u) * C + c;
                const float gg[4] = {yv[u][0] > 0.f ? g[u].x : 0.f, yv[u][1] > 0.f ? g[u].y : 0.f, yv[u][2] > 0.f ? g[u].z : 0.f,
                                     yv[u][3] > 0.f ? g[u].w : 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) { dpre[i + j] = from_f32<TD>(gg[j]); s[j] += gg[j]; }   // dbias from the unrounded gradient
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rg][ql][j] = s[j];
    __syncthreads();
    if (rg == 0 && c < C) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dbias[c + j] = (red[0][ql][j] + red[1][ql][j]) + (red[2][ql][j] + red[3][ql][j]);
    }
}

// logits[b][p][c] = x[b][p][c] + y2[b][p][c];  prob = softmax over p;  loss = mean(log_loss(labels, prob, eps=1e-7));
// dlogits = grad_scale * d loss / d logits (ginv = grad_scale / (B P C)).  One block per image: thread (c, lane) walks pixels lane, lane+LP, ...
constexpr int kPL = 15;   // pixel lanes per channel: 17 channels x 15 = 255 threads
template <typename TY>
__global__ __launch_bounds__(kThreads) void prn_loss_kernel(const float* __restrict__ x, const TY* __restrict__ y2,
                                                           const float* __restrict__ labels, int P, int C, float inv_total,
                                                           float ginv, float* __restrict__ logits, float* __restrict__ dlogits,
                                                           float* __restrict__ loss_part) {
    __shared__ float red[kThreads];
    __shared__ float stat[2][32];
    const int b = blockIdx.x;
    const int c = threadIdx.x / kPL, pl = threadIdx.x % kPL;
    const bool on = c < C;
    const long long base = (long long)b * P * C;
    auto reduce_c = [&](float v, bool is_max) -> float {   // over the kPL lanes of this channel
        red[threadIdx.x] = v;
        __syncthreads();
        float r = is_max ? -INFINITY : 0.f;
        if (on) {
            for (int k = 0; k < kPL; ++k) {
                const float t = red[c * kPL + k];
                r = is_max ? fmaxf(r, t) : r + t;
            }
        }
        __syncthreads();
        return r;
    };
    float m = -INFINITY;
    if (on)
        for (int p = pl; p < P; p += kPL) {
            const long long i = base + (long long)p * C + c;
            const float z = x[i] + to_f32(y2[i]);
            logits[i] = z;
            m = fmaxf(m, z);
        }
    m = reduce_c(m, true);
    float se = 0.f;
    if (on)
        for (int p = pl; p < P; p += kPL) se += expf(logits[base + (long long)p * C + c] - m);
    se = reduce_c(se, false);
    const float inv_se = on ? 1.f / se : 0.f;
    // loss and sum_j g_j p_j
    const float eps = 1e-7f;
    float l = 0.f, gp = 0.f;
    if (on)
        for (int p = pl; p < P; p += kPL) {
            const long long i = base + (long long)p * C + c;
            const float pr = expf(logits[i] - m) * inv_se;
            const float yv = labels[i];
            l += -yv * logf(pr + eps) - (1.f - yv) * logf(1.f - pr + eps);
            const float g = (-yv / (pr + eps) + (1.f - yv) / (1.f - pr + eps)) * ginv;
            gp += g * pr;
        }
    gp = reduce_c(gp, false);
    if (on && dlogits != nullptr)
        for (int p = pl; p < P; p += kPL) {
            const long long i = base + (long long)p * C + c;
            const float pr = expf(logits[i] - m) * inv_se;
            const float yv = labels[i];
            const float g = (-yv / (pr + eps) + (1.f - yv) / (1.f - pr + eps)) * ginv;
            dlogits[i] = pr * (g - gp);
        }
    // block loss partial (fixed order)
    red[threadIdx.x] = on ? l : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < kThreads; ++k) s += red[k];
        loss_part[b] = s * inv_total;
    }
    (void)stat;
}

// The same, coalesced: a block of 1024 threads per crop, of which LT = (1024 / C) * C are active (1020 for 17 channels) - thread t
// owns elements t, t + LT, t + 2 LT, ... of the crop's flattened [P][C] array, so consecutive threads read consecutive
// addresses and a thread's channel t % C never changes. The crop's logits (then numerators, then probabilities) live in LDS
// (56 x 36 x 17 floats = 137 KB: one block per CU): ONE read of x, y2 and labels from memory, one exp per element.
// Channel reductions: LDS, the LT / C threads of a channel in a fixed order.
constexpr int kLossThreads = 1024;
template <typename TY>
__global__ __launch_bounds__(kLossThreads) void prn_loss_wide_kernel(const float* __restrict__ x, const TY* __restrict__ y2,
                                                                    const float* __restrict__ labels, int P, int C, float inv_total,
                                                                    float ginv, float* __restrict__ logits, float* __restrict__ dlogits,
                                                                    float* __restrict__ loss_part) {
    extern __shared__ __attribute__((aligned(16))) float zs[];       // [P * C]
    __shared__ float red[kLossThreads];
    __shared__ float chan[32];
    const int t = threadIdx.x;
    const int per = kLossThreads / C, LT = per * C;      // threads per channel, active threads
    const bool on = t < LT;
    const int c = t % C;
    const long long base = (long long)blockIdx.x * P * C;
    const int n = P * C;
    auto reduce_c = [&](float v, bool is_max) -> float {
        red[t] = v;
        __syncthreads();
        if (t < C) {
            float r = is_max ? -INFINITY : 0.f;
            for (int k = 0; k < per; ++k) {
                const float q = red[k * C + t];
                r = is_max ? fmaxf(r, q) : r + q;
            }
            chan[t] = r;
        }
        __syncthreads();
        const float r = chan[c];
        __syncthreads();
        return r;
    };
    float m = -INFINITY;
    if (on)
        for (int i = t; i < n; i += LT) {
            const float z = x[base + i] + to_f32(y2[base + i]);
            logits[base + i] = z;
            zs[i] = z;
            m = fmaxf(m, z);
        }
    m = reduce_c(m, true);
    float se = 0.f;
    if (on)
        for (int i = t; i < n; i += LT) {
            const float e = expf(zs[i] - m);
            zs[i] = e;
            se += e;
        }
    se = reduce_c(se, false);
    const float inv_se = 1.f / se;
    const float eps = 1e-7f;
    float l = 0.f, gp = 0.f;
    if (on)
        for (int i = t; i < n; i += LT) {
            const float pr = zs[i] * inv_se;
            const float yv = labels[base + i];
            l += -yv * logf(pr + eps) - (1.f - yv) * logf(1.f - pr + eps);
            const float g = (-yv / (pr + eps) + (1.f - yv) / (1.f - pr + eps)) * ginv;
            gp += g * pr;
            zs[i] = pr;
        }
    gp = reduce_c(gp, false);
    if (on && dlogits != nullptr)
        for (int i = t; i < n; i += LT) {        // (the labels once more, out of the L2)
            const float pr = zs[i];
            const float yv = labels[base + i];
            const float g = (-yv / (pr + eps) + (1.f - yv) / (1.f - pr + eps)) * ginv;
            dlogits[base + i] = pr * (g - gp);
        }
    red[t] = on ? l : 0.f;
    __syncthreads();
    if (t < 64) {   // fixed order: 16 strided terms per lane, then the butterfly
        float s = 0.f;
        for (int k = t; k < kLossThreads; k += 64) s += red[k];
        s = wave_sum(s);
        if (t == 0) loss_part[blockIdx.x] = s * inv_total;
    }
}

int blocks_for(long long n) {
    long long b = (n + kThreads - 1) / kThreads;
    if (b > 8192) b = 8192;
    return (int)(b < 1 ? 1 : b);
}
}  // namespace

/* storage types of the PRN entry points: MPN_F32, MPN_BF16, MPN_F16 (config 5 of BASELINE.json runs the PRN in fp16) */
#define PRN_DISPATCH_ONE(dtype, NAME, ...)                               \
    do {                                                                 \
        if ((dtype) == MPN_F32) { using NAME = float; __VA_ARGS__; }     \
        else if ((dtype) == MPN_BF16) { using NAME = bf16_t; __VA_ARGS__; } \
        else if ((dtype) == MPN_F16) { using NAME = half_t; __VA_ARGS__; }  \
        else MPN_FAIL(MPN_ERR_BAD_DTYPE, "prn: unsupported dtype %d", (int)(dtype)); \
    } while (0)
#define PRN_DISPATCH_TWO(dt_in, dt_out, ...) PRN_DISPATCH_ONE(dt_in, TI, PRN_DISPATCH_ONE(dt_out, TO, __VA_ARGS__))

// ---------------------------------------------------------------- skinny "NT" GEMM, split over K
// C[M][N] = A[M][K] * B[N][K]^T with both operands stored K-contiguous (16-bit): exactly the MFMA's fragment order, so a lane
// loads its 8 consecutive k of a row straight from HBM - no LDS, no transposed copy of the big operand. The PRN's fc2 data
// gradient dH[B,1024] = dPre2[B,n] * W2[1024,n]^T took a transposed 16-bit copy of W2 per step (69-90 us for 35 M elements)
// plus a transposed dPre2; here W2 is read as stored (the plain cast the Adam kernel writes).
// Block = 128 x 128 outputs over one K range (4 waves of 64 x 64: 4 + 4 fragment loads per 16 MFMAs and 32-deep step, next
// step's fragments in flight); partials [split][M][N] f32, finished by mpn_reduce_partials (fixed order).
namespace {
constexpr int kNtK = 512;       // K per split (a multiple of 64): two blocks per CU at the PRN's size (1024: one, 54 us; 512: see DESIGN)
template <typename T>
__global__ __launch_bounds__(kThreads) void gemm_nt_kernel(const T* __restrict__ a, const T* __restrict__ b, float* __restrict__ part,
                                                           int M, int N, int K, int n_tiles, int m_tiles) {
    typedef H16<T> HT;
    typedef typename HT::x8 x8;
    typedef typename HT::acc_t acc_t;
    int bid = blockIdx.x;
    const int nt = bid % n_tiles; bid /= n_tiles;
    const int mt = bid % m_tiles;
    const int split = bid / m_tiles;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int wm = wv & 1, wn = wv >> 1;
    const int k_begin = split * kNtK, k_end = min(K, k_begin + kNtK);
    // rows of this lane's fragments (clamped: rows past the edge are loaded from row 0 and never stored)
    const T* ap[4];
    const T* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mt * 128 + wm * 64 + i * 16 + l15, n = nt * 128 + wn * 64 + i * 16 + l15;
        ap[i] = a + (long long)(m < M ? m : 0) * K + 8 * kg;
        bp[i] = b + (long long)(n < N ? n : 0) * K + 8 * kg;
    }
    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (acc_t){0.f, 0.f, 0.f, 0.f};
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    auto ld = [&](const T* p, int k) __attribute__((always_inline)) -> x8 {
        // (K % 8 == 0: a lane's 8 elements are all inside or all outside)
        const u32x4_t q = (k + 8 * kg < k_end) ? *reinterpret_cast<const u32x4_t*>(p + k) : (u32x4_t){0u, 0u, 0u, 0u};
        return __builtin_bit_cast(x8, q);
    };
    x8 fa[4], fb[4], ga[4], gb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { fa[i] = ld(ap[i], k_begin); fb[i] = ld(bp[i], k_begin); }
    for (int k = k_begin; k < k_end; k += 64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ga[i] = ld(ap[i], k + 32); gb[i] = ld(bp[i], k + 32); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = HT::mfma(fb[j], fa[i], acc[i][j]);   // D[n][m]: a lane ends up with 4 consecutive n of one m
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[i] = ld(ap[i], k + 64); fb[i] = ld(bp[i], k + 64); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = HT::mfma(gb[j], ga[i], acc[i][j]);
    }
    float* dst = part + (long long)split * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mt * 128 + wm * 64 + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = nt * 128 + wn * 64 + j * 16 + 4 * kg;
            if (m < M && n0 < N)   // (N % 4 == 0)
                *reinterpret_cast<float4*>(dst + (long long)m * N + n0) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}
}  // namespace

/* K ranges (= rows of the partial slab) of mpn_gemm_nt */
extern "C" int mpn_gemm_nt_num_parts(int K) { return K > 0 ? (K + kNtK - 1) / kNtK : 0; }

/* part[s][M][N] (f32, s < mpn_gemm_nt_num_parts(K)) = A[M][K] * B[N][K]^T over the s-th K range; A, B 16-bit (MPN_BF16 / MPN_F16),
 * row-major with K contiguous, K % 8 == 0, N % 4 == 0; finish with mpn_reduce_partials(part, parts, M * N, out). */
extern "C" int mpn_gemm_nt(const void* a, const void* b, float* part, int M, int N, int K, int dtype, mpn_stream_t stream) {
    MPN_REQUIRE(a && b && part, MPN_ERR_BAD_ARG, "gemm_nt: null pointer");
    MPN_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 4 == 0, MPN_ERR_BAD_SHAPE, "gemm_nt: needs K %% 8 == 0 and N %% 4 == 0 (M %d, N %d, K %d)", M, N, K);
    MPN_REQUIRE(dtype == MPN_BF16 || dtype == MPN_F16, MPN_ERR_BAD_DTYPE, "gemm_nt: 16-bit operands only (dtype %d)", dtype);
    MPN_REQUIRE(mpn_aligned16(a) && mpn_aligned16(b) && mpn_aligned16(part), MPN_ERR_BAD_ALIGN, "gemm_nt: pointers must be 16-byte aligned");
    const int n_tiles = (N + 127) / 128, m_tiles = (M + 127) / 128, splits = mpn_gemm_nt_num_parts(K);
    const long long blocks = (long long)n_tiles * m_tiles * splits;
    MPN_REQUIRE(blocks < (1ll << 31), MPN_ERR_BAD_SHAPE, "gemm_nt: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPN_BF16)
        gemm_nt_kernel<bf16_t><<<(unsigned)blocks, kThreads, 0, st>>>((const bf16_t*)a, (const bf16_t*)b, part, M, N, K, n_tiles, m_tiles);
    else
        gemm_nt_kernel<half_t><<<(unsigned)blocks, kThreads, 0, st>>>((const half_t*)a, (const half_t*)b, part, M, N, K, n_tiles, m_tiles);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* out[C][R] = cast(in[R][C]); in_dtype / out_dtype: MPN_F32, MPN_BF16 or MPN_F16 */
extern "C" int mpn_transpose_cast(const void* in, int in_dtype, void* out, int out_dtype, int R, int C, mpn_stream_t stream) {
    MPN_REQUIRE(in && out && R > 0 && C > 0, MPN_ERR_BAD_ARG, "transpose_cast: bad arguments");
    const dim3 grid((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32));
    hipStream_t st = (hipStream_t)stream;
    PRN_DISPATCH_TWO(in_dtype, out_dtype, (transpose_cast_kernel<TI, TO><<<grid, kThreads, 0, st>>>((const TI*)in, (TO*)out, R, C)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* out[i] = cast(in[i]) */
extern "C" int mpn_cast(const void* in, int in_dtype, void* out, int out_dtype, long long n, mpn_stream_t stream) {
    MPN_REQUIRE(in && out && n > 0, MPN_ERR_BAD_ARG, "cast: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int g = blocks_for(n);
    PRN_DISPATCH_TWO(in_dtype, out_dtype, (cast_kernel<TI, TO><<<g, kThreads, 0, st>>>((const TI*)in, (TO*)out, n)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* y[R][C] = relu(pre + bias[c]);  pre: f32 (pre_dtype MPN_F32) or storage type, y: out_dtype */
extern "C" int mpn_bias_relu_fwd(const void* pre, int pre_dtype, const float* bias, void* y, int out_dtype, int R, int C,
                                 mpn_stream_t stream) {
    MPN_REQUIRE(pre && bias && y && R > 0 && C > 0, MPN_ERR_BAD_ARG, "bias_relu_fwd: bad arguments");
    const long long n = (long long)R * C;
    hipStream_t st = (hipStream_t)stream;
    const int g = blocks_for(n);
    if (C % 8 == 0 && mpn_aligned16(pre) && mpn_aligned16(y) && mpn_aligned16(bias)) {
        const int gv = blocks_for(n / 8);
        PRN_DISPATCH_TWO(pre_dtype, out_dtype, (bias_relu_fwd_vec_kernel<TI, TO><<<gv, kThreads, 0, st>>>((const TI*)pre, bias, (TO*)y, n / 8, C / 8)));
    } else {
        PRN_DISPATCH_TWO(pre_dtype, out_dtype, (bias_relu_fwd_kernel<TI, TO><<<g, kThreads, 0, st>>>((const TI*)pre, bias, (TO*)y, n, C)));
    }
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* dpre = (y > 0) * dy (storage dtype of the GEMM operand), dbias[c] = column sums; dy f32; y and dpre share one dtype */
extern "C" int mpn_bias_relu_bwd(const void* y, int y_dtype, const float* dy, void* dpre, int dpre_dtype, float* dbias, int R,
                                 int C, mpn_stream_t stream) {
    MPN_REQUIRE(y && dy && dpre && dbias && R > 0 && C > 0, MPN_ERR_BAD_ARG, "bias_relu_bwd: bad arguments");
    MPN_REQUIRE(y_dtype == dpre_dtype, MPN_ERR_BAD_DTYPE, "bias_relu_bwd: y and dpre must share one dtype");
    hipStream_t st = (hipStream_t)stream;
    const int g = (C + kThreads - 1) / kThreads;
    if (C % 4 == 0 && mpn_aligned16(dy)) {
        const int gv = (C / 4 + 63) / 64;
        PRN_DISPATCH_ONE(y_dtype, TS, (bias_relu_bwd_vec_kernel<TS, TS><<<gv, kThreads, 0, st>>>((const TS*)y, dy, (TS*)dpre, dbias, R, C)));
    } else {
        PRN_DISPATCH_ONE(y_dtype, TS, (bias_relu_bwd_kernel<TS, TS><<<g, kThreads, 0, st>>>((const TS*)y, dy, (TS*)dpre, dbias, R, C)));
    }
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

template <typename TS>
__global__ void prn_residual_kernel(const float* __restrict__ x, const TS* __restrict__ y2, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = x[i] + to_f32(y2[i]);
}

/* logits = x + y2 (detector/prn.py:24: the residual connection at inference; the loss kernel forms it itself) */
extern "C" int mpn_prn_residual(const float* x, const void* y2, int y2_dtype, long long n, float* logits, mpn_stream_t stream) {
    MPN_REQUIRE(x && y2 && logits && n > 0, MPN_ERR_BAD_ARG, "prn_residual: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int g = blocks_for(n);
    PRN_DISPATCH_ONE(y2_dtype, TS, (prn_residual_kernel<TS><<<g, kThreads, 0, st>>>(x, (const TS*)y2, logits, n)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* PRN loss (prn_model.py:16-30): logits = x + y2 [B][P][C]; softmax over P; mean log_loss; C <= 17.
 * logits, dlogits (may be NULL): f32 [B][P][C]; loss_part: f32 [B] (sum = loss).
 * grad_scale: dlogits = grad_scale * dloss/dlogits - the static loss scale of the fp16 build (the unscaled gradient of a
 * mean over B*P*C ~ 4e6 terms is below fp16's normal range once stored as the GEMM operand dpre2); 1 otherwise. The
 * caller hands 1 / grad_scale to mpn_adam_step. */
extern "C" int mpn_prn_loss(const float* x, const void* y2, int y2_dtype, const float* labels, int B, int P, int C,
                            float* logits, float* dlogits, float* loss_part, float grad_scale, mpn_stream_t stream) {
    MPN_REQUIRE(x && y2 && labels && logits && loss_part, MPN_ERR_BAD_ARG, "prn_loss: null pointer");
    MPN_REQUIRE(grad_scale > 0.f, MPN_ERR_BAD_ARG, "prn_loss: grad_scale must be positive");
    MPN_REQUIRE(B > 0 && P > 0 && C > 0 && C * kPL <= kThreads, MPN_ERR_BAD_SHAPE, "prn_loss: C must be <= %d", kThreads / kPL);
    const float inv_total = 1.0f / ((float)B * (float)P * (float)C);
    hipStream_t st = (hipStream_t)stream;
    const size_t zbytes = (size_t)P * C * sizeof(float);
    if (zbytes <= 150 * 1024) {   // the crop's logits fit the LDS of one CU (56 x 36 x 17 floats = 137 KB do)
        PRN_DISPATCH_ONE(y2_dtype, TS, {
            static mpn_attr_mask_t attr_mask{0};
            MPN_HIP(mpn_ensure_dynamic_lds((const void*)prn_loss_wide_kernel<TS>, 150 * 1024, &attr_mask));
            prn_loss_wide_kernel<TS><<<B, kLossThreads, zbytes, st>>>(x, (const TS*)y2, labels, P, C, inv_total, inv_total * grad_scale, logits, dlogits, loss_part);
        });
    } else {
        PRN_DISPATCH_ONE(y2_dtype, TS, (prn_loss_kernel<TS><<<B, kThreads, 0, st>>>(x, (const TS*)y2, labels, P, C, inv_total, inv_total * grad_scale, logits, dlogits, loss_part)));
    }
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
