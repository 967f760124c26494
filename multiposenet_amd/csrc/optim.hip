// K13: optimizer step of train_keypoints.py / keypoints_model.py:107-120 as ONE fused pass over a flat
// parameter arena: tf.clip_by_value(g, -200, 200), TF-1.15 AdamOptimizer
//   lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   theta -= lr_t * m / (sqrt(v) + eps)      (eps OUTSIDE the bias correction, unlike torch.optim.Adam)
// with lr from tf.train.cosine_decay(alpha=1e-4). Step count and learning rate live in device memory so the
// whole training step can be replayed from a hipGraph. Also: deterministic reduction of partial slabs, axpy.
#include "common.h"
#include <string.h>

namespace {
constexpr int kThreads = 256;

__global__ void adam_prepare_kernel(long long* __restrict__ step, float* __restrict__ hyper, double lr0,
                                    double decay_steps, double alpha, double beta1, double beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long gs = *step;
    const double s = (double)gs < decay_steps ? (double)gs : decay_steps;
    const double cosine = 0.5 * (1.0 + cos(3.14159265358979323846 * s / decay_steps));
    const double lr = lr0 * ((1.0 - alpha) * cosine + alpha);
    const double t = (double)(gs + 1);
    const double lr_t = lr * sqrt(1.0 - pow(beta2, t)) / (1.0 - pow(beta1, t));
    hyper[0] = (float)lr_t;
    hyper[1] = (float)lr;
    *step = gs + 1;
}

// A block takes U consecutive 256-vector pieces per iteration (contiguous 8 KB per stream) and issues all its loads first:
// 52 M floats cold 282 -> 259 us against one vector per thread at grid stride (tools/bench_adam.py).
// CAST: the updated values of up to kAdamCast ranges of the arena also leave as 16-bit copies (dst[i - begin] = (T) theta[i]) -
// the GEMM operand of a dense layer in the storage dtype without a second pass over its f32 master (the pose residual
// network refreshed its two 35 M-element operands with a 61 us cast pass and a 145 us pack pass per step)
constexpr int kAdamCast = 4;
struct AdamCast { long long begin4[kAdamCast], end4[kAdamCast]; void* dst[kAdamCast]; int count, dtype; };
template <typename T> __device__ __forceinline__ void adam_cast_store(void* dst, long long i4, const float4& q) {
    uint2 o;
    if constexpr (sizeof(T) == 2 && !__is_same(T, half_t)) {      // bf16: one convert per pair (common.h)
        o.x = pack_bf16x2(q.x, q.y);
        o.y = pack_bf16x2(q.z, q.w);
    } else {
        const T a = (T)q.x, b = (T)q.y, c = (T)q.z, d = (T)q.w;
        o.x = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
        o.y = (unsigned)__builtin_bit_cast(unsigned short, c) | ((unsigned)__builtin_bit_cast(unsigned short, d) << 16);
    }
    reinterpret_cast<uint2*>(dst)[i4] = o;
}
template <int U, bool CAST>
__global__ __launch_bounds__(kThreads) void adam_apply_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                      float* __restrict__ m, float* __restrict__ v, long long n4,
                                                                      const float* __restrict__ hyper, float beta1, float beta2,
                                                                      float eps, float clip, float grad_scale, const AdamCast cj) {
    const float lr_t = hyper[0];
    for (long long i0 = (long long)blockIdx.x * (U * kThreads) + threadIdx.x; i0 < n4; i0 += (long long)gridDim.x * (U * kThreads)) {
        float4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * kThreads;
            const long long ic = i < n4 ? i : i0;
            pp[u] = reinterpret_cast<const float4*>(p)[ic];
            gg[u] = reinterpret_cast<const float4*>(g)[ic];
            mm[u] = reinterpret_cast<const float4*>(m)[ic];
            vv[u] = reinterpret_cast<const float4*>(v)[ic];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * kThreads;
            float* pa = reinterpret_cast<float*>(&pp[u]);
            const float* ga = reinterpret_cast<const float*>(&gg[u]);
            float* ma = reinterpret_cast<float*>(&mm[u]);
            float* va = reinterpret_cast<float*>(&vv[u]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float gj = ga[j] * grad_scale;
                gj = fminf(fmaxf(gj, -clip), clip);
                ma[j] = beta1 * ma[j] + (1.0f - beta1) * gj;
                va[j] = beta2 * va[j] + (1.0f - beta2) * gj * gj;
                pa[j] -= lr_t * ma[j] / (sqrtf(va[j]) + eps);
            }
            if (i < n4) {
                reinterpret_cast<float4*>(p)[i] = pp[u];
                reinterpret_cast<float4*>(m)[i] = mm[u];
                reinterpret_cast<float4*>(v)[i] = vv[u];
                if (CAST) {
#pragma unroll
                    for (int r = 0; r < kAdamCast; ++r)
                        if (r < cj.count && i >= cj.begin4[r] && i < cj.end4[r]) {
                            if (cj.dtype == MPN_F16) adam_cast_store<half_t>(cj.dst[r], i - cj.begin4[r], pp[u]);
                            else adam_cast_store<bf16_t>(cj.dst[r], i - cj.begin4[r], pp[u]);
                        }
                }
            }
        }
    }
}

// out[j] (+)= scale * sum_p part[p][j], fixed order. Bandwidth-bound (up to 75 MB of split-K slabs per layer).
// One pass has only n/1024 blocks (e.g. 144 for a 3x3 128->128 kernel), so big slabs are reduced in two passes:
// pass 1 cuts the parts into `nslices` ranges (grid.y) and writes each range's sum IN PLACE over the first row of
// its own range (only that thread reads it); pass 2 combines those nslices rows. Fixed order, no atomics.
// MODE 0: plain rows p = 0..nparts-1 -> out.  MODE 1 (pass 1): range sums -> part row p0(slice).
// MODE 2 (pass 2): rows p0(0..nslices-1) of the original slab -> out.
template <int MODE>
__global__ __launch_bounds__(kThreads) void reduce_partials_kernel(float* __restrict__ part, int nparts, long long n,
                                                                   float* __restrict__ out, int accumulate, float scale,
                                                                   int nslices) {
    const long long n4 = n >> 2;
    int p0 = 0, p1 = nparts;
    if (MODE == 1) {
        p0 = (int)((long long)nparts * blockIdx.y / nslices);
        p1 = (int)((long long)nparts * (blockIdx.y + 1) / nslices);
    } else if (MODE == 2) {
        p1 = nslices;
    }
    auto row = [&](int p) -> const float4* {
        const long long r = (MODE == 2) ? ((long long)nparts * p / nslices) : p;
        return reinterpret_cast<const float4*>(part + r * n);
    };
    for (long long j = (long long)blockIdx.x * kThreads + threadIdx.x; j < n4; j += (long long)gridDim.x * kThreads) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a, d = a;
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row(p + u)[j];
#pragma unroll
            for (int u = 0; u < 8; u += 4) {
                a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
                b.x += v[u + 1].x; b.y += v[u + 1].y; b.z += v[u + 1].z; b.w += v[u + 1].w;
                c.x += v[u + 2].x; c.y += v[u + 2].y; c.z += v[u + 2].z; c.w += v[u + 2].w;
                d.x += v[u + 3].x; d.y += v[u + 3].y; d.z += v[u + 3].z; d.w += v[u + 3].w;
            }
        }
        for (; p < p1; ++p) {
            const float4 v = row(p)[j];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        float4 r;
        r.x = (a.x + b.x) + (c.x + d.x); r.y = (a.y + b.y) + (c.y + d.y);
        r.z = (a.z + b.z) + (c.z + d.z); r.w = (a.w + b.w) + (c.w + d.w);
        if (MODE == 1) {
            reinterpret_cast<float4*>(part + (long long)p0 * n)[j] = r;
        } else {
            r.x *= scale; r.y *= scale; r.z *= scale; r.w *= scale;
            if (accumulate) {
                const float4 o = reinterpret_cast<const float4*>(out)[j];
                r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
            }
            reinterpret_cast<float4*>(out)[j] = r;
        }
    }
}

__global__ __launch_bounds__(kThreads) void reduce_partials_scalar_kernel(const float* __restrict__ part, int nparts,
                                                                          long long n, float* __restrict__ out,
                                                                          int accumulate, float scale) {
    const long long j = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (j >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 4 <= nparts; p += 4) {
        s0 += part[(long long)p * n + j];
        s1 += part[(long long)(p + 1) * n + j];
        s2 += part[(long long)(p + 2) * n + j];
        s3 += part[(long long)(p + 3) * n + j];
    }
    for (; p < nparts; ++p) s0 += part[(long long)p * n + j];
    const float s = ((s0 + s1) + (s2 + s3)) * scale;
    out[j] = accumulate ? out[j] + s : s;
}

// ---- all slab reductions of a training step in ONE launch (46 slabs -> 87 launches of ~5 us each otherwise).
// A block owns `cols` consecutive columns (float4 or float) of one slab and splits the slab's rows over 256/cols row
// slices; slices are combined through LDS in a fixed order, so the result is deterministic.
struct ReduceDesc {
    const float* part;
    float* out;
    long long n;          // floats per row
    int nparts;
    int cols;             // power of two <= 256
    int vec;              // 1: columns are float4 (n % 4 == 0), 0: floats
    int block_begin, block_count;
    float scale;
    int pad_;
};

__global__ __launch_bounds__(kThreads) void reduce_partials_batched_kernel(const ReduceDesc* __restrict__ descs, int ndesc) {
    __shared__ float4 red[kThreads];
    __shared__ int job;
    // the job that owns this block: ONE parallel read of the block_begin column + a ballot (a binary search is ~6 dependent
    // L2 round trips per block, more than the work of a block that sums 4 rows)
    if (threadIdx.x < 64) {
        int cnt = 0;
        for (int base = 0; base < ndesc; base += 64) {
            const int i = base + (int)threadIdx.x;
            const bool le = i < ndesc && descs[i].block_begin <= (int)blockIdx.x;
            cnt += __popcll(__ballot(le));
        }
        if (threadIdx.x == 0) job = cnt - 1;
    }
    __syncthreads();
    const ReduceDesc d = descs[job];
    const int lb = blockIdx.x - d.block_begin;
    const int c = threadIdx.x & (d.cols - 1), sl = threadIdx.x / d.cols, nsl = kThreads / d.cols;
    const long long ncol = d.vec ? (d.n >> 2) : d.n;
    const long long j = (long long)lb * d.cols + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (j < ncol) {
        if (d.vec) {
            int p = sl;
            for (; p + 3 * nsl < d.nparts; p += 4 * nsl) {
                // (slabs are dead after this read: non-temporal)
                typedef float f4_t __attribute__((ext_vector_type(4)));
                const f4_t v0 = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(d.part + (long long)p * d.n) + j);
                const f4_t v1 = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(d.part + (long long)(p + nsl) * d.n) + j);
                const f4_t v2 = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(d.part + (long long)(p + 2 * nsl) * d.n) + j);
                const f4_t v3 = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(d.part + (long long)(p + 3 * nsl) * d.n) + j);
                a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
                b.x += v1.x; b.y += v1.y; b.z += v1.z; b.w += v1.w;
                a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
                b.x += v3.x; b.y += v3.y; b.z += v3.z; b.w += v3.w;
            }
            for (; p < d.nparts; p += nsl) {
                const float4 v = reinterpret_cast<const float4*>(d.part + (long long)p * d.n)[j];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        } else {
            for (int p = sl; p < d.nparts; p += nsl) a.x += d.part[(long long)p * d.n + j];
        }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (sl == 0 && j < ncol) {
        float4 r = red[c];
        for (int q = 1; q < nsl; ++q) {
            const float4 v = red[q * d.cols + c];
            r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
        }
        if (d.vec) {
            r.x *= d.scale; r.y *= d.scale; r.z *= d.scale; r.w *= d.scale;
            reinterpret_cast<float4*>(d.out)[j] = r;
        } else {
            d.out[j] = r.x * d.scale;
        }
    }
}

__global__ __launch_bounds__(kThreads) void axpy_kernel(long long n, float a, const float* __restrict__ x, float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads)
        y[i] += a * x[i];
}

// acc[0] += scale * sum(w^2)/2 : ONE block, every thread a fixed strided share in f64, fixed-order tree -> the same bits
// on every run (tf.nn.l2_loss = sum(t^2)/2; keypoints_model.py:137). Regularisation is off in the keypoint run, so this
// is a correctness path, not a hot one.
__global__ __launch_bounds__(1024) void l2_loss_kernel(long long n, const float* __restrict__ w, double scale,
                                                        float* __restrict__ acc) {
    __shared__ double red[1024];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const double v = (double)w[i];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[0] = (float)((double)acc[0] + scale * 0.5 * red[0]);
}
// The regularisation term of a whole model in two launches (add_weight_decay sums tf.nn.l2_loss over every kernel,
// keypoints_model.py:129-138; the person detector trains with weight_decay 5e-5, train_person_detector.py): one single-block
// launch per variable took 2.4 ms of the detector's 9 ms step (a 4 MB pointwise kernel through ONE block: 476 us).
constexpr int kL2Max = 96;             // tensors per launch
constexpr int kL2Chunk = 16384;        // elements per block
struct L2Batch { const float* w[kL2Max]; long long n[kL2Max]; int begin[kL2Max + 1]; int count; };
__global__ __launch_bounds__(256) void l2_partial_kernel(const L2Batch b, double* __restrict__ partial) {
    __shared__ double red[256];
    int t = 0;
    for (int k = 1; k < b.count; ++k)
        if ((int)blockIdx.x >= b.begin[k]) t = k;                       // block-uniform
    const long long i0 = (long long)((int)blockIdx.x - b.begin[t]) * kL2Chunk;
    const long long i1 = i0 + kL2Chunk < b.n[t] ? i0 + kL2Chunk : b.n[t];
    const float* __restrict__ w = b.w[t];
    double s = 0.0;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const double v = (double)w[i];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void l2_final_kernel(const double* __restrict__ partial, int n, double scale, float* __restrict__ acc) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];         // fixed order: deterministic
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[0] = (float)((double)acc[0] + scale * 0.5 * red[0]);
}
// y[t] += a * x[t] for up to kAxpyMax tensors in one grid (the weight-decay gradients of the detector head were 18 launches
// of ~5 us each); blocks of kAxpyChunk elements, job table in the kernel arguments.
constexpr int kAxpyMax = 64;
constexpr int kAxpyChunk = 4096;
struct AxpyBatch { const float* x[kAxpyMax]; float* y[kAxpyMax]; long long n[kAxpyMax]; int begin[kAxpyMax + 1]; int count; };
__global__ __launch_bounds__(256) void axpy_batched_kernel(const AxpyBatch b, float a) {
    int t = 0;
    for (int k = 1; k < b.count; ++k)
        if ((int)blockIdx.x >= b.begin[k]) t = k;                       // block-uniform
    const long long i0 = (long long)((int)blockIdx.x - b.begin[t]) * kAxpyChunk;
    const long long i1 = i0 + kAxpyChunk < b.n[t] ? i0 + kAxpyChunk : b.n[t];
    const float* __restrict__ x = b.x[t];
    float* __restrict__ y = b.y[t];
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) y[i] += a * x[i];
}
}  // namespace

/* step: device int64 global_step (incremented); hyper: device f32[4] -> {lr_t, lr, -, -} */
extern "C" int mpn_adam_prepare(long long* step, float* hyper, double initial_learning_rate, double decay_steps, double alpha,
                                double beta1, double beta2, mpn_stream_t stream) {
    MPN_REQUIRE(step && hyper, MPN_ERR_BAD_ARG, "adam_prepare: null pointer");
    MPN_REQUIRE(decay_steps > 0, MPN_ERR_BAD_ARG, "adam_prepare: decay_steps must be positive");
    adam_prepare_kernel<<<1, 1, 0, (hipStream_t)stream>>>(step, hyper, initial_learning_rate, decay_steps, alpha, beta1, beta2);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* n must be a multiple of 4 (the arena pads each tensor); all pointers 16-byte aligned */
extern "C" int mpn_adam_step(float* params, const float* grads, float* m, float* v, long long n, const float* hyper,
                             float beta1, float beta2, float eps, float clip, float grad_scale, mpn_stream_t stream) {
    MPN_REQUIRE(params && grads && m && v && hyper, MPN_ERR_BAD_ARG, "adam_step: null pointer");
    MPN_REQUIRE(n > 0 && n % 4 == 0, MPN_ERR_BAD_SHAPE, "adam_step: n must be a positive multiple of 4");
    MPN_REQUIRE(mpn_aligned16(params) && mpn_aligned16(grads) && mpn_aligned16(m) && mpn_aligned16(v), MPN_ERR_BAD_ALIGN,
                "adam_step: arenas must be 16-byte aligned");
    const long long n4 = n / 4;
    constexpr int U = 2;
    long long blocks = (n4 + U * kThreads - 1) / (U * kThreads);
    if (blocks > 4096) blocks = 4096;
    adam_apply_kernel<U, false><<<(int)blocks, kThreads, 0, (hipStream_t)stream>>>(params, grads, m, v, n4, hyper, beta1, beta2, eps,
                                                                                  clip, grad_scale, AdamCast{});
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* mpn_adam_step that also writes 16-bit copies of `ncast` (<= 4) ranges of the updated arena: dst[r][i] = (T) params[begin[r] + i]
 * for i < count[r], T = fp16 / bf16 by cast_dtype. begin / count multiples of 4 (the arena's tensors start on 16 bytes),
 * dst 8-byte aligned. */
extern "C" int mpn_adam_step_cast(float* params, const float* grads, float* m, float* v, long long n, const float* hyper,
                                  float beta1, float beta2, float eps, float clip, float grad_scale, int ncast,
                                  const long long* cast_begin, const long long* cast_count, void* const* cast_dst, int cast_dtype,
                                  mpn_stream_t stream) {
    MPN_REQUIRE(params && grads && m && v && hyper, MPN_ERR_BAD_ARG, "adam_step_cast: null pointer");
    MPN_REQUIRE(n > 0 && n % 4 == 0, MPN_ERR_BAD_SHAPE, "adam_step_cast: n must be a positive multiple of 4");
    MPN_REQUIRE(mpn_aligned16(params) && mpn_aligned16(grads) && mpn_aligned16(m) && mpn_aligned16(v), MPN_ERR_BAD_ALIGN,
                "adam_step_cast: arenas must be 16-byte aligned");
    MPN_REQUIRE(ncast >= 1 && ncast <= kAdamCast && cast_begin && cast_count && cast_dst, MPN_ERR_BAD_ARG, "adam_step_cast: 1..4 ranges");
    MPN_REQUIRE(cast_dtype == MPN_F16 || cast_dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "adam_step_cast: 16-bit copies only (dtype %d)", cast_dtype);
    AdamCast cj = {};
    cj.count = ncast; cj.dtype = cast_dtype;
    for (int r = 0; r < ncast; ++r) {
        MPN_REQUIRE(cast_begin[r] >= 0 && cast_count[r] > 0 && cast_begin[r] % 4 == 0 && cast_count[r] % 4 == 0 &&
                        cast_begin[r] + cast_count[r] <= n && cast_dst[r] && (((uintptr_t)cast_dst[r]) & 7u) == 0,
                    MPN_ERR_BAD_ARG, "adam_step_cast: range %d", r);
        cj.begin4[r] = cast_begin[r] / 4; cj.end4[r] = (cast_begin[r] + cast_count[r]) / 4; cj.dst[r] = cast_dst[r];
    }
    const long long n4 = n / 4;
    constexpr int U = 2;
    long long blocks = (n4 + U * kThreads - 1) / (U * kThreads);
    if (blocks > 4096) blocks = 4096;
    adam_apply_kernel<U, true><<<(int)blocks, kThreads, 0, (hipStream_t)stream>>>(params, grads, m, v, n4, hyper, beta1, beta2, eps,
                                                                                 clip, grad_scale, cj);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_reduce_partials(const float* part, int nparts, long long n, float* out, int accumulate, float scale,
                                   mpn_stream_t stream) {
    MPN_REQUIRE(part && out, MPN_ERR_BAD_ARG, "reduce_partials: null pointer");
    MPN_REQUIRE(nparts > 0 && n > 0, MPN_ERR_BAD_SHAPE, "reduce_partials: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (n % 4 == 0 && mpn_aligned16(part) && mpn_aligned16(out)) {
        long long bx = ((n >> 2) + kThreads - 1) / kThreads;
        if (bx > 2048) bx = 2048;
        int nslices = (int)(1024 / bx);
        if (nslices > 16) nslices = 16;
        if (nslices > nparts / 8) nslices = nparts / 8;
        float* slab = const_cast<float*>(part);   // scratch: pass 1 folds range sums into it
        if (nslices >= 2) {
            reduce_partials_kernel<1><<<dim3((unsigned)bx, (unsigned)nslices), kThreads, 0, st>>>(slab, nparts, n, out, 0, 1.f,
                                                                                          nslices);
            MPN_LAUNCH_CHECK();
            reduce_partials_kernel<2><<<dim3((unsigned)bx, 1), kThreads, 0, st>>>(slab, nparts, n, out, accumulate, scale,
                                                                              nslices);
        } else {
            reduce_partials_kernel<0><<<dim3((unsigned)bx, 1), kThreads, 0, st>>>(slab, nparts, n, out, accumulate, scale, 1);
        }
    } else {  // rows that are not 16-byte multiples (the 64*18+18 head gradient): scalar kernel, tiny tensors only
        reduce_partials_scalar_kernel<<<(int)((n + kThreads - 1) / kThreads), kThreads, 0, st>>>(part, nparts, n, out,
                                                                                              accumulate, scale);
    }
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" size_t mpn_reduce_desc_bytes(void) { return sizeof(ReduceDesc); }

/* Fills ONE host-side descriptor (mpn_reduce_desc_bytes() bytes) of the batched slab reduction
 * out[j] = scale * sum_p part[p][j] and returns the number of blocks the job needs; block_begin = running sum. */
extern "C" int mpn_reduce_desc_fill(void* desc_host, const float* part, int nparts, long long n, float* out, float scale,
                                    int block_begin) {
    if (!desc_host || !part || !out || nparts <= 0 || n <= 0) return -1;
    ReduceDesc d;
    d.part = part; d.out = out; d.n = n; d.nparts = nparts; d.scale = scale; d.pad_ = 0;
    d.vec = (n % 4 == 0 && mpn_aligned16(part) && mpn_aligned16(out)) ? 1 : 0;
    const long long ncol = d.vec ? n / 4 : n;
    int slices = 1;                       // ~32 rows per thread, at least 16 columns per block
    while (slices < 16 && nparts / slices > 32) slices <<= 1;
    int cols = kThreads / slices;
    while (cols > 16 && (long long)cols / 2 >= ncol) cols >>= 1;   // tiny slabs: fewer columns, more row slices
    d.cols = cols;
    d.block_begin = block_begin;
    d.block_count = (int)((ncol + cols - 1) / cols);
    memcpy(desc_host, &d, sizeof(d));
    return d.block_count;
}

extern "C" int mpn_reduce_partials_batched(const void* descs_device, int ndesc, int total_blocks, mpn_stream_t stream) {
    MPN_REQUIRE(descs_device && ndesc > 0 && total_blocks > 0, MPN_ERR_BAD_ARG, "reduce batched: bad arguments");
    reduce_partials_batched_kernel<<<total_blocks, kThreads, 0, (hipStream_t)stream>>>((const ReduceDesc*)descs_device, ndesc);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_axpy(long long n, float a, const float* x, float* y, mpn_stream_t stream) {
    MPN_REQUIRE(x && y && n > 0, MPN_ERR_BAD_ARG, "axpy: bad arguments");
    long long blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 2048) blocks = 2048;
    axpy_kernel<<<(int)blocks, kThreads, 0, (hipStream_t)stream>>>(n, a, x, y);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* y[t][i] += a * x[t][i] for `count` tensors (host arrays of device pointers / element counts): one launch per 64 tensors.
 * The gradient of the weight-decay term, keypoints_model.py:129-138. */
extern "C" int mpn_axpy_batched(int count, const float* const* x, float* const* y, const long long* n, float a,
                                mpn_stream_t stream) {
    MPN_REQUIRE(count > 0 && x && y && n, MPN_ERR_BAD_ARG, "axpy_batched: bad arguments");
    for (int t0 = 0; t0 < count; t0 += kAxpyMax) {
        AxpyBatch b;
        b.count = count - t0 < kAxpyMax ? count - t0 : kAxpyMax;
        int begin = 0;
        for (int k = 0; k < b.count; ++k) {
            MPN_REQUIRE(x[t0 + k] && y[t0 + k] && n[t0 + k] > 0, MPN_ERR_BAD_ARG, "axpy_batched: tensor %d", t0 + k);
            b.x[k] = x[t0 + k]; b.y[k] = y[t0 + k]; b.n[k] = n[t0 + k]; b.begin[k] = begin;
            begin += (int)((n[t0 + k] + kAxpyChunk - 1) / kAxpyChunk);
        }
        b.begin[b.count] = begin;
        axpy_batched_kernel<<<begin, 256, 0, (hipStream_t)stream>>>(b, a);
        MPN_LAUNCH_CHECK();
    }
    return MPN_OK;
}

/* acc[0] += scale * l2_loss(w), l2_loss(t) = sum(t^2)/2 (tf.nn.l2_loss): the regularisation term that
 * keypoints_model.py:24-27,79 adds to the total loss when weight_decay > 0. Deterministic (one block, fixed order). */
extern "C" int mpn_l2_loss_accumulate(long long n, const float* w, float scale, float* acc, mpn_stream_t stream) {
    MPN_REQUIRE(w && acc && n > 0, MPN_ERR_BAD_ARG, "l2_loss_accumulate: bad arguments");
    l2_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(n, w, (double)scale, acc);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}


/* acc[0] += scale * sum_t l2_loss(w[t]) over `count` tensors (host arrays of device pointers / element counts) in two
 * launches: per-block f64 partial sums of 16 384 elements into `workspace`, then one block adds them in a fixed order.
 * workspace: mpn_l2_loss_batched_workspace_bytes(count, n) bytes. Deterministic. */
extern "C" size_t mpn_l2_loss_batched_workspace_bytes(int count, const long long* n) {
    size_t blocks = 0;
    for (int t = 0; t < count; ++t) blocks += (size_t)((n[t] + kL2Chunk - 1) / kL2Chunk);
    return blocks * sizeof(double);
}
extern "C" int mpn_l2_loss_batched(int count, const float* const* w, const long long* n, float scale, float* acc, void* workspace,
                                   size_t workspace_bytes, mpn_stream_t stream) {
    MPN_REQUIRE(count > 0 && w && n && acc && workspace, MPN_ERR_BAD_ARG, "l2_loss_batched: bad arguments");
    MPN_REQUIRE(workspace_bytes >= mpn_l2_loss_batched_workspace_bytes(count, n), MPN_ERR_WORKSPACE, "l2_loss_batched: workspace too small");
    double* partial = reinterpret_cast<double*>(workspace);
    int total = 0;
    for (int t0 = 0; t0 < count; t0 += kL2Max) {
        L2Batch b;
        b.count = count - t0 < kL2Max ? count - t0 : kL2Max;
        int begin = 0;
        for (int k = 0; k < b.count; ++k) {
            MPN_REQUIRE(w[t0 + k] && n[t0 + k] > 0, MPN_ERR_BAD_ARG, "l2_loss_batched: tensor %d", t0 + k);
            b.w[k] = w[t0 + k]; b.n[k] = n[t0 + k]; b.begin[k] = begin;
            begin += (int)((n[t0 + k] + kL2Chunk - 1) / kL2Chunk);
        }
        b.begin[b.count] = begin;
        l2_partial_kernel<<<begin, 256, 0, (hipStream_t)stream>>>(b, partial + total);
        MPN_LAUNCH_CHECK();
        total += begin;
    }
    l2_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(partial, total, (double)scale, acc);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
