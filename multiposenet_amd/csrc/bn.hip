// K4: fused batch normalisation pieces (training and inference), NHWC.
// Replaces tf.layers.batch_normalization(fused=True) at detector/backbones/mobilenet_v1.py:29-38
// and detector/utils/layer_utils.py:9-16 (momentum 0.95, eps 1e-3).
//
// Design: a conv kernel writes its RAW output once plus per-tile partial sums; bn_finalize
// turns the partials into a per-channel affine (scale, shift) and updates the moving
// statistics; every CONSUMER applies `act(x*scale+shift)` while loading (conv_mfma, dwconv,
// bilinear, head), so normalised activations never make a round trip through HBM.
// Backward: bn_bwd_reduce (sum g, sum g*xhat with the activation mask folded in),
// bn_bwd_finalize (dgamma, dbeta, per-channel coefficients), bn_bwd_apply (dx, in place).
// All reductions are deterministic: partial slabs + fixed-order f64 finalisation, no atomics.
#include "common.h"
#include <string.h>

namespace {

constexpr int kThreads = 256;
constexpr long long kApplyBlocks = 16384;   // bn_bwd_apply: 4 x 256 vectors per block and iteration

// thread -> (row lane, channel vector) mapping shared by the streaming kernels below
struct RowMap {
    int cvec;   // 16-byte vectors per row
    int ppb;    // rows processed per pass by one block
    int vg;     // this thread's channel vector
    int prow;   // this thread's row lane
    bool active;
};
__device__ __forceinline__ RowMap make_rowmap(int C, int VE) {
    RowMap m;
    m.cvec = C / VE;
    m.ppb = kThreads / m.cvec;
    if (m.ppb < 1) m.ppb = 1;
    m.vg = threadIdx.x % m.cvec;
    m.prow = threadIdx.x / m.cvec;
    m.active = m.prow < m.ppb;
    return m;
}

// Block-level reduction of NV per-thread values over threads that share a channel vector.
// out (per block) is written by the first `cvec` threads: part[which][c].
template <int NV, int VE>
__device__ __forceinline__ void block_reduce_store(const RowMap& m, float (&v)[NV][VE], float* smem,
                                                   float* __restrict__ dst, long long which_stride) {
    // smem: [kThreads][NV*VE]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < VE; ++j) smem[threadIdx.x * (NV * VE) + k * VE + j] = m.active ? v[k][j] : 0.f;
    __syncthreads();
    if ((int)threadIdx.x < m.cvec) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                float t = 0.f;
                for (int r = 0; r < m.ppb; ++r) t += smem[(r * m.cvec + m.vg) * (NV * VE) + k * VE + j];
                dst[k * which_stride + m.vg * VE + j] = t;
            }
    }
}

// ---------------------------------------------------------------- forward statistics (standalone)
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const T* __restrict__ x, long long M, int C,
                                                            float* __restrict__ part, int rows_per_block) {
    constexpr int VE = Vec16<T>::N;
    __shared__ float smem[kThreads * 2 * VE];
    const RowMap m = make_rowmap(C, VE);
    float acc[2][VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    if (m.active) {
        constexpr int U = 4;  // rows in flight per thread
        for (long long r = r0 + m.prow; r < r1; r += (long long)U * m.ppb) {
            Vec16<T> v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long rr = r + (long long)u * m.ppb;
                if (rr < r1) v[u].load(x + rr * C + m.vg * VE); else v[u].zero();
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float f[VE];
                v[u].unpack(f);
#pragma unroll
                for (int j = 0; j < VE; ++j) { acc[0][j] += f[j]; acc[1][j] += f[j] * f[j]; }
            }
        }
    }
    block_reduce_store<2, VE>(m, acc, smem, part + (long long)blockIdx.x * 2 * C, C);
}

// ---------------------------------------------------------------- finalize
// part [nparts][2][C] -> mean, biased var -> scale/shift (+ moving-average update with the
// unbiased variance, TF-1.15 fused batch-norm semantic). 16 channels x 16 part-lanes per block.
constexpr int kFinThreads = 1024, kFinLanes = 64;   // 64 part-lanes x 16 channels per block

// fixed-order f64 reduction of part[nparts][2][C] for the block's 16 channels; result in (s, q) of the threads pl == 0
// (= threadIdx.x < 16, channel cl). Thread t reads float4 pieces: t & 3 = which 4 of the 16 channels, t >> 2 = one of 256
// row lanes (a 1x1 conv at 256x256 leaves 16 384 partial rows: with one float per thread and 64 row lanes that finalize
// took 58 us in 2-8 blocks); the 16 row lanes of a wave combine by shuffles, the 16 waves through LDS.
// CPB = channels per block: 16, or 4 when there are thousands of partial rows (16 384 after a 1x1 conv at 256x256): the
// reduction is then bound by how many CUs pull on the slab, and C/16 = 4 blocks took 54 us for 8 MB.
template <int CPB>
__device__ __forceinline__ void reduce_parts(const float* __restrict__ part_, int nparts, int C_, int cblock,
                                             double (*red)[kFinLanes][16], double& s, double& q, int rstride = 1) {
    // rstride > 1: a compacted slab (slab_compact_kernel) - live rows 0, rstride, 2 rstride, ...
    const float* __restrict__ part = part_;
    const long long C = (long long)C_ * rstride;           // (row p of the walk below lives at p * rstride * 2 * C_)
    const int Cc = C_;
    constexpr int G4 = CPB / 4;
    constexpr int kRowLanes = kFinThreads / G4;
    const int t = threadIdx.x, g4 = t & (G4 - 1), r = t / G4;
    const int c4 = cblock * CPB + g4 * 4;
    double sa[4] = {0.0, 0.0, 0.0, 0.0}, qa[4] = {0.0, 0.0, 0.0, 0.0};
    if (c4 < Cc) {
        int p = r;
        for (; p + kRowLanes < nparts; p += 2 * kRowLanes) {   // 4 independent 16-byte loads in flight
            const float4 a0 = *reinterpret_cast<const float4*>(part + (long long)p * 2 * C + c4);
            const float4 b0 = *reinterpret_cast<const float4*>(part + (long long)p * 2 * C + Cc + c4);
            const float4 a1 = *reinterpret_cast<const float4*>(part + (long long)(p + kRowLanes) * 2 * C + c4);
            const float4 b1 = *reinterpret_cast<const float4*>(part + (long long)(p + kRowLanes) * 2 * C + Cc + c4);
            sa[0] += (double)a0.x + (double)a1.x; sa[1] += (double)a0.y + (double)a1.y;
            sa[2] += (double)a0.z + (double)a1.z; sa[3] += (double)a0.w + (double)a1.w;
            qa[0] += (double)b0.x + (double)b1.x; qa[1] += (double)b0.y + (double)b1.y;
            qa[2] += (double)b0.z + (double)b1.z; qa[3] += (double)b0.w + (double)b1.w;
        }
        for (; p < nparts; p += kRowLanes) {
            const float4 a0 = *reinterpret_cast<const float4*>(part + (long long)p * 2 * C + c4);
            const float4 b0 = *reinterpret_cast<const float4*>(part + (long long)p * 2 * C + Cc + c4);
            sa[0] += (double)a0.x; sa[1] += (double)a0.y; sa[2] += (double)a0.z; sa[3] += (double)a0.w;
            qa[0] += (double)b0.x; qa[1] += (double)b0.y; qa[2] += (double)b0.z; qa[3] += (double)b0.w;
        }
    }
    // the row lanes of a wave (lane bits above the channel-group bits), fixed butterfly order
#pragma unroll
    for (int o = G4; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) { sa[j] += __shfl_xor(sa[j], o, 64); qa[j] += __shfl_xor(qa[j], o, 64); }
    const int wave = t >> 6, lane = t & 63;
    if (lane < G4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[0][wave][lane * 4 + j] = sa[j]; red[1][wave][lane * 4 + j] = qa[j]; }
    }
    __syncthreads();
    s = 0.0; q = 0.0;
    if (t < CPB) {
        for (int w = 0; w < kFinThreads / 64; ++w) { s += red[0][w][t]; q += red[1][w][t]; }
    }
}

// Thousands of partial rows (a 1x1 layer at 256 x 256 leaves 16 384) make the finalize a 45-55 us launch of a few blocks pulling on
// megabytes: first add groups of kCompact consecutive rows IN PLACE (f64 inside a group, the group's sum rounded to f32 into
// its first row), one block per group - hundreds of blocks - then finalize rows 0, kCompact, 2 kCompact, ...
constexpr int kCompact = 32, kCompactFrom = 4096;
__global__ __launch_bounds__(256) void slab_compact_kernel(float* __restrict__ part, int nparts, int C) {
    __shared__ double red[256][4];
    const int cols4 = (2 * C) >> 2;                               // float4 columns of a row ([2][C] floats)
    const int r0 = blockIdx.x * kCompact, r1 = min(r0 + kCompact, nparts);
    for (int cb = 0; cb < cols4; cb += 256) {                     // (a row is wider than the block only from C = 512 on)
        const int span = min(cols4 - cb, 256);
        const int lanes = 256 / span > 0 ? 256 / span : 1;        // row lanes per column
        const int col = threadIdx.x % span, rl = threadIdx.x / span;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        if (rl < lanes)
            for (int r = r0 + rl; r < r1; r += lanes) {
                const float4 v = *reinterpret_cast<const float4*>(part + (long long)r * 2 * C + (cb + col) * 4);
                a[0] += (double)v.x; a[1] += (double)v.y; a[2] += (double)v.z; a[3] += (double)v.w;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) red[threadIdx.x][j] = a[j];
        __syncthreads();
        if (rl == 0) {
            double t[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k = 0; k < lanes; ++k)                       // fixed order
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] += red[k * span + col][j];
            *reinterpret_cast<float4*>(part + (long long)r0 * 2 * C + (cb + col) * 4) = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
        }
        __syncthreads();
    }
}
// returns the row stride the finalize must walk with (1: untouched) and updates nparts
static int compact_slab(float* part, int& nparts, int C, hipStream_t st) {
    if (nparts < kCompactFrom || C % 2 != 0) return 1;
    const int groups = (nparts + kCompact - 1) / kCompact;
    slab_compact_kernel<<<groups, 256, 0, st>>>(part, nparts, C);
    nparts = groups;
    return kCompact;
}

// part [nparts][2][C] -> mean, biased var -> scale/shift (+ moving-average update with the
// unbiased variance, TF-1.15 fused batch-norm semantic).
template <int CPB>
__device__ __forceinline__ void bn_finalize_block(
    int cblock, double (*red)[kFinLanes][16],
    const float* __restrict__ part, int nparts, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ mov_mean, float* __restrict__ mov_var, float momentum,
    float eps, float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean,
    float* __restrict__ save_invstd, int rstride = 1) {
    const int c = cblock * CPB + threadIdx.x;
    // the per-channel parameters are requested BEFORE the slab reduction (behind its barrier the loads would start a second
    // memory round trip of their own: a finalize is a 4 us launch made of two dependent round trips and little else)
    const bool mine = (int)threadIdx.x < CPB && c < C;
    const float g_c = mine ? gamma[c] : 0.f, b_c = mine ? beta[c] : 0.f;
    const float mm_c = (mine && mov_mean) ? mov_mean[c] : 0.f, mv_c = (mine && mov_mean) ? mov_var[c] : 0.f;
    double s, q;
    reduce_parts<CPB>(part, nparts, C, cblock, red, s, q, rstride);
    if (mine) {
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = g_c * invstd;
        scale[c] = sc;
        shift[c] = b_c - (float)mean * sc;
        if (save_mean) save_mean[c] = (float)mean;
        if (save_invstd) save_invstd[c] = invstd;
        if (mov_mean) {
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            mov_mean[c] = mm_c * momentum + (float)mean * (1.f - momentum);
            mov_var[c] = mv_c * momentum + (float)unbiased * (1.f - momentum);
        }
    }
}

template <int CPB>
__global__ __launch_bounds__(kFinThreads) void bn_finalize_kernel(
    const float* __restrict__ part, int nparts, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ mov_mean, float* __restrict__ mov_var, float momentum,
    float eps, float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean,
    float* __restrict__ save_invstd, int rstride) {
    __shared__ double red[2][kFinLanes][16];
    // blocks are dealt round-robin over the 8 XCDs (one L2 each) and every block reads a 16- or 64-byte piece of EVERY slab
    // row: give the blocks of one XCD NEIGHBOURING channel groups, so that they share 128-byte lines in their L2 instead of
    // every XCD fetching every line (16 384 rows x 64 channels: 8 x 8.4 MB through the fabric, 48 us)
    int cblock = blockIdx.x;
    if ((gridDim.x & 7) == 0) cblock = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    bn_finalize_block<CPB>(cblock, red, part, nparts, C, count, gamma, beta, mov_mean, mov_var, momentum, eps, scale, shift,
                           save_mean, save_invstd, rstride);
}

// ---- several independent layers' finalizes in ONE launch (the four pyramid levels of the subnet produce their
// statistics side by side; a finalize is ~6 us of launch + memory latency for microseconds of work). The descriptors
// live in device memory, built once by the host (shapes and pointers are static).
struct BnFinDesc {
    const float* part; const float* gamma; const float* beta;
    float* mov_mean; float* mov_var; float* scale; float* shift; float* save_mean; float* save_invstd;
    double count;
    int nparts, C, block_begin, pad_;
};
struct BnBwdFinDesc {
    const float* part; float* dgamma; float* dbeta; float* k1; float* k2;
    const float* mean; const float* invstd;   // non-NULL: the slab's second row holds sum g * x (RAW x), not sum g * xhat
    double count;
    int nparts, C, block_begin, pad_;
};
template <typename D>
__device__ __forceinline__ int fin_job(const D* __restrict__ descs, int ndesc, int* job_s) {
    if (threadIdx.x < 64) {
        int cnt = 0;
        for (int base = 0; base < ndesc; base += 64) {
            const int i = base + (int)threadIdx.x;
            const bool le = i < ndesc && descs[i].block_begin <= (int)blockIdx.x;
            cnt += __popcll(__ballot(le));
        }
        if (threadIdx.x == 0) *job_s = cnt - 1;
    }
    __syncthreads();
    return *job_s;
}
__global__ __launch_bounds__(kFinThreads) void bn_finalize_batched_kernel(const BnFinDesc* __restrict__ descs, int ndesc,
                                                                         float momentum, float eps) {
    __shared__ double red[2][kFinLanes][16];
    __shared__ int job_s;
    const BnFinDesc d = descs[fin_job(descs, ndesc, &job_s)];
    bn_finalize_block<16>(blockIdx.x - d.block_begin, red, d.part, d.nparts, d.C, d.count, d.gamma, d.beta, d.mov_mean, d.mov_var,
                          momentum, eps, d.scale, d.shift, d.save_mean, d.save_invstd);
}

__global__ void bn_inference_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ mov_mean, const float* __restrict__ mov_var,
                                    float eps, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * (1.0f / sqrtf(mov_var[c] + eps));
    scale[c] = sc;
    shift[c] = beta[c] - mov_mean[c] * sc;
}

// ---------------------------------------------------------------- materialise act(x*scale+shift)
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_act_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                long long nvec, int C,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int act) {
    constexpr int VE = Vec16<T>::N;
    const int cvec = C / VE;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long long)gridDim.x * kThreads) {
        const int c0 = (int)(i % cvec) * VE;
        Vec16<T> v;
        v.load(x + i * VE);
        float f[VE];
        v.unpack(f);
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            float t = f[j] * scale[c0 + j] + shift[c0 + j];
            if (act != MPN_ACT_NONE) t = fmaxf(t, 0.f);
            if (act == MPN_ACT_RELU6) t = fminf(t, 6.f);
            f[j] = t;
        }
        v.pack(f);
        v.store(y + i * VE);
    }
}

// ---------------------------------------------------------------- backward
// g = dA * act'(x*scale+shift);  partials of sum(g) and sum(g*xhat), xhat = (x-mean)*invstd
template <typename T>
__device__ __forceinline__ void bn_bwd_reduce_body(
    const T* __restrict__ dA, const T* __restrict__ x, long long M, int C, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd, int act,
    float* __restrict__ part, int rows_per_block, const int blk, const long long dA_stride, const long long x_stride) {
    // dA_stride / x_stride: elements between consecutive rows (C for dense tensors; larger for channel slices of wider ones)
    constexpr int VE = Vec16<T>::N;
    __shared__ float smem[kThreads * 2 * VE];
    const RowMap m = make_rowmap(C, VE);
    float acc[2][VE];
    float sc[VE], sh[VE], is[VE], nmi[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) {
        acc[0][j] = 0.f; acc[1][j] = 0.f;
        const int c = m.vg * VE + j;
        sc[j] = scale[c]; sh[j] = shift[c]; is[j] = invstd[c]; nmi[j] = -mean[c] * invstd[c];
    }
    const long long r0 = (long long)blk * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    if (m.active) {
        constexpr int U = 4;  // rows in flight per thread (8 x 16-byte loads)
        for (long long r = r0 + m.prow; r < r1; r += (long long)U * m.ppb) {
            Vec16<T> vd[U], vx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long rr = r + (long long)u * m.ppb;
                if (rr < r1) {
                    vd[u].load(dA + rr * dA_stride + m.vg * VE);
                    vx[u].load(x + rr * x_stride + m.vg * VE);
                } else {
                    vd[u].zero();
                    vx[u].zero();
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float d[VE], f[VE];
                vd[u].unpack(d);
                vx[u].unpack(f);
#pragma unroll
                for (int j = 0; j < VE; ++j) {
                    const float pre = f[j] * sc[j] + sh[j];
                    bool pass = true;
                    if (act != MPN_ACT_NONE) pass = pre > 0.f;
                    if (act == MPN_ACT_RELU6) pass = pass && (pre < 6.f);
                    const float g = pass ? d[j] : 0.f;   // (zero-filled tail rows contribute g = 0)
                    acc[0][j] += g;
                    acc[1][j] += g * (f[j] * is[j] + nmi[j]);
                }
            }
        }
    }
    block_reduce_store<2, VE>(m, acc, smem, part + (long long)blk * 2 * C, C);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(
    const T* __restrict__ dA, const T* __restrict__ x, long long M, int C, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd, int act,
    float* __restrict__ part, int rows_per_block) {
    bn_bwd_reduce_body<T>(dA, x, M, C, scale, shift, mean, invstd, act, part, rows_per_block, blockIdx.x, C, C);
}

// up to four independent layers (the pyramid levels of a subnet stage) in one grid, largest first
constexpr int kBnGroup = 5;   // (the keypoint subnet has 4 pyramid levels, the RetinaNet head 5)
struct BnBwdJob {
    void* dA; const void* x; long long M;
    const float *scale, *shift, *mean, *invstd, *k1, *k2, *add_ch0;
    float* part;
    int rows_per_block, dA_stride, x_stride, pad_;   // strides: elements between rows (C = dense)
};
struct BnBwdGroup { BnBwdJob j[kBnGroup]; int begin[kBnGroup + 1]; int njobs, C, act, pad_; };
__device__ __forceinline__ int bn_group_job(const BnBwdGroup& g) {
    int job = 0;
#pragma unroll
    for (int k = 1; k < kBnGroup; ++k)
        if (k < g.njobs && (int)blockIdx.x >= g.begin[k]) job = k;
    return job;
}
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_grouped_kernel(const BnBwdGroup g) {
    const int job = bn_group_job(g);
    const BnBwdJob& q = g.j[job];
    bn_bwd_reduce_body<T>((const T*)q.dA, (const T*)q.x, q.M, g.C, q.scale, q.shift, q.mean, q.invstd, g.act, q.part,
                          q.rows_per_block, (int)blockIdx.x - g.begin[job], q.dA_stride, q.x_stride);
}

// part [nparts][2][C] -> dgamma, dbeta and the two per-channel coefficients of bn_bwd_apply
template <int CPB>
__device__ __forceinline__ void bn_bwd_finalize_block(int cblock, double (*red)[kFinLanes][16], const float* __restrict__ part,
                                                      int nparts, int C, double count, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, float* __restrict__ k1,
                                                      float* __restrict__ k2, const float* __restrict__ mean = nullptr,
                                                      const float* __restrict__ invstd = nullptr, int rstride = 1) {
    const int c = cblock * CPB + threadIdx.x;
    const bool mine = (int)threadIdx.x < CPB && c < C;
    const float m_c = (mine && mean != nullptr) ? mean[c] : 0.f, i_c = (mine && mean != nullptr) ? invstd[c] : 0.f;   // (before the reduction: see bn_finalize_block)
    double s, q;
    reduce_parts<CPB>(part, nparts, C, cblock, red, s, q, rstride);
    if (mine) {
        // producers that reduce in their epilogue (mpn_conv_bwd_data_bn) sum g * x: sum g * xhat = invstd * (sum g x - mean * sum g)
        if (mean != nullptr) q = (q - (double)m_c * s) * (double)i_c;
        dbeta[c] = (float)s;
        dgamma[c] = (float)q;
        k1[c] = (float)(s / count);
        k2[c] = (float)(q / count);
    }
}
template <int CPB>
__global__ __launch_bounds__(kFinThreads) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nparts, int C,
                                                                      double count, float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta,
                                                                      float* __restrict__ k1, float* __restrict__ k2,
                                                                      const float* __restrict__ mean, const float* __restrict__ invstd, int rstride) {
    __shared__ double red[2][kFinLanes][16];
    bn_bwd_finalize_block<CPB>(blockIdx.x, red, part, nparts, C, count, dgamma, dbeta, k1, k2, mean, invstd, rstride);
}
__global__ __launch_bounds__(kFinThreads) void bn_bwd_finalize_batched_kernel(const BnBwdFinDesc* __restrict__ descs, int ndesc) {
    __shared__ double red[2][kFinLanes][16];
    __shared__ int job_s;
    const BnBwdFinDesc d = descs[fin_job(descs, ndesc, &job_s)];
    bn_bwd_finalize_block<16>(blockIdx.x - d.block_begin, red, d.part, d.nparts, d.C, d.count, d.dgamma, d.dbeta, d.k1, d.k2, d.mean, d.invstd);
}

// dx = scale * (g - k1 - xhat*k2), written over dA (same storage type); optional extra gradient
// added to channel 0 (the auxiliary segmentation loss on p_l[...,0], keypoints_model.py:59-66).
template <typename T, bool STRIDED = false>
__device__ __forceinline__ void bn_bwd_apply_body(
    T* __restrict__ dA, const T* __restrict__ x, long long nvec, int C, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ k1, const float* __restrict__ k2, int act, const float* __restrict__ add_ch0,
    const int blk, const int nblk, const long long dA_stride = 0, const long long x_stride = 0) {
    // STRIDED: dA / x are channel slices of wider tensors - vector ii lives at (ii / cvec) * stride + (ii % cvec) * VE
    // (cvec is a power of two); dense: at ii * VE
    constexpr int VE = Vec16<T>::N;
    constexpr int U = 4;   // 16-byte vectors of each operand in flight per thread
    const int cvec = C / VE;
    const int cshift = __builtin_ctz(cvec);
    auto off_d = [&](long long ii) { return STRIDED ? (ii >> cshift) * dA_stride + (ii & (cvec - 1)) * VE : ii * VE; };
    auto off_x = [&](long long ii) { return STRIDED ? (ii >> cshift) * x_stride + (ii & (cvec - 1)) * VE : ii * VE; };
    // the grid stride (a multiple of kThreads) is a multiple of cvec (a power of two <= kThreads, checked by the launcher):
    // a thread keeps its channel vector, so the six per-channel parameters fold into registers ONCE:
    //   out = sc*(g - k1 - xhat*k2) = sc*g + cb*x + cc,  cb = -sc*k2*invstd,  cc = -sc*(k1 - mean*invstd*k2)
    // (re-loading them per element was 48 dword loads per 16-byte vector: bound by the texture unit, not by HBM)
    // a block takes U consecutive 256-vector pieces per iteration: 16 KB of contiguous addresses per operand (the four vectors
    // of a thread one grid stride apart - four 4-KB pieces per block, 8 MB from each other - ran 13-17 % slower on every
    // layer shape, tools/bench_apply.py; with up to 16 384 blocks the big layers are one iteration)
    const long long i0 = (long long)blk * (U * kThreads) + threadIdx.x;
    const int vg = (int)(i0 % cvec);
    float sc[VE], sh[VE], cb[VE], cc[VE];
#pragma unroll
    for (int j4 = 0; j4 < VE; j4 += 4) {
        const int c = vg * VE + j4;
        const float4 s4 = *reinterpret_cast<const float4*>(scale + c), h4 = *reinterpret_cast<const float4*>(shift + c);
        const float4 m4 = *reinterpret_cast<const float4*>(mean + c), i4 = *reinterpret_cast<const float4*>(invstd + c);
        const float4 a4 = *reinterpret_cast<const float4*>(k1 + c), b4 = *reinterpret_cast<const float4*>(k2 + c);
        const float ss[4] = {s4.x, s4.y, s4.z, s4.w}, hh[4] = {h4.x, h4.y, h4.z, h4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
        const float ii[4] = {i4.x, i4.y, i4.z, i4.w}, aa[4] = {a4.x, a4.y, a4.z, a4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sc[j4 + j] = ss[j];
            sh[j4 + j] = hh[j];
            cb[j4 + j] = -ss[j] * bb[j] * ii[j];
            cc[j4 + j] = -ss[j] * (aa[j] - mm[j] * ii[j] * bb[j]);
        }
    }
    const float lo = (act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    constexpr long long stride = kThreads;
    const long long step = (long long)nblk * (U * kThreads);
    for (long long i = i0; i < nvec; i += step) {
        Vec16<T> vd[U], vx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long ii = i + u * stride;
            const long long ic = ii < nvec ? ii : i;   // unconditional loads from a valid address (a predicated load waits)
            vd[u].load(dA + off_d(ic));
            {   // x is dead after this pass: a non-temporal load keeps it from displacing dx, which the conv data / weight
                // gradients read next (256x256x64 layer: 165 -> 131 us, the step -0.85 %)
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(x + off_x(ic)));
                __builtin_memcpy(&vx[u].raw, &q, 16);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long ii = i + u * stride;
            float d[VE], f[VE];
            vd[u].unpack(d);
            vx[u].unpack(f);
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                const float pre = f[j] * sc[j] + sh[j];
                const float g = (pre > lo && pre < hi) ? d[j] : 0.f;
                d[j] = sc[j] * g + (cb[j] * f[j] + cc[j]);
            }
            if (add_ch0 != nullptr && vg == 0 && ii < nvec) d[0] += add_ch0[ii / cvec];
            vd[u].pack(d);
            if (ii < nvec) vd[u].store(dA + off_d(ii));
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(
    T* __restrict__ dA, const T* __restrict__ x, long long nvec, int C, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ k1, const float* __restrict__ k2, int act, const float* __restrict__ add_ch0) {
    bn_bwd_apply_body<T>(dA, x, nvec, C, scale, shift, mean, invstd, k1, k2, act, add_ch0, blockIdx.x, gridDim.x);
}
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_grouped_kernel(const BnBwdGroup g) {
    const int job = bn_group_job(g);
    const BnBwdJob& q = g.j[job];
    if (q.dA_stride != g.C || q.x_stride != g.C)   // (block-uniform)
        bn_bwd_apply_body<T, true>((T*)q.dA, (const T*)q.x, q.M * (g.C / Vec16<T>::N), g.C, q.scale, q.shift, q.mean, q.invstd, q.k1,
                                   q.k2, g.act, q.add_ch0, (int)blockIdx.x - g.begin[job], g.begin[job + 1] - g.begin[job],
                                   q.dA_stride, q.x_stride);
    else
        bn_bwd_apply_body<T>((T*)q.dA, (const T*)q.x, q.M * (g.C / Vec16<T>::N), g.C, q.scale, q.shift, q.mean, q.invstd, q.k1, q.k2,
                             g.act, q.add_ch0, (int)blockIdx.x - g.begin[job], g.begin[job + 1] - g.begin[job]);
}

// any C (a channel vector per thread changes from iteration to iteration: parameters re-loaded per element)
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_generic_kernel(
    T* __restrict__ dA, const T* __restrict__ x, long long nvec, int C, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ k1, const float* __restrict__ k2, int act, const float* __restrict__ add_ch0) {
    constexpr int VE = Vec16<T>::N;
    const int cvec = C / VE;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long long)gridDim.x * kThreads) {
        const int vg = (int)(i % cvec);
        const int c0 = vg * VE;
        Vec16<T> vd, vx;
        vd.load(dA + i * VE);
        vx.load(x + i * VE);
        float d[VE], f[VE];
        vd.unpack(d);
        vx.unpack(f);
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            const int c = c0 + j;
            const float sc = scale[c];
            const float pre = f[j] * sc + shift[c];
            bool pass = true;
            if (act != MPN_ACT_NONE) pass = pre > 0.f;
            if (act == MPN_ACT_RELU6) pass = pass && (pre < 6.f);
            const float g = pass ? d[j] : 0.f;
            const float xhat = (f[j] - mean[c]) * invstd[c];
            d[j] = sc * (g - k1[c] - xhat * k2[c]);
        }
        if (add_ch0 != nullptr && vg == 0) d[0] += add_ch0[i / cvec];
        vd.pack(d);
        vd.store(dA + i * VE);
    }
}

int stream_blocks(long long nvec) {
    long long b = (nvec + kThreads - 1) / kThreads;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

int check_rows(long long M, int C, int dtype, int* ve_out) {
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "bn: dtype %d", dtype);
    const int ve = dtype == MPN_F32 ? 4 : 8;
    MPN_REQUIRE(M > 0 && C > 0 && C % ve == 0, MPN_ERR_BAD_SHAPE, "bn: C (%d) must be a multiple of %d", C, ve);
    MPN_REQUIRE(C / ve <= kThreads, MPN_ERR_BAD_SHAPE, "bn: C too large (%d)", C);
    *ve_out = ve;
    return MPN_OK;
}

}  // namespace

extern "C" int mpn_bn_stats_num_parts(long long M) {
    // 128 rows per block, at most 2048 blocks (8 waves per SIMD on 256 CUs); small tensors (the 32x32 and 16x16 maps)
    // go down to 32 rows per block for up to 1024 blocks: at 128 rows they occupied 64-256 CUs with ONE block each
    // (32 KB of loads in flight per CU, 0.9-3 TB/s)
    long long parts = (M + 127) / 128;
    if (parts > 2048) parts = 2048;
    if (parts < 1024) {
        parts = (M + 31) / 32;
        if (parts > 1024) parts = 1024;
    }
    if (parts < 1) parts = 1;
    return (int)parts;
}

static long long rows_per_block_for(long long M, int nparts) { return (M + nparts - 1) / nparts; }

extern "C" int mpn_bn_stats(const void* x, long long M, int C, int dtype, float* part, mpn_stream_t stream) {
    int ve;
    if (int rc = check_rows(M, C, dtype, &ve)) return rc;
    MPN_REQUIRE(x && part, MPN_ERR_BAD_ARG, "bn_stats: null pointer");
    const int nparts = mpn_bn_stats_num_parts(M);
    const int rpb = (int)rows_per_block_for(M, nparts);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (bn_stats_kernel<T><<<nparts, kThreads, 0, st>>>((const T*)x, M, C, part, rpb)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_bn_finalize(float* part, int nparts, int C, long long count, const float* gamma,
                               const float* beta, float* moving_mean, float* moving_var, float momentum, float eps,
                               float* scale, float* shift, float* save_mean, float* save_invstd,
                               mpn_stream_t stream) {
    MPN_REQUIRE(part && gamma && beta && scale && shift, MPN_ERR_BAD_ARG, "bn_finalize: null pointer");
    MPN_REQUIRE(nparts > 0 && C > 0 && count > 0, MPN_ERR_BAD_SHAPE, "bn_finalize: bad sizes");
    MPN_REQUIRE(C % 4 == 0 && mpn_aligned16(part), MPN_ERR_BAD_ALIGN, "bn_finalize: C must be a multiple of 4 and part 16-byte aligned");
    MPN_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), MPN_ERR_BAD_ARG, "bn_finalize: moving stats");
    const int rstride = compact_slab(part, nparts, C, (hipStream_t)stream);      // (thousands of rows: groups of 32 added in place first)
    bn_finalize_kernel<16><<<(C + 15) / 16, kFinThreads, 0, (hipStream_t)stream>>>(
        part, nparts, C, (double)count, gamma, beta, moving_mean, moving_var, momentum, eps, scale, shift, save_mean,
        save_invstd, rstride);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" size_t mpn_bn_fin_desc_bytes(void) { return sizeof(BnFinDesc); }
extern "C" size_t mpn_bn_bwd_fin_desc_bytes(void) { return sizeof(BnBwdFinDesc); }

/* One host-side descriptor of the batched finalize (mpn_bn_fin_desc_bytes() bytes); returns the blocks the job needs,
 * block_begin = running sum over the jobs; -1 on bad arguments. The caller copies the array to the device once. */
extern "C" int mpn_bn_fin_desc_fill(void* desc_host, const float* part, int nparts, int C, long long count, const float* gamma,
                                    const float* beta, float* moving_mean, float* moving_var, float* scale, float* shift,
                                    float* save_mean, float* save_invstd, int block_begin) {
    if (!desc_host || !part || !gamma || !beta || !scale || !shift || nparts <= 0 || C <= 0 || C % 4 != 0 || count <= 0 ||
        !mpn_aligned16(part) || ((moving_mean == nullptr) != (moving_var == nullptr)))
        return -1;
    BnFinDesc d;
    d.part = part; d.gamma = gamma; d.beta = beta; d.mov_mean = moving_mean; d.mov_var = moving_var; d.scale = scale;
    d.shift = shift; d.save_mean = save_mean; d.save_invstd = save_invstd; d.count = (double)count;
    d.nparts = nparts; d.C = C; d.block_begin = block_begin; d.pad_ = 0;
    memcpy(desc_host, &d, sizeof(d));
    return (C + 15) / 16;
}
extern "C" int mpn_bn_bwd_fin_desc_fill(void* desc_host, const float* part, int nparts, int C, long long count, float* dgamma,
                                        float* dbeta, float* k1, float* k2, int block_begin) {
    if (!desc_host || !part || !dgamma || !dbeta || !k1 || !k2 || nparts <= 0 || C <= 0 || C % 4 != 0 || count <= 0 ||
        !mpn_aligned16(part))
        return -1;
    BnBwdFinDesc d;
    d.part = part; d.dgamma = dgamma; d.dbeta = dbeta; d.k1 = k1; d.k2 = k2; d.mean = nullptr; d.invstd = nullptr; d.count = (double)count;
    d.nparts = nparts; d.C = C; d.block_begin = block_begin; d.pad_ = 0;
    memcpy(desc_host, &d, sizeof(d));
    return (C + 15) / 16;
}
/* The same for a slab whose second row holds sum g * x with the RAW x (written by mpn_conv_bwd_data_bn_grouped): mean / invstd
 * = the layer's saved batch statistics. */
extern "C" int mpn_bn_bwd_fin_desc_fill_raw(void* desc_host, const float* part, int nparts, int C, long long count, float* dgamma,
                                            float* dbeta, float* k1, float* k2, const float* mean, const float* invstd,
                                            int block_begin) {
    if (!mean || !invstd) return -1;
    const int blocks = mpn_bn_bwd_fin_desc_fill(desc_host, part, nparts, C, count, dgamma, dbeta, k1, k2, block_begin);
    if (blocks <= 0) return blocks;
    BnBwdFinDesc d;
    memcpy(&d, desc_host, sizeof(d));
    d.mean = mean; d.invstd = invstd;
    memcpy(desc_host, &d, sizeof(d));
    return blocks;
}
extern "C" int mpn_bn_finalize_batched(const void* descs_device, int ndesc, int total_blocks, float momentum, float eps,
                                       mpn_stream_t stream) {
    MPN_REQUIRE(descs_device && ndesc > 0 && total_blocks > 0, MPN_ERR_BAD_ARG, "bn_finalize_batched: bad arguments");
    bn_finalize_batched_kernel<<<total_blocks, kFinThreads, 0, (hipStream_t)stream>>>((const BnFinDesc*)descs_device, ndesc, momentum, eps);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
extern "C" int mpn_bn_bwd_finalize_batched(const void* descs_device, int ndesc, int total_blocks, mpn_stream_t stream) {
    MPN_REQUIRE(descs_device && ndesc > 0 && total_blocks > 0, MPN_ERR_BAD_ARG, "bn_bwd_finalize_batched: bad arguments");
    bn_bwd_finalize_batched_kernel<<<total_blocks, kFinThreads, 0, (hipStream_t)stream>>>((const BnBwdFinDesc*)descs_device, ndesc);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_bn_inference_affine(int C, const float* gamma, const float* beta, const float* moving_mean,
                                       const float* moving_var, float eps, float* scale, float* shift,
                                       mpn_stream_t stream) {
    MPN_REQUIRE(gamma && beta && moving_mean && moving_var && scale && shift, MPN_ERR_BAD_ARG, "bn_inference: null");
    MPN_REQUIRE(C > 0, MPN_ERR_BAD_SHAPE, "bn_inference: C");
    bn_inference_kernel<<<(C + 255) / 256, 256, 0, (hipStream_t)stream>>>(C, gamma, beta, moving_mean, moving_var, eps,
                                                                        scale, shift);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_bn_act_apply(const void* x, void* y, long long M, int C, int dtype, const float* scale,
                                const float* shift, int act, mpn_stream_t stream) {
    int ve;
    if (int rc = check_rows(M, C, dtype, &ve)) return rc;
    MPN_REQUIRE(x && y && scale && shift, MPN_ERR_BAD_ARG, "bn_act_apply: null pointer");
    const long long nvec = M * (C / ve);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (bn_act_apply_kernel<T><<<stream_blocks(nvec), kThreads, 0, st>>>(
                                  (const T*)x, (T*)y, nvec, C, scale, shift, act)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_bn_bwd_reduce(const void* dA, const void* x, long long M, int C, int dtype, const float* scale,
                                 const float* shift, const float* mean, const float* invstd, int act, float* part,
                                 mpn_stream_t stream) {
    int ve;
    if (int rc = check_rows(M, C, dtype, &ve)) return rc;
    MPN_REQUIRE(dA && x && scale && shift && mean && invstd && part, MPN_ERR_BAD_ARG, "bn_bwd_reduce: null pointer");
    const int nparts = mpn_bn_stats_num_parts(M);
    const int rpb = (int)rows_per_block_for(M, nparts);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (bn_bwd_reduce_kernel<T><<<nparts, kThreads, 0, st>>>(
                                  (const T*)dA, (const T*)x, M, C, scale, shift, mean, invstd, act, part, rpb)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* The backward passes of up to four independent layers of one channel count (the pyramid levels of a subnet stage) in ONE
 * grid each, largest first; per job dA, x, M (rows), scale .. invstd, part (reduce) / k1, k2, add_ch0 (apply). Results are
 * those of the per-layer launches, bit for bit; shapes the grouped grids do not cover run as those launches. */
static bool bn_group_ok(int njobs, int C, int dtype) {
    const int ve = dtype == MPN_F32 ? 4 : 8;
    return njobs <= kBnGroup && C % ve == 0 && kThreads % (C / ve) == 0;
}
extern "C" int mpn_bn_bwd_reduce_grouped(int njobs, void* const* dA, const void* const* x, const long long* M, int C, int dtype,
                                         const float* const* scale, const float* const* shift, const float* const* mean,
                                         const float* const* invstd, int act, float* const* part, const int* dA_stride,
                                         const int* x_stride, mpn_stream_t stream) {
    MPN_REQUIRE(njobs > 0 && dA && x && M && scale && shift && mean && invstd && part, MPN_ERR_BAD_ARG, "bn_bwd_reduce_grouped: bad arguments");
    MPN_REQUIRE(bn_group_ok(njobs, C, dtype) || (dA_stride == nullptr && x_stride == nullptr), MPN_ERR_BAD_SHAPE,
                "bn_bwd_reduce_grouped: row strides need a configuration the grouped grid covers");
    if (!bn_group_ok(njobs, C, dtype)) {
        for (int j = 0; j < njobs; ++j)
            if (int rc = mpn_bn_bwd_reduce(dA[j], x[j], M[j], C, dtype, scale[j], shift[j], mean[j], invstd[j], act, part[j], stream)) return rc;
        return MPN_OK;
    }
    BnBwdGroup g = {};
    int begin = 0;
    for (int j = 0; j < njobs; ++j) {
        int ve;
        if (int rc = check_rows(M[j], C, dtype, &ve)) return rc;
        MPN_REQUIRE(dA[j] && x[j] && scale[j] && shift[j] && mean[j] && invstd[j] && part[j], MPN_ERR_BAD_ARG, "bn_bwd_reduce_grouped: null pointer");
        const int nparts = mpn_bn_stats_num_parts(M[j]);
        BnBwdJob& q = g.j[j];
        q.dA = dA[j]; q.x = x[j]; q.M = M[j]; q.scale = scale[j]; q.shift = shift[j]; q.mean = mean[j]; q.invstd = invstd[j];
        q.part = part[j]; q.rows_per_block = (int)rows_per_block_for(M[j], nparts);
        q.dA_stride = (dA_stride && dA_stride[j] > 0) ? dA_stride[j] : C;
        q.x_stride = (x_stride && x_stride[j] > 0) ? x_stride[j] : C;
        MPN_REQUIRE(q.dA_stride >= C && q.x_stride >= C && q.dA_stride % ve == 0 && q.x_stride % ve == 0, MPN_ERR_BAD_SHAPE,
                    "bn_bwd_reduce_grouped: bad row stride");
        g.begin[j] = begin;
        begin += nparts;
    }
    for (int j = njobs; j <= kBnGroup; ++j) g.begin[j] = begin;
    g.njobs = njobs; g.C = C; g.act = act;
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (bn_bwd_reduce_grouped_kernel<T><<<begin, kThreads, 0, st>>>(g)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
extern "C" int mpn_bn_bwd_apply_grouped(int njobs, void* const* dA, const void* const* x, const long long* M, int C, int dtype,
                                        const float* const* scale, const float* const* shift, const float* const* mean,
                                        const float* const* invstd, const float* const* k1, const float* const* k2, int act,
                                        const float* const* add_ch0, const int* dA_stride, const int* x_stride,
                                        mpn_stream_t stream) {
    MPN_REQUIRE(njobs > 0 && dA && x && M && scale && shift && mean && invstd && k1 && k2 && add_ch0, MPN_ERR_BAD_ARG,
                "bn_bwd_apply_grouped: bad arguments");
    MPN_REQUIRE(bn_group_ok(njobs, C, dtype) || (dA_stride == nullptr && x_stride == nullptr), MPN_ERR_BAD_SHAPE,
                "bn_bwd_apply_grouped: row strides need a configuration the grouped grid covers");
    if (!bn_group_ok(njobs, C, dtype)) {
        for (int j = 0; j < njobs; ++j)
            if (int rc = mpn_bn_bwd_apply(dA[j], x[j], M[j], C, dtype, scale[j], shift[j], mean[j], invstd[j], k1[j], k2[j], act,
                                          add_ch0[j], stream)) return rc;
        return MPN_OK;
    }
    const int ve = dtype == MPN_F32 ? 4 : 8;
    BnBwdGroup g = {};
    int begin = 0;
    for (int j = 0; j < njobs; ++j) {
        int v2;
        if (int rc = check_rows(M[j], C, dtype, &v2)) return rc;
        MPN_REQUIRE(dA[j] && x[j] && scale[j] && shift[j] && mean[j] && invstd[j] && k1[j] && k2[j], MPN_ERR_BAD_ARG, "bn_bwd_apply_grouped: null pointer");
        BnBwdJob& q = g.j[j];
        q.dA = dA[j]; q.x = x[j]; q.M = M[j]; q.scale = scale[j]; q.shift = shift[j]; q.mean = mean[j]; q.invstd = invstd[j];
        q.k1 = k1[j]; q.k2 = k2[j]; q.add_ch0 = add_ch0[j];
        q.dA_stride = (dA_stride && dA_stride[j] > 0) ? dA_stride[j] : C;
        q.x_stride = (x_stride && x_stride[j] > 0) ? x_stride[j] : C;
        MPN_REQUIRE(q.dA_stride >= C && q.x_stride >= C && q.dA_stride % ve == 0 && q.x_stride % ve == 0, MPN_ERR_BAD_SHAPE,
                    "bn_bwd_apply_grouped: bad row stride");
        const long long nvec = M[j] * (C / ve);
        long long blocks = (nvec + 4 * kThreads - 1) / (4 * kThreads);   // as the per-layer launch
        if (blocks > kApplyBlocks) blocks = kApplyBlocks;
        g.begin[j] = begin;
        begin += (int)blocks;
    }
    for (int j = njobs; j <= kBnGroup; ++j) g.begin[j] = begin;
    g.njobs = njobs; g.C = C; g.act = act;
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (bn_bwd_apply_grouped_kernel<T><<<begin, kThreads, 0, st>>>(g)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_bn_bwd_finalize(float* part, int nparts, int C, long long count, float* dgamma,
                                   float* dbeta, float* k1, float* k2, mpn_stream_t stream) {
    MPN_REQUIRE(part && dgamma && dbeta && k1 && k2, MPN_ERR_BAD_ARG, "bn_bwd_finalize: null pointer");
    MPN_REQUIRE(nparts > 0 && C > 0 && count > 0, MPN_ERR_BAD_SHAPE, "bn_bwd_finalize: bad sizes");
    MPN_REQUIRE(C % 4 == 0 && mpn_aligned16(part), MPN_ERR_BAD_ALIGN, "bn_bwd_finalize: C must be a multiple of 4 and part 16-byte aligned");
    const int rstride = compact_slab(part, nparts, C, (hipStream_t)stream);
    bn_bwd_finalize_kernel<16><<<(C + 15) / 16, kFinThreads, 0, (hipStream_t)stream>>>(part, nparts, C, (double)count,
                                                                               dgamma, dbeta, k1, k2, nullptr, nullptr, rstride);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* The same for a slab whose second row holds sum g * x with the RAW x (mpn_conv_bwd_data_bn): mean / invstd = the layer's saved
 * batch statistics; sum g * xhat = invstd * (sum g x - mean * sum g) in f64. */
extern "C" int mpn_bn_bwd_finalize_raw(float* part, int nparts, int C, long long count, float* dgamma, float* dbeta, float* k1,
                                       float* k2, const float* mean, const float* invstd, mpn_stream_t stream) {
    MPN_REQUIRE(part && dgamma && dbeta && k1 && k2 && mean && invstd, MPN_ERR_BAD_ARG, "bn_bwd_finalize_raw: null pointer");
    MPN_REQUIRE(nparts > 0 && C > 0 && count > 0, MPN_ERR_BAD_SHAPE, "bn_bwd_finalize_raw: bad sizes");
    MPN_REQUIRE(C % 4 == 0 && mpn_aligned16(part), MPN_ERR_BAD_ALIGN, "bn_bwd_finalize_raw: C must be a multiple of 4 and part 16-byte aligned");
    const int rstride = compact_slab(part, nparts, C, (hipStream_t)stream);
    bn_bwd_finalize_kernel<16><<<(C + 15) / 16, kFinThreads, 0, (hipStream_t)stream>>>(part, nparts, C, (double)count,
                                                                               dgamma, dbeta, k1, k2, mean, invstd, rstride);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_bn_bwd_apply(void* dA, const void* x, long long M, int C, int dtype, const float* scale,
                                const float* shift, const float* mean, const float* invstd, const float* k1,
                                const float* k2, int act, const float* add_ch0, mpn_stream_t stream) {
    int ve;
    if (int rc = check_rows(M, C, dtype, &ve)) return rc;
    MPN_REQUIRE(dA && x && scale && shift && mean && invstd && k1 && k2, MPN_ERR_BAD_ARG, "bn_bwd_apply: null pointer");
    const long long nvec = M * (C / ve);
    hipStream_t st = (hipStream_t)stream;
    if (kThreads % (C / ve) == 0) {
        long long blocks = (nvec + 4 * kThreads - 1) / (4 * kThreads);   // 4 vectors per thread and iteration
        if (blocks > kApplyBlocks) blocks = kApplyBlocks;
        MPN_DISPATCH_DTYPE(dtype, (bn_bwd_apply_kernel<T><<<(unsigned)blocks, kThreads, 0, st>>>(
                                      (T*)dA, (const T*)x, nvec, C, scale, shift, mean, invstd, k1, k2, act, add_ch0)));
    } else {
        MPN_DISPATCH_DTYPE(dtype, (bn_bwd_apply_generic_kernel<T><<<stream_blocks(nvec), kThreads, 0, st>>>(
                                      (T*)dA, (const T*)x, nvec, C, scale, shift, mean, invstd, k1, k2, act, add_ch0)));
    }
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
