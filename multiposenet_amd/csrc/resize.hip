// K9/K10 + K7-backward: legacy bilinear up-sampling straight into the concat buffer, and the
// backward of the FPN's nearest-2x upsample.
//   * mpn_bilinear_up_fwd replaces tf.image.resize_bilinear (TF-1.15 legacy: align_corners=False,
//     no half-pixel centres; src = dst*in/out, lo=floor, hi=min(lo+1,in-1)) at
//     detector/keypoint_subnet.py:86 for the integer factors 1,2,4,8, applies the producer's
//     batch-norm affine + ReLU on load, and writes at a channel offset of the 512-channel concat
//     tensor, which removes tf.concat (keypoint_subnet.py:37) entirely.
//   * mpn_bilinear_up_bwd is its transpose in gather form (no atomics, deterministic).
//   * mpn_sumpool2x2_add is the gradient of nearest_neighbor_upsample (detector/fpn.py:58-76).
#include "common.h"

namespace {
constexpr int kThreads = 256;

template <typename T>
__device__ __forceinline__ void load_act(const T* p, const float* sc, const float* sh, int act, int c0, float* f) {
    constexpr int VE = Vec16<T>::N;
    Vec16<T> v;
    v.load(p);
    v.unpack(*reinterpret_cast<float(*)[VE]>(f));
    if (sc != nullptr) {
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            float t = f[j] * sc[c0 + j] + sh[c0 + j];
            if (act != MPN_ACT_NONE) t = fmaxf(t, 0.f);
            if (act == MPN_ACT_RELU6) t = fminf(t, 6.f);
            f[j] = t;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bilinear_up_fwd_kernel(
    const T* __restrict__ x, T* __restrict__ y, int N, int h, int w, int C, int u, int y_coff, int y_ctot,
    const float* __restrict__ sc, const float* __restrict__ sh, int act, long long total) {
    constexpr int VE = Vec16<T>::N;
    const int cvec = C / VE;
    const int OH = h * u, OW = w * u;
    const float inv = 1.0f / (float)u;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
        const int vg = (int)(i % cvec);
        long long r = i / cvec;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        const int y0 = oy / u, x0 = ox / u;
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float fy = (float)(oy - y0 * u) * inv, fx = (float)(ox - x0 * u) * inv;
        const int c0 = vg * VE;
        const T* base = x + (long long)n * h * w * C + c0;
        float a[VE], b[VE], c[VE], d[VE];
        load_act<T>(base + ((long long)y0 * w + x0) * C, sc, sh, act, c0, a);
        load_act<T>(base + ((long long)y0 * w + x1) * C, sc, sh, act, c0, b);
        load_act<T>(base + ((long long)y1 * w + x0) * C, sc, sh, act, c0, c);
        load_act<T>(base + ((long long)y1 * w + x1) * C, sc, sh, act, c0, d);
        float o[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            const float top = a[j] + (b[j] - a[j]) * fx;
            const float bot = c[j] + (d[j] - c[j]) * fx;
            o[j] = top + (bot - top) * fy;
        }
        Vec16<T> ov;
        ov.pack(o);
        ov.store(y + (((long long)n * OH + oy) * OW + ox) * y_ctot + y_coff + c0);
    }
}

// dX[n,iy,ix,c] = sum over output pixels of wy*wx*dY (gather form)
template <typename T>
__global__ __launch_bounds__(kThreads) void bilinear_up_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N,
                                                                   int h, int w, int C, int u, int y_coff, int y_ctot,
                                                                   long long total) {
    constexpr int VE = Vec16<T>::N;
    const int cvec = C / VE;
    const int OH = h * u, OW = w * u;
    const float inv = 1.0f / (float)u;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
        const int vg = (int)(i % cvec);
        long long r = i / cvec;
        const int ix = (int)(r % w); r /= w;
        const int iy = (int)(r % h);
        const int n = (int)(r / h);
        float acc[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) acc[j] = 0.f;
        const int oy_lo = max(0, (iy - 1) * u), oy_hi = iy * u + u - 1;
        const int ox_lo = max(0, (ix - 1) * u), ox_hi = ix * u + u - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const int y0 = oy / u, y1 = min(y0 + 1, h - 1);
            const float fy = (float)(oy - y0 * u) * inv;
            const float wy = (y0 == iy ? 1.f - fy : 0.f) + (y1 == iy ? fy : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const int x0 = ox / u, x1 = min(x0 + 1, w - 1);
                const float fx = (float)(ox - x0 * u) * inv;
                const float wx = (x0 == ix ? 1.f - fx : 0.f) + (x1 == ix ? fx : 0.f);
                if (wx == 0.f) continue;
                Vec16<T> v;
                v.load(dy + (((long long)n * OH + oy) * OW + ox) * y_ctot + y_coff + vg * VE);
                float f[VE];
                v.unpack(f);
                const float wgt = wy * wx;
#pragma unroll
                for (int j = 0; j < VE; ++j) acc[j] += wgt * f[j];
            }
        }
        Vec16<T> ov;
        ov.pack(acc);
        ov.store(dx + i * VE);
    }
}

// dst[n,y,x,c] (+)= sum_{dy,dx in 0..1} src[n,2y+dy,2x+dx,c]
template <typename T>
__global__ __launch_bounds__(kThreads) void sumpool2x2_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int h,
                                                              int w, int C, int accumulate, long long total) {
    constexpr int VE = Vec16<T>::N;
    const int cvec = C / VE;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
        const int vg = (int)(i % cvec);
        long long r = i / cvec;
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h);
        const int n = (int)(r / h);
        float acc[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) acc[j] = 0.f;
        if (accumulate) {
            Vec16<T> v;
            v.load(dst + i * VE);
            v.unpack(acc);
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                Vec16<T> v;
                v.load(src + (((long long)n * 2 * h + 2 * y + dy) * 2 * w + 2 * x + dx) * C + vg * VE);
                float f[VE];
                v.unpack(f);
#pragma unroll
                for (int j = 0; j < VE; ++j) acc[j] += f[j];
            }
        Vec16<T> ov;
        ov.pack(acc);
        ov.store(dst + i * VE);
    }
}

// dst += src (same shape)
template <typename T>
__global__ __launch_bounds__(kThreads) void add_inplace_kernel(T* __restrict__ dst, const T* __restrict__ src, long long nvec) {
    constexpr int VE = Vec16<T>::N;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long long)gridDim.x * kThreads) {
        Vec16<T> a, b;
        a.load(dst + i * VE);
        b.load(src + i * VE);
        float fa[VE], fb[VE];
        a.unpack(fa);
        b.unpack(fb);
#pragma unroll
        for (int j = 0; j < VE; ++j) fa[j] += fb[j];
        a.pack(fa);
        a.store(dst + i * VE);
    }
}

int blocks_for(long long total) {
    long long b = (total + kThreads - 1) / kThreads;
    if (b > 8192) b = 8192;
    return (int)(b < 1 ? 1 : b);
}

int check(int N, int h, int w, int C, int dtype, int* ve) {
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "resize: dtype %d", dtype);
    *ve = dtype == MPN_F32 ? 4 : 8;
    MPN_REQUIRE(N > 0 && h > 0 && w > 0 && C > 0 && C % *ve == 0, MPN_ERR_BAD_SHAPE, "resize: bad shape / C %% %d", *ve);
    return MPN_OK;
}
}  // namespace

extern "C" int mpn_bilinear_up_fwd(const void* x, void* y, int N, int h, int w, int C, int upsample, int y_channel_offset,
                                   int y_channels_total, int dtype, const float* in_scale, const float* in_shift, int in_act,
                                   mpn_stream_t stream) {
    int ve;
    if (int rc = check(N, h, w, C, dtype, &ve)) return rc;
    MPN_REQUIRE(x && y, MPN_ERR_BAD_ARG, "bilinear_fwd: null pointer");
    MPN_REQUIRE(upsample >= 1 && upsample <= 64, MPN_ERR_BAD_SHAPE, "bilinear: integer upsample factor expected");
    MPN_REQUIRE(y_channel_offset % ve == 0 && y_channels_total % ve == 0 && y_channel_offset + C <= y_channels_total,
                MPN_ERR_BAD_SHAPE, "bilinear: bad channel slice");
    MPN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), MPN_ERR_BAD_ARG, "bilinear: scale/shift mismatch");
    const long long total = (long long)N * h * upsample * w * upsample * (C / ve);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (bilinear_up_fwd_kernel<T><<<blocks_for(total), kThreads, 0, st>>>(
                                  (const T*)x, (T*)y, N, h, w, C, upsample, y_channel_offset, y_channels_total, in_scale,
                                  in_shift, in_act, total)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_bilinear_up_bwd(const void* dy, void* dx, int N, int h, int w, int C, int upsample, int y_channel_offset,
                                   int y_channels_total, int dtype, mpn_stream_t stream) {
    int ve;
    if (int rc = check(N, h, w, C, dtype, &ve)) return rc;
    MPN_REQUIRE(dy && dx, MPN_ERR_BAD_ARG, "bilinear_bwd: null pointer");
    MPN_REQUIRE(upsample >= 1 && upsample <= 64, MPN_ERR_BAD_SHAPE, "bilinear: integer upsample factor expected");
    MPN_REQUIRE(y_channel_offset % ve == 0 && y_channels_total % ve == 0 && y_channel_offset + C <= y_channels_total,
                MPN_ERR_BAD_SHAPE, "bilinear: bad channel slice");
    const long long total = (long long)N * h * w * (C / ve);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (bilinear_up_bwd_kernel<T><<<blocks_for(total), kThreads, 0, st>>>(
                                  (const T*)dy, (T*)dx, N, h, w, C, upsample, y_channel_offset, y_channels_total, total)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* src [N,2h,2w,C] -> dst [N,h,w,C]; accumulate != 0 adds to dst */
extern "C" int mpn_sumpool2x2(const void* src, void* dst, int N, int h, int w, int C, int accumulate, int dtype,
                              mpn_stream_t stream) {
    int ve;
    if (int rc = check(N, h, w, C, dtype, &ve)) return rc;
    MPN_REQUIRE(src && dst, MPN_ERR_BAD_ARG, "sumpool: null pointer");
    const long long total = (long long)N * h * w * (C / ve);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (sumpool2x2_kernel<T><<<blocks_for(total), kThreads, 0, st>>>((const T*)src, (T*)dst, N, h, w,
                                                                                          C, accumulate, total)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* dst += src, n elements (multiple of 16 bytes) */
extern "C" int mpn_add_inplace(void* dst, const void* src, long long n, int dtype, mpn_stream_t stream) {
    MPN_REQUIRE(dst && src && n > 0, MPN_ERR_BAD_ARG, "add_inplace: bad arguments");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "add_inplace: dtype %d", dtype);
    const int ve = dtype == MPN_F32 ? 4 : 8;
    MPN_REQUIRE(n % ve == 0, MPN_ERR_BAD_SHAPE, "add_inplace: n must be a multiple of %d", ve);
    const long long nvec = n / ve;
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (add_inplace_kernel<T><<<blocks_for(nvec), kThreads, 0, st>>>((T*)dst, (const T*)src, nvec)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
