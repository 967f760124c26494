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

// One block row of the grid = one INPUT row (n, y0): a thread owns one 16-byte channel vector of one input pixel, reads
// the 2x2 input neighbourhood once (batch-norm affine + activation applied once per tap, scale / shift loaded once as
// 16-byte vectors) and writes the u x u output pixels that interpolate inside it - 8 + 16/u lerps per output vector
// instead of 4 activated taps each. The arithmetic per output is unchanged: top = a+(b-a)fx, bot = c+(d-c)fx,
// out = top+(bot-top)fy (TF's order, resize_bilinear_op.cc), so results are bit-identical to the per-output form.
template <typename T>
__global__ __launch_bounds__(kThreads) void bilinear_up_fwd_kernel(
    const T* __restrict__ x, T* __restrict__ y, int h, int w, int C, int u, int y_coff, int y_ctot,
    const float* __restrict__ sc, const float* __restrict__ sh, int act) {
    constexpr int VE = Vec16<T>::N;
    const unsigned cvec = C / VE;
    const int OW = w * u;
    const unsigned t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= (unsigned)w * cvec) return;
    const int x0 = (int)(t / cvec), c0 = (int)(t - (unsigned)x0 * cvec) * VE;
    const int n = blockIdx.y / h, y0 = blockIdx.y - n * h;          // scalar
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float inv = 1.0f / (float)u;
    float s_[VE], b_[VE];
    const bool aff = sc != nullptr;
    // unconditional 16-byte loads under ONE branch: a per-element `aff ? sc[c] : 1.f` compiles to 2*VE predicated dword
    // loads with a 32-byte lane stride, and the texture unit then bounds the kernel (232 us vs 46 us per level)
    if (aff) {
#pragma unroll
        for (int j = 0; j < VE; j += 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(sc + c0 + j);
            const float4 b4 = *reinterpret_cast<const float4*>(sh + c0 + j);
            s_[j] = a4.x; s_[j + 1] = a4.y; s_[j + 2] = a4.z; s_[j + 3] = a4.w;
            b_[j] = b4.x; b_[j + 1] = b4.y; b_[j + 2] = b4.z; b_[j + 3] = b4.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < VE; ++j) { s_[j] = 1.f; b_[j] = 0.f; }
    }
    const float lo = (aff && act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (aff && act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    float a[VE], b[VE], c[VE], d[VE];
#define MPN_TAP(yy, xx, f)                                                                        \
    do {                                                                                          \
        Vec16<T> v_;                                                                              \
        v_.load(x + (((long long)n * h + (yy)) * w + (xx)) * C + c0);                             \
        v_.unpack(f);                                                                             \
        _Pragma("unroll") for (int j = 0; j < VE; ++j)                                            \
            f[j] = __builtin_amdgcn_fmed3f(f[j] * s_[j] + b_[j], lo, hi);                         \
    } while (0)
    MPN_TAP(y0, x0, a);
    MPN_TAP(y0, x1, b);
    MPN_TAP(y1, x0, c);
    MPN_TAP(y1, x1, d);
#undef MPN_TAP
    T* out = y + (((long long)n * h * u + (long long)y0 * u) * OW + (long long)x0 * u) * y_ctot + y_coff + c0;
    for (int k = 0; k < u; ++k) {
        const float fx = (float)k * inv;
        float top[VE], bot[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            top[j] = a[j] + (b[j] - a[j]) * fx;
            bot[j] = c[j] + (d[j] - c[j]) * fx;
        }
        for (int m = 0; m < u; ++m) {
            const float fy = (float)m * inv;
            float o[VE];
#pragma unroll
            for (int j = 0; j < VE; ++j) o[j] = top[j] + (bot[j] - top[j]) * fy;
            Vec16<T> ov;
            ov.pack(o);
            ov.store(out + ((long long)m * OW + k) * y_ctot);
        }
    }
}

// dX[n,iy,ix,c] = sum over output pixels of wy*wx*dY (gather form, deterministic). The weight of output row oy for
// input row iy is the tent 1 - |oy - iy*u| / u over |oy - iy*u| < u, except that the last input row also owns the
// clamped rows oy >= (h-1)*u with weight 1 (y0 == y1 == h-1 there); same in x. No divisions in the loops.
template <typename T, int U>
__global__ __launch_bounds__(kThreads) void bilinear_up_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int h,
                                                                   int w, int C, int y_coff, int y_ctot) {
    constexpr int VE = Vec16<T>::N;
    constexpr int u = U;
    constexpr int NT = 2 * U - 1;                                    // taps per axis
    const unsigned cvec = C / VE;
    const int OH = h * u, OW = w * u;
    const unsigned t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= (unsigned)w * cvec) return;
    const int ix = (int)(t / cvec), c0 = (int)(t - (unsigned)ix * cvec) * VE;
    const int n = blockIdx.y / h, iy = blockIdx.y - n * h;          // scalar
    constexpr float inv = 1.0f / (float)U;
    float acc[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) acc[j] = 0.f;
    const bool last_y = iy == h - 1, last_x = ix == w - 1;
    // the x taps of this thread: weights and (clamped) offsets once, so that the row loop is NT independent loads
    float wxs[NT];
    int xoffs[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const int dxo = k - (U - 1);
        const bool ok = dxo >= 0 || ix > 0;
        const float fx = (float)(dxo >= 0 ? dxo : u + dxo) * inv;
        wxs[k] = !ok ? 0.f : (dxo < 0 ? fx : (last_x ? (1.f - fx) + fx : 1.f - fx));
        xoffs[k] = (ix * u + (ok ? dxo : 0)) * y_ctot;
    }
    const int dy_lo = iy == 0 ? 0 : -(u - 1);
    const T* base = dy + (long long)n * OH * OW * y_ctot + y_coff + c0;
    for (int dyo = dy_lo; dyo < u; ++dyo) {
        const int oy = iy * u + dyo;
        const float fy = (float)(dyo >= 0 ? dyo : u + dyo) * inv;   // forward's fraction of output row oy
        const float wy = dyo < 0 ? fy : (last_y ? (1.f - fy) + fy : 1.f - fy);
        const T* row = base + (long long)oy * OW * y_ctot;
        Vec16<T> v[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) v[k].load(row + xoffs[k]);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            float f[VE];
            v[k].unpack(f);
            const float wgt = wy * wxs[k];
#pragma unroll
            for (int j = 0; j < VE; ++j) acc[j] += wgt * f[j];
        }
    }
    Vec16<T> ov;
    ov.pack(acc);
    ov.store(dx + (((long long)n * h + iy) * w + ix) * C + c0);
}

// The same sums with every element of dY loaded ONCE by ONE thread (the gather above loads it (2U-1)^2 / U^2 = 2.3-3.5 times
// and walks 2U-1 dependent rows per thread: 37-48 us per level of the subnet where the bytes take 27). The tent is separable:
// a thread owns up to IT (output column, 16-byte channel vector) items of one channel group and walks DOWN the output rows of
// a strip of R input rows. Output row oy = p*U + k feeds input row p with 1 - k/U and input row min(p + 1, h - 1) with k/U,
// so two running sums per item (the input row being finished, the next one) take every row exactly once; at each finished
// input row the column sums go to LDS and w * cvn threads combine 2U-1 of them with the x tent. A strip re-reads only the
// U-1 rows above its first input row. Sums are f32 in a fixed order (rows, then columns): deterministic.
template <typename T, int U>
__global__ __launch_bounds__(kThreads) void bilinear_up_bwd_walk_kernel(const T* __restrict__ dy, T* __restrict__ dx, int h,
                                                                        int w, int C, int y_coff, int y_ctot, int R,
                                                                        int cvb, int ncg, int nstrips) {
    constexpr int VE = Vec16<T>::N;
    constexpr int IT = 4;                                            // items per thread (the launcher keeps OW * cvb <= IT * kThreads)
    constexpr int G = 2;                                             // rows in flight per thread (8 vectors: 32 KB per block; 4 rows: equal or slower)
    constexpr int NT = 2 * U - 1;
    constexpr float inv = 1.0f / (float)U;
    extern __shared__ __attribute__((aligned(16))) float colsum[];   // [OW * cvn][VE]
    // XCD-aware block -> work map (dwconv.hip): the blocks of one XCD take neighbouring strips, which share U-1 rows
    int wid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wid & 7;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wid >> 3);
    }
    const int cg = wid % ncg;
    const int strip = (wid / ncg) % nstrips;
    const int n = wid / (ncg * nstrips);
    const int cvec = C / VE;
    const int cv0 = cg * cvb, cvn = min(cvb, cvec - cv0);
    const int OW = w * U, OH = h * U;
    const int items = OW * cvn;
    const int iy0 = strip * R, iy1 = min(h, iy0 + R);
    const int tid = threadIdx.x;

    int off[IT];                                                     // element offset of the item inside an output row
    bool valid[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int i = tid + j * kThreads;
        valid[j] = i < items;
        const int ii = valid[j] ? i : 0;
        const int col = ii / cvn, cv = ii - col * cvn;
        off[j] = col * y_ctot + y_coff + (cv0 + cv) * VE;
    }
    const T* base = dy + (long long)n * OH * OW * y_ctot;
    float cur[IT][VE], nxt[IT][VE];
#pragma unroll
    for (int j = 0; j < IT; ++j)
#pragma unroll
        for (int e = 0; e < VE; ++e) nxt[j][e] = 0.f;

    // The walk as ONE flat sequence of groups of <= G rows that never straddle an input row: the loads of group g + 1 are
    // issued right after the sums of group g, BEFORE the LDS exchange that ends an input row, so they fly under it.
    Vec16<T> v[G][IT];
    auto issue = [&](int p, int k0, int k1) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (k0 + g < k1) {
                const T* row = base + (long long)(p * U + k0 + g) * OW * y_ctot;
#pragma unroll
                for (int j = 0; j < IT; ++j) v[g][j].load(row + off[j]);
            }
        }
    };
    // rows k0 .. k1-1 of the U output rows under input row p: cur += wc * row (rows of the strip only), nxt += wn * row
    auto sums = [&](int k0, int k1, bool into_cur, bool last_row) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int k = k0 + g;
            if (k < k1) {
                const float fy = (float)k * inv;
                const float wc = !into_cur ? 0.f : (last_row ? (1.f - fy) + fy : 1.f - fy);
                const float wn = last_row ? 0.f : fy;
#pragma unroll
                for (int j = 0; j < IT; ++j) {
                    float f[VE];
                    v[g][j].unpack(f);
#pragma unroll
                    for (int e = 0; e < VE; ++e) {
                        cur[j][e] += wc * f[e];
                        nxt[j][e] += wn * f[e];
                    }
                }
            }
        }
    };
#pragma unroll
    for (int j = 0; j < IT; ++j)
#pragma unroll
        for (int e = 0; e < VE; ++e) cur[j][e] = 0.f;

    int p = iy0 > 0 ? iy0 - 1 : iy0, k = iy0 > 0 ? 1 : 0;           // the U-1 rows above the strip feed its first input row
    issue(p, k, min(k + G, U));
#pragma unroll 1
    for (;;) {
        const int k1 = min(k + G, U);
        const int np = k1 == U ? p + 1 : p, nk = k1 == U ? 0 : k1;
        const bool more = np < iy1;
        sums(k, k1, p >= iy0, p == h - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(np, nk, min(nk + G, U));
        if (k1 == U) {
            if (p >= iy0) {
                const int iy = p;
#pragma unroll
                for (int j = 0; j < IT; ++j) {
                    if (valid[j]) {
                        float* s = colsum + (size_t)(tid + j * kThreads) * VE;
#pragma unroll
                        for (int e = 0; e < VE; e += 4)
                            *reinterpret_cast<float4*>(s + e) = make_float4(cur[j][e], cur[j][e + 1], cur[j][e + 2], cur[j][e + 3]);
                    }
                }
                __syncthreads();
                for (int o = tid; o < w * cvn; o += kThreads) {
                    const int ix = o / cvn, cv = o - ix * cvn;
                    const bool last_x = ix == w - 1;
                    float acc[VE];
#pragma unroll
                    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
#pragma unroll 3
                    for (int tk = 0; tk < NT; ++tk) {
                        const int dxo = tk - (U - 1);
                        const bool ok = dxo >= 0 || ix > 0;
                        const float fx = (float)(dxo >= 0 ? dxo : U + dxo) * inv;
                        const float wx = !ok ? 0.f : (dxo < 0 ? fx : (last_x ? (1.f - fx) + fx : 1.f - fx));
                        const float* s = colsum + (size_t)((ix * U + (ok ? dxo : 0)) * cvn + cv) * VE;
#pragma unroll
                        for (int e = 0; e < VE; e += 4) {
                            const float4 q = *reinterpret_cast<const float4*>(s + e);
                            acc[e] += wx * q.x; acc[e + 1] += wx * q.y; acc[e + 2] += wx * q.z; acc[e + 3] += wx * q.w;
                        }
                    }
                    Vec16<T> ov;
                    ov.pack(acc);
                    ov.store(dx + (((long long)n * h + iy) * w + ix) * C + (cv0 + cv) * VE);
                }
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < IT; ++j)
#pragma unroll
                for (int e = 0; e < VE; ++e) { cur[j][e] = nxt[j][e]; nxt[j][e] = 0.f; }
        }
        if (!more) break;
        p = np; k = nk;
    }
}

// upsample 1 forward: act(x * scale + shift) written into the concat slice (the general kernel loads all four taps of
// every output although fx = fy = 0), 4 vectors in flight per thread
template <typename T>
__global__ __launch_bounds__(kThreads) void slice_affine_store_kernel(const T* __restrict__ x, T* __restrict__ y, long long nvec,
                                                                      int C, int coff, int ctot, const float* __restrict__ sc,
                                                                      const float* __restrict__ sh, int act) {
    constexpr int VE = Vec16<T>::N;
    constexpr int U = 4;
    const unsigned cvec = C / VE;
    const long long stride = (long long)gridDim.x * kThreads;   // a multiple of cvec (checked by the launcher)
    const long long i0 = (long long)blockIdx.x * kThreads + threadIdx.x;
    const int c0 = (int)(i0 % cvec) * VE;
    float s_[VE], b_[VE];
    const bool aff = sc != nullptr;
    if (aff) {
#pragma unroll
        for (int j = 0; j < VE; j += 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(sc + c0 + j);
            const float4 b4 = *reinterpret_cast<const float4*>(sh + c0 + j);
            s_[j] = a4.x; s_[j + 1] = a4.y; s_[j + 2] = a4.z; s_[j + 3] = a4.w;
            b_[j] = b4.x; b_[j + 1] = b4.y; b_[j + 2] = b4.z; b_[j + 3] = b4.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < VE; ++j) { s_[j] = 1.f; b_[j] = 0.f; }
    }
    const float lo = (aff && act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (aff && act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    for (long long i = i0; i < nvec; i += U * stride) {
        Vec16<T> v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k].load(x + (i + k * stride < nvec ? i + k * stride : i) * VE);
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long long ii = i + k * stride;
            if (ii < nvec) {
                float f[VE];
                v[k].unpack(f);
#pragma unroll
                for (int j = 0; j < VE; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * s_[j] + b_[j], lo, hi);
                v[k].pack(f);
                v[k].store(y + (ii / cvec) * ctot + coff + c0);
            }
        }
    }
}

// upsample 1: the gradient of a concat slice is the slice itself - a strided copy, 4 vectors in flight per thread
// (the tent kernel above ran it as 32 768 blocks of one load + one store per thread: 2.2 TB/s)
template <typename T>
__global__ __launch_bounds__(kThreads) void slice_copy_kernel(const T* __restrict__ src, T* __restrict__ dst, long long nvec,
                                                              int C, int coff, int ctot) {
    constexpr int VE = Vec16<T>::N;
    constexpr int U = 4;
    const unsigned cvec = C / VE;
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += U * stride) {
        Vec16<T> v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long long ii = i + k * stride < nvec ? i + k * stride : i;
            const long long pix = ii / cvec;
            v[k].load(src + pix * ctot + coff + (ii - pix * cvec) * VE);
        }
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (i + k * stride < nvec) v[k].store(dst + (i + k * stride) * VE);
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
    MPN_REQUIRE(mpn_aligned16(in_scale) && mpn_aligned16(in_shift), MPN_ERR_BAD_ALIGN, "bilinear: scale/shift must be 16-byte aligned");
    MPN_REQUIRE((long long)N * h <= 65535, MPN_ERR_BAD_SHAPE, "bilinear: N * height must be <= 65535");
    const dim3 grid((unsigned)mpn_div_up((long long)w * (C / ve), kThreads), (unsigned)(N * h));
    hipStream_t st = (hipStream_t)stream;
    if (upsample == 1 && kThreads % (C / ve) == 0) {
        const long long nvec = (long long)N * h * w * (C / ve);
        long long blocks = mpn_div_up(nvec, 4 * kThreads);
        if (blocks > 4096) blocks = 4096;
        MPN_DISPATCH_DTYPE(dtype, (slice_affine_store_kernel<T><<<(unsigned)blocks, kThreads, 0, st>>>(
                                      (const T*)x, (T*)y, nvec, C, y_channel_offset, y_channels_total, in_scale, in_shift, in_act)));
        MPN_LAUNCH_CHECK();
        return MPN_OK;
    }
    MPN_DISPATCH_DTYPE(dtype, (bilinear_up_fwd_kernel<T><<<grid, kThreads, 0, st>>>(
                                  (const T*)x, (T*)y, h, w, C, upsample, y_channel_offset, y_channels_total, in_scale,
                                  in_shift, in_act)));
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
    MPN_REQUIRE((long long)N * h <= 65535, MPN_ERR_BAD_SHAPE, "bilinear: N * height must be <= 65535");
    const dim3 grid((unsigned)mpn_div_up((long long)w * (C / ve), kThreads), (unsigned)(N * h));
    hipStream_t st = (hipStream_t)stream;
    if (upsample == 1) {
        const long long nvec = (long long)N * h * w * (C / ve);
        long long blocks = mpn_div_up(nvec, 4 * kThreads);
        if (blocks > 4096) blocks = 4096;
        MPN_DISPATCH_DTYPE(dtype, (slice_copy_kernel<T><<<(unsigned)blocks, kThreads, 0, st>>>(
                                      (const T*)dy, (T*)dx, nvec, C, y_channel_offset, y_channels_total)));
        MPN_LAUNCH_CHECK();
        return MPN_OK;
    }
    MPN_REQUIRE(upsample == 2 || upsample == 4 || upsample == 8, MPN_ERR_BAD_SHAPE, "bilinear bwd: upsample must be 1, 2, 4 or 8");
    const int OW = w * upsample, cvec = C / ve;
    if (upsample >= 4 && OW <= 4 * kThreads) {
        // the walk: channel groups of cvb vectors so that one output row of a group is at most 4 items per thread; strips of
        // R input rows so that the launch has >= 512 blocks (each strip re-reads upsample - 1 rows; 256 / 1024 blocks are
        // slower). On the subnet's levels, [32,128,128,128 of 512] cold: 32.6-33.5 us (x4), 33.6-34.2 (x8) against the
        // gather's 40.6-40.9; x2 stays on the gather (36.6 us there, 41-44 here: an LDS exchange every two rows).
        int cvb = 4 * kThreads / OW;
        if (cvb > cvec) cvb = cvec;
        const int ncg = mpn_div_up(cvec, cvb);
        int nstrips = mpn_div_up(512, (long long)N * ncg);
        if (nstrips > h) nstrips = h;
        const int R = mpn_div_up(h, nstrips);
        nstrips = mpn_div_up(h, R);
        const long long blocks = (long long)N * ncg * nstrips;
        MPN_REQUIRE(blocks < (1ll << 31), MPN_ERR_BAD_SHAPE, "bilinear bwd: too many blocks");
        const size_t lds = (size_t)OW * cvb * ve * sizeof(float);
        MPN_DISPATCH_DTYPE(dtype, {
            if (upsample == 2) bilinear_up_bwd_walk_kernel<T, 2><<<(unsigned)blocks, kThreads, lds, st>>>((const T*)dy, (T*)dx, h, w, C, y_channel_offset, y_channels_total, R, cvb, ncg, nstrips);
            else if (upsample == 4) bilinear_up_bwd_walk_kernel<T, 4><<<(unsigned)blocks, kThreads, lds, st>>>((const T*)dy, (T*)dx, h, w, C, y_channel_offset, y_channels_total, R, cvb, ncg, nstrips);
            else bilinear_up_bwd_walk_kernel<T, 8><<<(unsigned)blocks, kThreads, lds, st>>>((const T*)dy, (T*)dx, h, w, C, y_channel_offset, y_channels_total, R, cvb, ncg, nstrips);
        });
        MPN_LAUNCH_CHECK();
        return MPN_OK;
    }
    MPN_DISPATCH_DTYPE(dtype, {
        if (upsample == 2) bilinear_up_bwd_kernel<T, 2><<<grid, kThreads, 0, st>>>((const T*)dy, (T*)dx, h, w, C, y_channel_offset, y_channels_total);
        else if (upsample == 4) bilinear_up_bwd_kernel<T, 4><<<grid, kThreads, 0, st>>>((const T*)dy, (T*)dx, h, w, C, y_channel_offset, y_channels_total);
        else bilinear_up_bwd_kernel<T, 8><<<grid, kThreads, 0, st>>>((const T*)dy, (T*)dx, h, w, C, y_channel_offset, y_channels_total);
    });
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
