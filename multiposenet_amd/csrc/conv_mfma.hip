// K5/K6/K8: dense 1x1 and 3x3 (stride 1, pad 1) convolutions as implicit GEMM on the MFMA
// matrix cores, NHWC.  Replaces slim.conv2d 1x1 (mobilenet_v1.py:73), conv2d_same k=1/k=3
// (layer_utils.py:19-39 as used at fpn.py:38,39,50,52 and keypoint_subnet.py:38,75,77) and, with
// transposed/flipped packed weights, their data-gradients.
//
//   out[m, co] = sum_{tap, ci} act(bn(x))[pixel(m)+tap, ci] * W[tap, ci, co]
//
// Block = 256 threads (4 waves, 2x2), output tile 128 pixels x BN channels; for 3x3 the 128
// pixels are an 8x16 spatial patch whose 10x18 input halo is staged ONCE per channel chunk in
// LDS (batch-norm affine + ReLU/ReLU6 of the producer applied on the way in, zero padding
// injected here) and then serves all 9 taps: A fragments are read straight from the halo image
// with a tap offset.  Only the weights stream: they are pre-packed (mpn_conv_pack_weights) in
// exactly the LDS image order, so a 2-k-step stage is one contiguous 16 KB copy, double
// buffered.  LDS pixel rows are 256 B (or 128 B) = one bank row; 16-byte slots are
// XOR-swizzled with the pixel index so the 16 rows of a ds_read_b128 fragment hit 16 slots.
// bf16 uses v_mfma_f32_16x16x32_bf16; the f32 parity build uses 4x v_mfma_f32_16x16x4_f32 on
// the same 16-byte fragments (k permuted identically in A and B).
// Epilogue: accumulators -> LDS -> full 16-byte NHWC stores; optional fused nearest-2x
// upsample-add (fpn.py:51) and per-tile batch-norm partial sums (sum, sum of squares).
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    using Frag = bf16x8_t;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    using Frag = f32x4_t;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x4_t& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    }
};

struct ConvParams {
    const void* x;
    const void* wp;
    void* y;
    const float* in_scale;
    const float* in_shift;
    int in_act;
    float* stats_part;
    const void* up_res;
    int N, H, W, Cin, Cout;
    int tiles_x, tiles_y;
    long long M;
    int row_bytes;  // 128 or 256
    int nchunk;
    int n_tiles;
    long long wp_tile_bytes;  // packed bytes per n-tile
};

constexpr int kThreads = 256;
constexpr int kHaloW = 18, kHaloH = 10;

template <typename T>
__device__ __forceinline__ void apply_affine_act(Vec16<T>& v, const float* sc, const float* sh, int act) {
    constexpr int VE = Vec16<T>::N;
    float f[VE];
    v.unpack(f);
#pragma unroll
    for (int j = 0; j < VE; ++j) {
        float t = f[j] * sc[j] + sh[j];
        if (act != MPN_ACT_NONE) t = fmaxf(t, 0.f);
        if (act == MPN_ACT_RELU6) t = fminf(t, 6.f);
        f[j] = t;
    }
    v.pack(f);
}

template <typename T, int TAPS, int BN>
__global__ __launch_bounds__(kThreads, 2) void conv_mfma_kernel(const ConvParams p) {
    constexpr int ES = (int)sizeof(T);
    constexpr int VE = 16 / ES;
    constexpr int CCE = 256 / ES;  // channels per full chunk
    constexpr int NPIX = TAPS == 9 ? kHaloW * kHaloH : 128;
    constexpr int NT = BN / 32;                 // 16-col tiles per wave
    constexpr int STAGE_BYTES = 2 * BN * 64;    // two k-steps of weights
    constexpr int BVEC = STAGE_BYTES / (kThreads * 16);
    constexpr int AVEC = (NPIX * 16 + kThreads - 1) / kThreads;
    using Frag = typename Mma<T>::Frag;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Bs = smem + NPIX * 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;

    const int ntile = blockIdx.x % p.n_tiles;
    const int mtile = blockIdx.x / p.n_tiles;
    const int n0 = ntile * BN;

    // ---- tile coordinates
    int img = 0, oy0 = 0, ox0 = 0;
    long long m0 = 0;
    if (TAPS == 9) {
        const int tx = mtile % p.tiles_x;
        const int t2 = mtile / p.tiles_x;
        const int ty = t2 % p.tiles_y;
        img = t2 / p.tiles_y;
        oy0 = ty * 8;
        ox0 = tx * 16;
    } else {
        m0 = (long long)mtile * 128;
    }

    const int RB = p.row_bytes;
    const int slots = RB >> 4;              // 8 or 16
    const int swz_shift = (slots == 16) ? 0 : 1;
    const int ksteps = RB >> 6;             // 64-byte k-steps per chunk (2 or 4)
    const int stages_per_tap = RB >> 7;     // 1 or 2
    const int total_stages = p.nchunk * TAPS * stages_per_tap;

    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const unsigned char* __restrict__ wsrc =
        reinterpret_cast<const unsigned char*>(p.wp) + (long long)ntile * p.wp_tile_bytes;

    // ---- accumulators
    f32x4_t acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // per-lane A row -> halo pixel base (without tap offset)
    int pbase[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = wm * 64 + mt * 16 + l15;
        pbase[mt] = (TAPS == 9) ? ((row >> 4) * kHaloW + (row & 15)) : row;
    }
    const int b_lane_off = (wn * (BN / 2) + l15) * 64 + lq * 16;

    // ---- weight stage 0 -> buffer 0
    // (named registers, not an array: hipcc left a conditionally written uint4[] in scratch)
    static_assert(BVEC == 2 || BVEC == 4, "weight stage is 2 or 4 vectors per thread");
    uint4 br0, br1, br2, br3;
#define MPN_BLOAD(src)                                                                    \
    do {                                                                                  \
        const unsigned char* s_ = (src) + (size_t)tid * 16;                               \
        br0 = *reinterpret_cast<const uint4*>(s_);                                        \
        br1 = *reinterpret_cast<const uint4*>(s_ + kThreads * 16);                        \
        if (BVEC == 4) {                                                                  \
            br2 = *reinterpret_cast<const uint4*>(s_ + 2 * kThreads * 16);                \
            br3 = *reinterpret_cast<const uint4*>(s_ + 3 * kThreads * 16);                \
        }                                                                                 \
    } while (0)
#define MPN_BSTORE(dst)                                                                   \
    do {                                                                                  \
        unsigned char* d_ = (dst) + (size_t)tid * 16;                                     \
        *reinterpret_cast<uint4*>(d_) = br0;                                              \
        *reinterpret_cast<uint4*>(d_ + kThreads * 16) = br1;                              \
        if (BVEC == 4) {                                                                  \
            *reinterpret_cast<uint4*>(d_ + 2 * kThreads * 16) = br2;                      \
            *reinterpret_cast<uint4*>(d_ + 3 * kThreads * 16) = br3;                      \
        }                                                                                 \
    } while (0)
    br2 = br3 = make_uint4(0u, 0u, 0u, 0u);
    MPN_BLOAD(wsrc);
    MPN_BSTORE(Bs);

    int s = 0;
    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        __syncthreads();  // every wave has finished reading the previous A image
        // ================= stage the A image (halo or flat rows) for this channel chunk
        {
            const int slot = tid & (slots - 1);  // kThreads % slots == 0 -> fixed per thread
            const int ce = chunk * CCE + slot * VE;
            const bool cvalid = ce < p.Cin;
            float sc[VE], sh[VE];
            const bool affine = (p.in_scale != nullptr);
            if (affine && cvalid) {
#pragma unroll
                for (int j = 0; j < VE; ++j) { sc[j] = p.in_scale[ce + j]; sh[j] = p.in_shift[ce + j]; }
            } else {
#pragma unroll
                for (int j = 0; j < VE; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
            }
            const int nvec = NPIX * slots;
            const int pix_step = kThreads / slots;
            Vec16<T> v[AVEC];
            bool inb[AVEC];
#pragma unroll
            for (int i = 0; i < AVEC; ++i) {
                const int vi = tid + i * kThreads;
                const int pix = (tid / slots) + i * pix_step;
                bool ok = cvalid && (vi < nvec);
                long long off = 0;
                if (TAPS == 9) {
                    const int hy = pix / kHaloW, hx = pix - hy * kHaloW;
                    const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                    ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    off = (((long long)img * p.H + iy) * p.W + ix) * p.Cin + ce;
                } else {
                    const long long m = m0 + pix;
                    ok = ok && (m < p.M);
                    off = m * p.Cin + ce;
                }
                inb[i] = ok;
                if (ok) v[i].load(x + off); else v[i].zero();
            }
#pragma unroll
            for (int i = 0; i < AVEC; ++i) {
                const int vi = tid + i * kThreads;
                if (vi < nvec) {
                    const int pix = (tid / slots) + i * pix_step;
                    if (affine && inb[i]) apply_affine_act<T>(v[i], sc, sh, p.in_act);
                    const int sslot = slot ^ ((pix >> swz_shift) & (slots - 1));
                    *reinterpret_cast<uint4*>(As + pix * RB + (sslot << 4)) =
                        *reinterpret_cast<const uint4*>(&v[i].raw);
                }
            }
        }
        __syncthreads();

        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = (TAPS == 9) ? ((tap / 3) * kHaloW + (tap % 3)) : 0;
            for (int h = 0; h < stages_per_tap; ++h) {
                const bool more = (s + 1 < total_stages);
                if (more) MPN_BLOAD(wsrc + (size_t)(s + 1) * STAGE_BYTES);
                const unsigned char* Bb = Bs + (s & 1) * STAGE_BYTES;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int kstep = h * 2 + k2;
                    if (kstep < ksteps) {
                        Frag a[4], b[NT];
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            const int pix = pbase[mt] + toff;
                            const int slot = kstep * 4 + lq;
                            const int sslot = slot ^ ((pix >> swz_shift) & (slots - 1));
                            a[mt] = *reinterpret_cast<const Frag*>(As + pix * RB + (sslot << 4));
                        }
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            b[nt] = *reinterpret_cast<const Frag*>(Bb + k2 * (BN * 64) + nt * 1024 + b_lane_off);
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) Mma<T>::run(a[mt], b[nt], acc[mt][nt]);
                    }
                }
                if (more) MPN_BSTORE(Bs + ((s + 1) & 1) * STAGE_BYTES);
                __syncthreads();
                ++s;
            }
        }
    }

#undef MPN_BLOAD
#undef MPN_BSTORE
    // ================= epilogue: accumulators -> LDS tile [128][BN] f32 (row stride padded)
    constexpr int OST = BN * 4 + 16;  // bytes per row
    unsigned char* Os = smem;         // all MFMA-phase LDS reads are behind the last barrier
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 64 + mt * 16 + lq * 4 + r;
                const int col = wn * (BN / 2) + nt * 16 + l15;
                *reinterpret_cast<float*>(Os + row * OST + col * 4) = acc[mt][nt][r];
            }
    __syncthreads();

    constexpr int VPR = BN / VE;               // 16-byte vectors per output row
    constexpr int ROWS_PER_PASS = kThreads / VPR;
    const int vcol = tid % VPR;
    const int rrow = tid / VPR;
    T* __restrict__ y = reinterpret_cast<T*>(p.y);
    const T* __restrict__ res = reinterpret_cast<const T*>(p.up_res);
    float ssum[VE], ssq[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
    const bool col_ok = (n0 + vcol * VE) < p.Cout;

    for (int row = rrow; row < 128; row += ROWS_PER_PASS) {
        bool ok = col_ok;
        long long pixel;  // flat NHW index
        int n_i, oy, ox;
        if (TAPS == 9) {
            oy = oy0 + (row >> 4);
            ox = ox0 + (row & 15);
            n_i = img;
            ok = ok && oy < p.H && ox < p.W;
            pixel = ((long long)img * p.H + oy) * p.W + ox;
        } else {
            pixel = m0 + row;
            ok = ok && pixel < p.M;
            ox = (int)(pixel % p.W);
            const long long t = pixel / p.W;
            oy = (int)(t % p.H);
            n_i = (int)(t / p.H);
        }
        if (!ok) continue;
        float f[VE];
        const float* src = reinterpret_cast<const float*>(Os + row * OST) + vcol * VE;
#pragma unroll
        for (int j = 0; j < VE; j += 4) {
            const float4 q = *reinterpret_cast<const float4*>(src + j);
            f[j] = q.x; f[j + 1] = q.y; f[j + 2] = q.z; f[j + 3] = q.w;
        }
        if (res != nullptr) {
            const int h2 = p.H >> 1, w2 = p.W >> 1;
            const long long roff = (((long long)n_i * h2 + (oy >> 1)) * w2 + (ox >> 1)) * p.Cout + n0 + vcol * VE;
            Vec16<T> rv;
            rv.load(res + roff);
            float g[VE];
            rv.unpack(g);
#pragma unroll
            for (int j = 0; j < VE; ++j) f[j] += g[j];
        }
#pragma unroll
        for (int j = 0; j < VE; ++j) { ssum[j] += f[j]; ssq[j] += f[j] * f[j]; }
        Vec16<T> ov;
        ov.pack(f);
        ov.store(y + pixel * p.Cout + n0 + vcol * VE);
    }

    if (p.stats_part != nullptr) {
        __syncthreads();  // done reading the output tile; reuse LDS for the reduction
        float* red = reinterpret_cast<float*>(smem);  // [ROWS_PER_PASS][2][BN]
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            red[(rrow * 2 + 0) * BN + vcol * VE + j] = ssum[j];
            red[(rrow * 2 + 1) * BN + vcol * VE + j] = ssq[j];
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;
            float t = 0.f;
            for (int r = 0; r < ROWS_PER_PASS; ++r) t += red[(r * 2 + which) * BN + c];
            if (n0 + c < p.Cout)
                p.stats_part[((long long)mtile * 2 + which) * p.Cout + n0 + c] = t;
        }
    }
}

// ------------------------------------------------------------------ weight packing
// Packed order (per n-tile of BN output channels): [chunk][tap][stage][kstep(2)][BN][64 bytes].
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int Cin_o, int Cout_o,
                                    int taps, int transpose, int BN, int n_tiles, int nchunk, int row_bytes,
                                    long long total_elems) {
    constexpr int ES = (int)sizeof(T);
    constexpr int EPK = 64 / ES;   // elements per k-step row
    constexpr int CCE = 256 / ES;
    const int Kin = transpose ? Cout_o : Cin_o;     // GEMM K channels
    const int Nout = transpose ? Cin_o : Cout_o;    // GEMM N channels
    const int stages_per_tap = row_bytes >> 7;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_elems;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int e = (int)(r % EPK); r /= EPK;
        const int n = (int)(r % BN); r /= BN;
        const int ks = (int)(r % 2); r /= 2;
        const int st = (int)(r % stages_per_tap); r /= stages_per_tap;
        const int tap = (int)(r % taps); r /= taps;
        const int chunk = (int)(r % nchunk); r /= nchunk;
        const int ntile = (int)r;
        const int c = chunk * CCE + (st * 2 + ks) * EPK + e;
        const int co = ntile * BN + n;
        float v = 0.f;
        if (c < Kin && co < Nout && (st * 2 + ks) * 64 < row_bytes) {
            if (!transpose) v = w[((long long)tap * Cin_o + c) * Cout_o + co];
            else v = w[((long long)(taps - 1 - tap) * Cin_o + co) * Cout_o + c];
        }
        out[i] = from_f32<T>(v);
    }
}

struct PackGeom {
    int BN, n_tiles, nchunk, row_bytes;
    long long tile_bytes, total_bytes;
};

PackGeom pack_geom(int Kin, int Nout, int taps, int es) {
    PackGeom g;
    g.BN = (Nout % 128 == 0) ? 128 : 64;
    g.n_tiles = (Nout + g.BN - 1) / g.BN;
    const int kbytes = Kin * es;
    g.row_bytes = kbytes <= 128 ? 128 : 256;
    g.nchunk = (kbytes + 255) / 256;
    const int stages_per_tap = g.row_bytes >> 7;
    g.tile_bytes = (long long)g.nchunk * taps * stages_per_tap * 2 * g.BN * 64;
    g.total_bytes = g.tile_bytes * g.n_tiles;
    return g;
}

}  // namespace

extern "C" size_t mpn_conv_packed_bytes(int Cin, int Cout, int ksize, int transpose, int dtype) {
    const int es = dtype == MPN_F32 ? 4 : 2;
    const PackGeom g = pack_geom(transpose ? Cout : Cin, transpose ? Cin : Cout, ksize * ksize, es);
    return (size_t)g.total_bytes;
}

extern "C" int mpn_conv_pack_weights(const float* w_hwio, int Cin, int Cout, int ksize, int transpose,
                                     int dtype, void* out, mpn_stream_t stream) {
    MPN_REQUIRE(ksize == 1 || ksize == 3, MPN_ERR_BAD_SHAPE, "conv pack: ksize must be 1 or 3");
    MPN_REQUIRE(w_hwio && out, MPN_ERR_BAD_ARG, "conv pack: null pointer");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "conv pack: dtype");
    const int es = dtype == MPN_F32 ? 4 : 2;
    const int taps = ksize * ksize;
    const PackGeom g = pack_geom(transpose ? Cout : Cin, transpose ? Cin : Cout, taps, es);
    const long long total = g.total_bytes / es;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPN_F32)
        pack_weights_kernel<float><<<blocks, 256, 0, st>>>(w_hwio, (float*)out, Cin, Cout, taps, transpose, g.BN,
                                                          g.n_tiles, g.nchunk, g.row_bytes, total);
    else
        pack_weights_kernel<bf16_t><<<blocks, 256, 0, st>>>(w_hwio, (bf16_t*)out, Cin, Cout, taps, transpose, g.BN,
                                                           g.n_tiles, g.nchunk, g.row_bytes, total);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_conv_num_parts(int N, int H, int W, int ksize) {
    if (ksize == 3) return N * ((H + 7) / 8) * ((W + 15) / 16);
    return (int)(((long long)N * H * W + 127) / 128);
}

template <typename T, int TAPS, int BN>
static int launch_conv(const ConvParams& p, int m_tiles, hipStream_t st) {
    constexpr int NPIX = TAPS == 9 ? kHaloW * kHaloH : 128;
    constexpr int main_bytes = NPIX * 256 + 2 * (2 * BN * 64);
    constexpr int epi_bytes = 128 * (BN * 4 + 16);
    constexpr int smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    static bool attr_set = false;
    if (!attr_set) {
        MPN_HIP(hipFuncSetAttribute((const void*)conv_mfma_kernel<T, TAPS, BN>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    conv_mfma_kernel<T, TAPS, BN><<<dim3((unsigned)(m_tiles * p.n_tiles)), dim3(kThreads), smem, st>>>(p);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_conv_fwd(const void* x, const void* w_packed, void* y, int N, int H, int W, int Cin, int Cout,
                            int ksize, int dtype, const float* in_scale, const float* in_shift, int in_act,
                            float* stats_part, const void* up_res, mpn_stream_t stream) {
    MPN_REQUIRE(ksize == 1 || ksize == 3, MPN_ERR_BAD_SHAPE, "conv: ksize must be 1 or 3 (got %d)", ksize);
    MPN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, MPN_ERR_BAD_SHAPE, "conv: bad shape");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "conv: dtype %d", dtype);
    const int es = dtype == MPN_F32 ? 4 : 2;
    const int ve = 16 / es;
    MPN_REQUIRE(Cin % ve == 0 && Cout % ve == 0, MPN_ERR_BAD_SHAPE,
                "conv: Cin (%d) and Cout (%d) must be multiples of %d", Cin, Cout, ve);
    MPN_REQUIRE(x && w_packed && y, MPN_ERR_BAD_ARG, "conv: null pointer");
    MPN_REQUIRE(mpn_aligned16(x) && mpn_aligned16(w_packed) && mpn_aligned16(y), MPN_ERR_BAD_ALIGN,
                "conv: pointers must be 16-byte aligned");
    MPN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), MPN_ERR_BAD_ARG, "conv: scale/shift mismatch");
    MPN_REQUIRE(up_res == nullptr || (ksize == 1 && H % 2 == 0 && W % 2 == 0), MPN_ERR_BAD_ARG,
                "conv: upsample-add epilogue needs ksize 1 and even H, W");
    const PackGeom g = pack_geom(Cin, Cout, ksize * ksize, es);
    ConvParams p;
    p.x = x; p.wp = w_packed; p.y = y;
    p.in_scale = in_scale; p.in_shift = in_shift; p.in_act = in_act;
    p.stats_part = stats_part; p.up_res = up_res;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 7) / 8;
    p.M = (long long)N * H * W;
    p.row_bytes = g.row_bytes; p.nchunk = g.nchunk; p.n_tiles = g.n_tiles; p.wp_tile_bytes = g.tile_bytes;
    const int m_tiles = mpn_conv_num_parts(N, H, W, ksize);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPN_F32) {
        if (ksize == 3) return g.BN == 128 ? launch_conv<float, 9, 128>(p, m_tiles, st) : launch_conv<float, 9, 64>(p, m_tiles, st);
        return g.BN == 128 ? launch_conv<float, 1, 128>(p, m_tiles, st) : launch_conv<float, 1, 64>(p, m_tiles, st);
    }
    if (ksize == 3) return g.BN == 128 ? launch_conv<bf16_t, 9, 128>(p, m_tiles, st) : launch_conv<bf16_t, 9, 64>(p, m_tiles, st);
    return g.BN == 128 ? launch_conv<bf16_t, 1, 128>(p, m_tiles, st) : launch_conv<bf16_t, 1, 64>(p, m_tiles, st);
}
