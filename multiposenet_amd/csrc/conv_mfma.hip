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
#include "pointwise.h"
#include "conv3x3.h"
#include <type_traits>
#include <string.h>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    using Frag = bf16x8_t;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<half_t> {
    using Frag = H16<half_t>::x8;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x4_t& c) { c = H16<half_t>::mfma(a, b, c); }
};
template <> struct Mma<float> {
    using Frag = f32x4_t;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x4_t& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    }
};

__device__ __forceinline__ bf16x4_t o_tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4_t __attribute__((address_space(3)))*)(p));
}

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
    int x_stride, y_stride;   // elements between consecutive pixels of x / y (>= Cin / Cout: channel slices of wider tensors)
    int tiles_x, tiles_y;
    long long M;
    int row_bytes;  // 128 or 256: bytes of K staged per pixel row and channel chunk
    int nchunk;
    int n_tiles;
    long long wp_tile_bytes;  // packed bytes per n-tile
    int xcd_remap;            // XCD-aware block -> tile map
    // data gradients that also reduce for the batch-norm they feed (mpn_conv_bwd_data_bn, thin 1x1 layers): y is written MASKED
    // by that layer's activation (from bnr_x * bnr_scale + bnr_shift) and stats_part receives the sums of g and g * bnr_x
    const void* bnr_x;
    const float* bnr_scale;
    const float* bnr_shift;
    int bnr_act, bnr_xs;
#ifdef MPN_DIAG
    unsigned long long* dbg;  // diagnostic build only (tools/build_variant.sh -DMPN_DIAG): per-block s_memtime stamps, or NULL
#endif
};

// slots 0-4: s_memtime (shader clock) at the phase boundaries; slots 5, 6: s_memrealtime (100 MHz) at the first and last one
#ifdef MPN_DIAG
#define MPN_STAMP(k) do { if (p.dbg && threadIdx.x == 0) { p.dbg[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    if ((k) == 0) p.dbg[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime(); \
    if ((k) == 4) p.dbg[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define MPN_STAMP(k) do { } while (0)
#endif

constexpr int kThreads = 256;
constexpr int kHaloW = 18, kHaloH = 10;

// LDS pixel-row stride of the A image. Rows are PADDED instead of XOR-swizzled so that every fragment address is
// `per-lane base + compile-time/uniform offset` (ds_read_b128 with an immediate): the tap, k-step and m-tile cost
// no VALU in the MFMA loop. 160-byte rows (128 + 32) are conflict-free for the ds_read_b128 lane groups; 272-byte
// rows (256 + 16) leave one 2-way pair per group (5 instead of 4 LDS cycles) and let two blocks share a CU's LDS.
__host__ __device__ constexpr int a_row_stride(int row_bytes) { return row_bytes == 256 ? 272 : 160; }

// B (weights) image: [k-step][BN rows][64 B]; the four 16-byte slots of a row are XOR-swizzled with f(row>>2),
// f = {0,2,3,1}, which makes the 16 rows x 1 slot pattern of a ds_read_b128 lane group hit 16 distinct bank slots.
// The swizzle is applied by the PACK kernel (the global image is the LDS image), so staging stays a linear copy.
__host__ __device__ constexpr int b_swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

template <typename T>
__device__ __forceinline__ void apply_affine_act(Vec16<T>& v, const float* sc, const float* sh, float lo, float hi) {
    constexpr int VE = Vec16<T>::N;
    float f[VE];
    v.unpack(f);
#pragma unroll
    for (int j = 0; j < VE; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * sc[j] + sh[j], lo, hi);   // clamp = 1 VALU op
    v.pack(f);
}

// 4 consecutive channels of the storage type <-> f32
__device__ __forceinline__ void store4(float* p, const f32x4_t& v) { *reinterpret_cast<f32x4_t*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, const f32x4_t& v) {
    uint2 q;
    q.x = pack_bf16x2(v[0], v[1]);
    q.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = q;
}
__device__ __forceinline__ void store4(half_t* p, const f32x4_t& v) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    const h4_t h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ f32x4_t load4(const half_t* p) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    const h4_t h = __builtin_bit_cast(h4_t, *reinterpret_cast<const uint2*>(p));
    return (f32x4_t){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ f32x4_t load4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
__device__ __forceinline__ f32x4_t load4(const bf16_t* p) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    f32x4_t v;
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
    v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
    return v;
}

// Weight stages of 2 k-steps, 2 LDS buffers, DMA distance 1. A block owns 128 pixels (8 x 16 for 3x3).
// (the body is a device function so that the plain kernel and the grouped kernel - several independent launches of the
//  same instance, e.g. the four pyramid levels of one subnet stage, in ONE grid - share it; blk / nwg = this job's block
//  index and block count)
// Variants that were built, measured and removed again (256-pixel tiles, a 3-slot / 5-slot weight ring, the 32x32x16
// MFMA shape, a warp-specialised persistent kernel, batch-norm finalizes fused into the last-finishing blocks): DESIGN.md 4c.
template <typename T, int TAPS, int BN, int RB, bool BNR = false>
__device__ __forceinline__ void conv_mfma_body(const ConvParams& p, const int blk, const int nwg_job) {
    static_assert(!BNR || sizeof(T) == 2, "the fused batch-norm backward reduction: 16-bit storage");
    constexpr int MT = 4;                       // 16-pixel m-tiles per wave
    constexpr int ES = (int)sizeof(T);
    constexpr int VE = 16 / ES;
    constexpr int CCE = RB / ES;   // channels per chunk
    constexpr int NPIX = TAPS == 9 ? kHaloW * kHaloH : MT * 32;
    constexpr int NT = BN / 32;                 // 16-channel tiles per wave
    constexpr int KSPS = 2;                     // k-steps per weight stage
    constexpr int STAGE_BYTES = KSPS * BN * 64;
    constexpr int BVEC = STAGE_BYTES / (kThreads * 16);
    static_assert(BVEC >= 1, "a weight stage is at least one 16-byte vector per thread");
    constexpr int RS = a_row_stride(RB);
    constexpr int SLOTS = RB >> 4;              // 16-byte slots per pixel row: 8 or 16
    constexpr int KSTEPS = RB >> 6;             // 64-byte k-steps per chunk: 2 or 4
    constexpr int SPT = KSTEPS / KSPS;          // weight stages per tap
    constexpr int AVEC = (NPIX * SLOTS + kThreads - 1) / kThreads;
    using Frag = typename Mma<T>::Frag;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Bs = smem + NPIX * RS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;

    // XCD-aware block -> tile map: the dispatcher deals consecutive block ids round-robin over the 8 XCDs (each with its
    // own L2), so the blocks that share an XCD (same id % 8) get a CONTIGUOUS range of work ids: the n-tiles of one pixel
    // tile (which re-read the same A rows) and neighbouring pixel tiles (which share halo rows) then hit in one L2
    // instead of fetching the rows once per XCD. Bijective for any grid size.
    int wid = blk;
    if (p.xcd_remap) {
        const int nwg = nwg_job, q = nwg >> 3, r = nwg & 7, xcd = wid & 7;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wid >> 3);
    }
    const int ntile = wid % p.n_tiles;
    const int mtile = wid / p.n_tiles;
    const int n0 = ntile * BN;

    // ---- tile coordinates
    int img = 0, oy0 = 0, ox0 = 0;
    long long m0 = 0;
    if (TAPS == 9) {
        const int tx = mtile % p.tiles_x;
        const int t2 = mtile / p.tiles_x;
        const int ty = t2 % p.tiles_y;
        img = t2 / p.tiles_y;
        oy0 = ty * (MT * 2);
        ox0 = tx * 16;
    } else {
        m0 = (long long)mtile * (MT * 32);
    }
    const int total_stages = p.nchunk * TAPS * SPT;

    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const unsigned char* __restrict__ wsrc =
        reinterpret_cast<const unsigned char*>(p.wp) + (long long)ntile * p.wp_tile_bytes;

    // ---- accumulators: acc[mt][nt] holds D^T: lane (l15, lq) -> pixel mt*16+l15, channels nt*16+lq*4+{0..3}
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // per-lane fragment base addresses: everything else is a compile-time or wave-uniform offset
    const unsigned char* abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = wm * (MT * 16) + mt * 16 + l15;
        const int pix = (TAPS == 9) ? ((row >> 4) * kHaloW + (row & 15)) : row;
        abase[mt] = As + pix * RS + lq * 16;
    }
    const unsigned char* bbase = Bs + (wn * (BN / 2) + l15) * 64 + ((lq ^ b_swz(l15)) << 4);

    // ---- weights: LDS-DMA (global_load_lds_dwordx4), no VGPR staging and no ds_write. Stage s+1 is in flight into
    // buffer (s+1)%2 while stage s is computed from buffer s%2; the packed image is already in LDS order, so the
    // destination of every wave-instruction is the linear 1 KB piece `wave-uniform base + lane*16`.
    auto b_issue = [&](int stage, int buf) {
        const unsigned char* src = wsrc + (size_t)stage * STAGE_BYTES + (size_t)tid * 16;
        unsigned char* dst = Bs + buf * STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < BVEC; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * kThreads * 16),
                                             (__attribute__((address_space(3))) void*)(dst + i * kThreads * 16), 16, 0, 0);
    };
    b_issue(0, 0);

    MPN_STAMP(0);
    const bool affine = (p.in_scale != nullptr);
    const float act_lo = (p.in_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float act_hi = (p.in_act == MPN_ACT_RELU6) ? 6.f : INFINITY;

    // fragment register sets (double-buffered across k-steps so LDS latency hides under the MFMAs)
    Frag aP[MT], bP[NT], aQ[MT], bQ[NT];
    auto load_frags = [&](auto& a, auto& b, int a_off, int b_off) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const Frag*>(abase[mt] + a_off);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const Frag*>(bbase + b_off + nt * 1024);
    };
    auto mma_all = [&](const auto& a, const auto& b) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Mma<T>::run(b[nt], a[mt], acc[mt][nt]);   // D^T = W^T x A^T
    };

    int s = 0;
    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        __syncthreads();  // every wave has finished reading the previous A image
        // ================= stage the A image (halo or flat rows) for this channel chunk
        {
            const int slot = tid & (SLOTS - 1);  // kThreads % SLOTS == 0 -> fixed per thread
            const int ce = chunk * CCE + slot * VE;
            const bool cvalid = ce < p.Cin;
            float sc[VE], sh[VE];
            if (affine && cvalid) {
#pragma unroll
                for (int j = 0; j < VE; ++j) { sc[j] = p.in_scale[ce + j]; sh[j] = p.in_shift[ce + j]; }
            } else {
#pragma unroll
                for (int j = 0; j < VE; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
            }
            constexpr int nvec = NPIX * SLOTS;
            constexpr int pix_step = kThreads / SLOTS;
            Vec16<T> v[AVEC];
            bool inb[AVEC];
#pragma unroll
            for (int i = 0; i < AVEC; ++i) {
                const int vi = tid + i * kThreads;
                const int pix = (tid / SLOTS) + i * pix_step;
                bool ok = cvalid && (vi < nvec);
                long long off = 0;
                if (TAPS == 9) {
                    const int hy = pix / kHaloW, hx = pix - hy * kHaloW;
                    const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                    ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    off = (((long long)img * p.H + iy) * p.W + ix) * p.x_stride + ce;
                } else {
                    const long long m = m0 + pix;
                    ok = ok && (m < p.M);
                    off = m * p.x_stride + ce;
                }
                inb[i] = ok;
                if (ok) v[i].load(x + off); else v[i].zero();
            }
#pragma unroll
            for (int i = 0; i < AVEC; ++i) {
                const int vi = tid + i * kThreads;
                if (vi < nvec) {
                    const int pix = (tid / SLOTS) + i * pix_step;
                    if (affine && inb[i]) apply_affine_act<T>(v[i], sc, sh, act_lo, act_hi);
                    *reinterpret_cast<uint4*>(As + pix * RS + (slot << 4)) = *reinterpret_cast<const uint4*>(&v[i].raw);
                }
            }
        }
        // the LDS-DMA of this chunk's first weight stage and the A image must both have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (chunk == 0) MPN_STAMP(1);

        for (int sl = 0; sl < TAPS * SPT; ++sl, ++s) {
            const int tap = sl / SPT, h = sl - tap * SPT;
            const int ky = (tap * 11) >> 5;   // tap / 3 for tap < 9
            const int a_off = ((TAPS == 9) ? (ky * kHaloW + (tap - 3 * ky)) * RS : 0) + h * (KSPS * 64);   // wave-uniform
            const int b_off = (s & 1) * STAGE_BYTES;
            if (s + 1 < total_stages) b_issue(s + 1, (s + 1) & 1);   // buffer (s+1)%2 was last read in stage s-1: all waves are past it
            load_frags(aP, bP, a_off, b_off);
            load_frags(aQ, bQ, a_off + 64, b_off + BN * 64);
            mma_all(aP, bP);
            mma_all(aQ, bQ);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next stage's weights have landed (this wave's pieces)
            __syncthreads();
        }
    }
    MPN_STAMP(2);

    T* __restrict__ y = reinterpret_cast<T*>(p.y);
    const T* __restrict__ res = reinterpret_cast<const T*>(p.up_res);

    // ================= bf16 epilogue through an LDS image of the output tile (the staging buffers are dead: every wave
    // is past the last stage's barrier). The direct epilogue below costs the vector issue port more than the MFMAs of
    // the tile do: 16 stores per wave that each touch 16 x 32-byte segments, and 128 DPP adds for the statistics.
    // Here: (1) each wave writes its 64 px x BN/2 ch as bf16 into O[128 px][BN ch] (ds_write_b64, conflict-free with
    // 8 bytes of row padding); (2) the per-channel sum and sum of squares come from the MATRIX unit: the wave reads
    // its own part of O back with the transposing ds_read_b64_tr_b16 as 16x16x32 operands F[k = pixel][channel] and
    // accumulates ones x F (column sums) and F^T x F (Gram matrix: the diagonal is the sum of squares, products of
    // bf16 values are exact in f32) - 2 reads + 2 MFMAs per 32 px x 16 ch, no cross-lane VALU work; the k order is
    // free (both operands are the same registers), so each 32-lane half reads 8 rows 4 apart = 8 distinct bank
    // windows; (3) after one barrier, whole 16-byte pieces of pixel rows go to global memory, 256 contiguous bytes per
    // 16 lanes. The statistics are those of the ROUNDED outputs - exactly the tensor the consumer normalises.
    if constexpr (sizeof(T) == 2) {
        // The upsample-add variant (FPN laterals, fpn.py:42-47: y = conv(x) + nearest-2x(res)) takes the same route: the residual
        // is added to the f32 accumulators on the way INTO the image (one rounding, as in the direct epilogue). Its 1x1 tile is 128
        // consecutive pixels: the tile origin's (image, row, column) come from one scalar division, a lane's pixel from compares
        // (the direct epilogue's two 64-bit divisions per pixel and its 32-byte-segment stores cost lateral2 a third of its time).
        // Maps narrower than 32 columns or smaller than a tile keep the direct epilogue.
        const bool res_tiled = res != nullptr && TAPS == 1 && p.W >= 32 && (long long)p.H * p.W >= 128;
        if (res == nullptr || res_tiled) {
            constexpr int RSO = BN * 2 + 8;
            constexpr int ROWS = MT * 32;
            unsigned char* O = smem;
            float* red = reinterpret_cast<float*>(smem + ROWS * RSO);   // [2 wm][2][BN]
            int rn0 = 0, roy0 = 0, rox0 = 0;                             // (scalar) image / row / column of the tile's first pixel
            if (res_tiled) {
                rox0 = (int)(m0 % p.W);
                const long long t_ = m0 / p.W;
                roy0 = (int)(t_ % p.H);
                rn0 = (int)(t_ / p.H);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = wm * (MT * 16) + mt * 16 + l15;
                bool ok;
                if (TAPS == 9) ok = (oy0 + (row >> 4)) < p.H && (ox0 + (row & 15)) < p.W;
                else ok = (m0 + row) < p.M;
                long long roff = 0;
                if (res_tiled) {
                    const int v_ = rox0 + row;                                               // < W + 128
                    const int q_ = (v_ >= p.W) + (v_ >= 2 * p.W) + (v_ >= 3 * p.W) + (v_ >= 4 * p.W);   // (W >= 32: at most four wraps)
                    const int ox_ = v_ - q_ * p.W;
                    int oy_ = roy0 + q_, n_ = rn0;
                    if (oy_ >= p.H) { oy_ -= p.H; ++n_; }                                    // (H * W >= 128: at most one image boundary)
                    roff = ok ? (((long long)n_ * (p.H >> 1) + (oy_ >> 1)) * (p.W >> 1) + (ox_ >> 1)) * p.Cout : 0;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4_t v = acc[mt][nt];
                    if (res_tiled) {
                        const int c_ = n0 + wn * (BN / 2) + nt * 16 + lq * 4;
                        if (c_ < p.Cout) v += load4(res + roff + c_);
                    }
                    if (!ok) v = (f32x4_t){0.f, 0.f, 0.f, 0.f};   // out-of-image pixels must not count in the statistics
                    store4(reinterpret_cast<T*>(O + row * RSO + (wn * (BN / 2) + nt * 16 + lq * 4) * 2), v);
                }
            }
            MPN_STAMP(3);
            if (!BNR && p.stats_part != nullptr) {
                f32x4_t sa[NT], ga[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { sa[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; ga[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
                using HX = H16<std::conditional_t<sizeof(T) == 2, T, bf16_t>>;   // (this branch is 16-bit storage only)
                typename HX::x8 ones;
#pragma unroll
                for (int j = 0; j < 8; ++j) ones[j] = 1.0f;
                // lane 4q+pp of a 16-lane group supplies the address of block row q, channels 4pp..4pp+3
                const unsigned char* tb = O + (wm * (MT * 16) + (lq >> 1) * 2 + 4 * ((lq & 1) * 4 + (l15 >> 2))) * RSO +
                                          (wn * (BN / 2) + 4 * (l15 & 3)) * 2;
#pragma unroll
                for (int ks = 0; ks < MT / 2; ++ks)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const typename HX::x4 lo = HX::tr_read(tb + ks * 32 * RSO + nt * 32);
                        const typename HX::x4 hi = HX::tr_read(tb + (ks * 32 + 1) * RSO + nt * 32);
                        const typename HX::x8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        sa[nt] = HX::mfma(ones, f, sa[nt]);
                        ga[nt] = HX::mfma(f, f, ga[nt]);
                    }
                // D layout: lane (l15 = column, lq) holds rows 4 lq + r. Column sums: every row; Gram diagonal: row == column
                const int r = l15 & 3;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float q = r == 0 ? ga[nt][0] : (r == 1 ? ga[nt][1] : (r == 2 ? ga[nt][2] : ga[nt][3]));
                    if (lq == (l15 >> 2)) {
                        const int cl = wn * (BN / 2) + nt * 16 + l15;
                        red[(wm * 2 + 0) * BN + cl] = sa[nt][0];
                        red[(wm * 2 + 1) * BN + cl] = q;
                    }
                }
            }
            __syncthreads();
            {
                constexpr int SL = BN / 8;            // 16-byte pieces per pixel row
                constexpr int PPP = kThreads / SL;    // pixel rows per pass
                const int slot = tid % SL;
                const bool cok = n0 + slot * 8 < p.Cout;
                // BNR: a thread keeps its 16-byte piece (8 channels) over its ROWS / PPP rows: the fed layer's raw rows (requested
                // up front), the mask, and the running sums of g and g * x
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                uint4 bx[BNR ? ROWS / PPP : 1];
                f32x2_t bsc[4], bsh[4], bs[4], bq[4];
                float blo = -INFINITY, bhi = INFINITY;
                if constexpr (BNR) {
                    const int cch = cok ? n0 + slot * 8 : 0;
                    const T* xw = reinterpret_cast<const T*>(p.bnr_x) + cch;
#pragma unroll
                    for (int i = 0; i < ROWS / PPP; ++i) {
                        long long px;
                        if (TAPS == 9) {                              // (tile pixels outside the image: a valid address, masked below)
                            const int row = tid / SL + i * PPP;
                            const int oy = min(oy0 + (row >> 4), p.H - 1), ox = min(ox0 + (row & 15), p.W - 1);
                            px = ((long long)img * p.H + oy) * p.W + ox;
                        } else {
                            px = m0 + tid / SL + i * PPP;
                            if (px >= p.M) px = p.M - 1;              // (rows past the end hold dy = 0)
                        }
                        bx[i] = *reinterpret_cast<const uint4*>(xw + px * p.bnr_xs);
                    }
                    const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(p.bnr_scale + cch), s1 = *reinterpret_cast<const f32x4_t*>(p.bnr_scale + cch + 4);
                    const f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(p.bnr_shift + cch), h1 = *reinterpret_cast<const f32x4_t*>(p.bnr_shift + cch + 4);
                    bsc[0] = (f32x2_t){s0[0], s0[1]}; bsc[1] = (f32x2_t){s0[2], s0[3]}; bsc[2] = (f32x2_t){s1[0], s1[1]}; bsc[3] = (f32x2_t){s1[2], s1[3]};
                    bsh[0] = (f32x2_t){h0[0], h0[1]}; bsh[1] = (f32x2_t){h0[2], h0[3]}; bsh[2] = (f32x2_t){h1[0], h1[1]}; bsh[3] = (f32x2_t){h1[2], h1[3]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { bs[j] = (f32x2_t){0.f, 0.f}; bq[j] = (f32x2_t){0.f, 0.f}; }
                    blo = p.bnr_act != MPN_ACT_NONE ? 0.f : -INFINITY;
                    bhi = p.bnr_act == MPN_ACT_RELU6 ? 6.f : INFINITY;
                }
#pragma unroll
                for (int i = 0; i < ROWS / PPP; ++i) {
                    const int row = tid / SL + i * PPP;
                    bool ok;
                    long long pixel;
                    if (TAPS == 9) {
                        const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
                        ok = oy < p.H && ox < p.W;
                        pixel = ((long long)img * p.H + oy) * p.W + ox;
                    } else {
                        pixel = m0 + row;
                        ok = pixel < p.M;
                    }
                    if ((BNR || ok) && cok) {
                        uint2 a = *reinterpret_cast<const uint2*>(O + row * RSO + slot * 16);
                        uint2 b = *reinterpret_cast<const uint2*>(O + row * RSO + slot * 16 + 8);
                        if constexpr (BNR) {
                            // g = dy where the fed batch-norm's activation passes (the test of bn_bwd_reduce / bn_bwd_apply), else 0
                            const unsigned xu[4] = {bx[i].x, bx[i].y, bx[i].z, bx[i].w};
                            unsigned du[4] = {a.x, a.y, b.x, b.y};
                            if (TAPS == 9 && !ok) { du[0] = 0u; du[1] = 0u; du[2] = 0u; du[3] = 0u; }   // (a 3x3 tile past the image edge is not zero)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const f32x2_t xf = {to_f32(__builtin_bit_cast(T, (unsigned short)(xu[j] & 0xffffu))),
                                                    to_f32(__builtin_bit_cast(T, (unsigned short)(xu[j] >> 16)))};
                                const f32x2_t pre = xf * bsc[j] + bsh[j];
                                const unsigned m = ((pre[0] > blo && pre[0] < bhi) ? 0x0000ffffu : 0u) | ((pre[1] > blo && pre[1] < bhi) ? 0xffff0000u : 0u);
                                du[j] &= m;
                                const f32x2_t gf = {to_f32(__builtin_bit_cast(T, (unsigned short)(du[j] & 0xffffu))),
                                                    to_f32(__builtin_bit_cast(T, (unsigned short)(du[j] >> 16)))};
                                bs[j] += gf;
                                bq[j] += gf * xf;
                            }
                            a = make_uint2(du[0], du[1]); b = make_uint2(du[2], du[3]);
                        }
                        if (ok) *reinterpret_cast<uint4*>(y + pixel * p.y_stride + n0 + slot * 8) = make_uint4(a.x, a.y, b.x, b.y);
                    }
                }
                if constexpr (BNR) {
                    // the row lanes of a piece inside a wave (lane bits above log2 SL), fixed butterfly; then the four waves through
                    // `red` in a fixed order: (wave 0 + wave 1) + (wave 2 + wave 3)
#pragma unroll
                    for (int o = SL; o < 64; o <<= 1)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            bs[j][0] += __shfl_xor(bs[j][0], o, 64); bs[j][1] += __shfl_xor(bs[j][1], o, 64);
                            bq[j][0] += __shfl_xor(bq[j][0], o, 64); bq[j][1] += __shfl_xor(bq[j][1], o, 64);
                        }
                    const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
                        if ((wv & 1) == ph && ln < SL) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float* r0 = red + ((wv >> 1) * 2 + 0) * BN + ln * 8 + 2 * j;
                                float* r1 = red + ((wv >> 1) * 2 + 1) * BN + ln * 8 + 2 * j;
                                if (ph == 0) { r0[0] = bs[j][0]; r0[1] = bs[j][1]; r1[0] = bq[j][0]; r1[1] = bq[j][1]; }
                                else { r0[0] += bs[j][0]; r0[1] += bs[j][1]; r1[0] += bq[j][0]; r1[1] += bq[j][1]; }
                            }
                        }
                        __syncthreads();
                    }
                }
            }
            if (p.stats_part != nullptr && tid < 2 * BN) {
                const int which = tid / BN, c = tid % BN;
                if (n0 + c < p.Cout)
                    p.stats_part[((long long)mtile * 2 + which) * p.Cout + n0 + c] = red[which * BN + c] + red[(2 + which) * BN + c];
            }
            MPN_STAMP(4);
            return;
        }
    }

    // ================= direct epilogue (f32 parity build, and the upsample-add variant): each lane owns 4 consecutive
    // output channels of one pixel per (mt, nt) tile -> one 8-byte (bf16) / 16-byte (f32) store; no LDS round trip.
    const int cbase = n0 + wn * (BN / 2) + lq * 4;   // + nt*16
    f32x4_t ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { ssum[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; ssq[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = wm * 64 + mt * 16 + l15;
        bool ok;
        long long pixel;
        int n_i, oy, ox;
        if (TAPS == 9) {
            oy = oy0 + (row >> 4);
            ox = ox0 + (row & 15);
            n_i = img;
            ok = oy < p.H && ox < p.W;
            pixel = ((long long)img * p.H + oy) * p.W + ox;
        } else {
            pixel = m0 + row;
            ok = pixel < p.M;
            ox = (int)(pixel % p.W);
            const long long t = pixel / p.W;
            oy = (int)(t % p.H);
            n_i = (int)(t / p.H);
        }
        long long roff = 0;
        if (res != nullptr) roff = (((long long)n_i * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * p.Cout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = cbase + nt * 16;
            if (ok && c < p.Cout) {
                f32x4_t v = acc[mt][nt];
                if (res != nullptr) v += load4(res + roff + c);
                ssum[nt] += v;
                ssq[nt] += v * v;
                store4(y + pixel * p.y_stride + c, v);
            }
        }
    }

    MPN_STAMP(3);
    if (p.stats_part != nullptr) {
        // reduce over the 16 pixels (lanes l15) of each 16-lane row with DPP adds (pure VALU, no LDS crossbar):
        // quad_perm xor1, xor2, then row_half_mirror and row_mirror complete the 16-lane sum in every lane.
        auto row_sum16 = [](float v) -> float {
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
            return v;
        };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ssum[nt][r] = row_sum16(ssum[nt][r]);
                ssq[nt][r] = row_sum16(ssq[nt][r]);
            }
        float* red = reinterpret_cast<float*>(Bs);   // [2 wm][2][BN]; the weight buffers are dead (last barrier passed)
        if (l15 == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cl = wn * (BN / 2) + nt * 16 + lq * 4 + r;
                    red[(wm * 2 + 0) * BN + cl] = ssum[nt][r];
                    red[(wm * 2 + 1) * BN + cl] = ssq[nt][r];
                }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;
            if (n0 + c < p.Cout)
                p.stats_part[((long long)mtile * 2 + which) * p.Cout + n0 + c] = red[which * BN + c] + red[(2 + which) * BN + c];
        }
    }
    MPN_STAMP(4);
}

template <typename T, int TAPS, int BN, int RB, bool BNR = false>
__global__ __launch_bounds__(kThreads, 2) void conv_mfma_kernel(const ConvParams p) {
    conv_mfma_body<T, TAPS, BN, RB, BNR>(p, blockIdx.x, gridDim.x);
}

// up to five independent jobs of one kernel instance in one grid (largest first): the small pyramid levels are a few
// dozen to a few hundred tiles each - as launches of their own they are latency-bound tails of 15-45 us, inside the
// level-2 grid they fill in at its throughput
constexpr int kMaxGroup = 5;   // (the keypoint subnet has 4 pyramid levels, the RetinaNet head 5)
struct ConvGroup {
    ConvParams p[kMaxGroup];
    int begin[kMaxGroup + 1];   // first block of each job; begin[njobs] = grid size
    int njobs;
};
template <typename T, int TAPS, int BN, int RB, bool BNR = false>
__global__ __launch_bounds__(kThreads, 2) void conv_mfma_grouped_kernel(const ConvGroup g) {
    int job = 0;
#pragma unroll
    for (int j = 1; j < kMaxGroup; ++j)
        if (j < g.njobs && (int)blockIdx.x >= g.begin[j]) job = j;   // wave-uniform
    conv_mfma_body<T, TAPS, BN, RB, BNR>(g.p[job], (int)blockIdx.x - g.begin[job], g.begin[job + 1] - g.begin[job]);
}


// ------------------------------------------------------------------ weight packing
// Packed order (per n-tile of BN output channels): [chunk][tap][stage][kstep(2)][BN][64 bytes].
struct PackDesc {           // one (conv, direction) packing job; lives in device memory for the batched kernel
    const float* w;
    void* out;
    int Cin_o, Cout_o, taps, transpose, BN, n_tiles, nchunk, row_bytes;
    long long total_elems;
    int block_begin, block_count;   // blocks [block_begin, block_begin + block_count) of the batched launch
};

template <typename T>
__device__ __forceinline__ void pack_one(const float* __restrict__ w, T* __restrict__ out, int Cin_o, int Cout_o, int taps,
                                         int transpose, int BN, int nchunk, int row_bytes, long long i) {
    constexpr int ES = (int)sizeof(T);
    constexpr int EPK = 64 / ES;   // elements per k-step row
    const int Kin = transpose ? Cout_o : Cin_o;     // GEMM K channels
    const int Nout = transpose ? Cin_o : Cout_o;    // GEMM N channels
    if (row_bytes == -2) {  // conv3x3.h, 64-channel tiles: [n-tile][chunk 64][kx][ky][k-step][co 64][64 bytes]
        long long r = i;
        const int e = (int)(r % 32); r /= 32;
        const int n = (int)(r % 64); r /= 64;
        const int ks = (int)(r % 2); r /= 2;
        const int ky = (int)(r % 3); r /= 3;
        const int kx = (int)(r % 3); r /= 3;
        const int chunk = (int)(r % nchunk); r /= nchunk;
        const int c = chunk * 64 + ks * 32 + e, co = (int)r * 64 + n, tap = ky * 3 + kx;
        const float v = !transpose ? w[((long long)tap * Cin_o + c) * Cout_o + co] : w[((long long)(8 - tap) * Cin_o + co) * Cout_o + c];
        const int q = e >> 3, within = e & 7;
        out[i - e + ((q ^ mpn_c3::swz(n & 15)) * 8 + within)] = from_f32<T>(v);
        return;
    }
    if (row_bytes < 0) {    // the 256-pixel 3x3 kernel's image (conv3x3.h): [n-tile][chunk 64][kx][k-step][ky][co 128][64 bytes]
        long long r = i;
        const int e = (int)(r % 32); r /= 32;
        const int n = (int)(r % 128); r /= 128;
        const int ky = (int)(r % 3); r /= 3;
        const int ks = (int)(r % 2); r /= 2;
        const int kx = (int)(r % 3); r /= 3;
        const int chunk = (int)(r % nchunk); r /= nchunk;
        const int c = chunk * 64 + ks * 32 + e, co = (int)r * 128 + n, tap = ky * 3 + kx;
        const float v = !transpose ? w[((long long)tap * Cin_o + c) * Cout_o + co] : w[((long long)(8 - tap) * Cin_o + co) * Cout_o + c];
        const int q = e >> 3, within = e & 7;
        out[i - e + ((q ^ mpn_c3::swz(n & 15)) * 8 + within)] = from_f32<T>(v);
        return;
    }
    if (row_bytes == 0) {   // plain [N][K]: W^T for the forward, the HWIO matrix itself for the data gradient
        const int n = (int)(i / Kin), k = (int)(i - (long long)n * Kin);
        out[i] = from_f32<T>(transpose ? w[(long long)n * Cout_o + k] : w[(long long)k * Cout_o + n]);
        return;
    }
    const int CCE = row_bytes / ES;   // channels per chunk
    const int stages_per_tap = row_bytes >> 7;
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
    // destination slot inside the 64-byte row is XOR-swizzled (see b_swz): the global image is the LDS image
    constexpr int EPS = 16 / ES;   // elements per 16-byte slot
    const int q = e / EPS, within = e - q * EPS;
    const long long dst = i - e + (long long)((q ^ b_swz(n & 15)) * EPS + within);
    out[dst] = from_f32<T>(v);
}

template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int Cin_o, int Cout_o,
                                    int taps, int transpose, int BN, int n_tiles, int nchunk, int row_bytes,
                                    long long total_elems) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_elems;
         i += (long long)gridDim.x * blockDim.x)
        pack_one<T>(w, out, Cin_o, Cout_o, taps, transpose, BN, nchunk, row_bytes, i);
}

// all packing jobs of the network in ONE launch (68 tiny launches per optimizer step otherwise)
template <typename T>
__global__ void pack_weights_batched_kernel(const PackDesc* __restrict__ descs, int ndesc) {
    __shared__ int job;           // the job that owns this block: one parallel read of block_begin + a ballot
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
    const PackDesc d = descs[job];
    const int lb = blockIdx.x - d.block_begin;
    for (long long i = (long long)lb * blockDim.x + threadIdx.x; i < d.total_elems; i += (long long)d.block_count * blockDim.x)
        pack_one<T>(d.w, reinterpret_cast<T*>(d.out), d.Cin_o, d.Cout_o, d.taps, d.transpose, d.BN, d.nchunk, d.row_bytes, i);
}

struct PackGeom {
    int BN, n_tiles, nchunk, row_bytes;   // row_bytes == 0: plain [N][K] matrix for the GEMM kernel of pointwise.hip; -1: conv3x3.hip
    long long tile_bytes, total_bytes;
};

PackGeom pack_geom(int Kin, int Nout, int taps, int es, int dtype) {
    PackGeom g;
    if (dtype == MPN_BF16 && pw_gemm_eligible(Kin, Nout, taps, es)) {
        g.BN = 256; g.n_tiles = Nout / 256; g.nchunk = 1; g.row_bytes = 0;
        g.tile_bytes = 256ll * Kin * es; g.total_bytes = (long long)Nout * Kin * es;
        return g;
    }
    if ((dtype == MPN_BF16 || dtype == MPN_F16) && mpn_c3::eligible(Kin, Nout, taps, es)) {
        g.BN = 128; g.n_tiles = Nout / 128; g.nchunk = Kin / 64; g.row_bytes = -1;
        g.tile_bytes = mpn_c3::tile_bytes(Kin); g.total_bytes = mpn_c3::packed_bytes(Kin, Nout);
        return g;
    }
    if ((dtype == MPN_BF16 || dtype == MPN_F16) && mpn_c3::eligible64(Kin, Nout, taps, es)) {
        g.BN = 64; g.n_tiles = Nout / 64; g.nchunk = Kin / 64; g.row_bytes = -2;
        g.tile_bytes = mpn_c3::tile_bytes64(Kin); g.total_bytes = mpn_c3::packed_bytes(Kin, Nout);
        return g;
    }
    g.BN = (Nout % 128 == 0) ? 128 : 64;
    g.n_tiles = (Nout + g.BN - 1) / g.BN;
    const int kbytes = Kin * es;
    // chunk (= LDS pixel row) width: 128 bytes of K for 3x3 (a 28.8 KB halo image: the two co-resident blocks of a CU
    // alternate their staging and MFMA phases at twice the granularity), 256 for 1x1 when K is that wide
    // 1x1: 256-byte chunks only from 1024 bytes of K on (512 bf16 channels): with 128 / 256 input channels the 128-byte
    // chunks (20 KB A image, three blocks per CU) measured faster - 128->128 @128x128: 57.7 -> 53.1 us, 256->256 @64x64:
    // 42.4 -> 37.2 us - while the 512 / 1024-channel layers on the small maps prefer fewer, longer chunks
    const int pref = taps == 9 ? 128 : (kbytes >= 1024 ? 256 : 128);
    g.row_bytes = (kbytes <= 128 || pref == 128) ? 128 : 256;
    g.nchunk = (kbytes + g.row_bytes - 1) / g.row_bytes;
    const int stages_per_tap = g.row_bytes >> 7;
    g.tile_bytes = (long long)g.nchunk * taps * stages_per_tap * 2 * g.BN * 64;
    g.total_bytes = g.tile_bytes * g.n_tiles;
    return g;
}

}  // namespace

extern "C" size_t mpn_conv_packed_bytes(int Cin, int Cout, int ksize, int transpose, int dtype) {
    const int es = dtype == MPN_F32 ? 4 : 2;
    const PackGeom g = pack_geom(transpose ? Cout : Cin, transpose ? Cin : Cout, ksize * ksize, es, dtype);
    return (size_t)g.total_bytes;
}

extern "C" int mpn_conv_pack_weights(const float* w_hwio, int Cin, int Cout, int ksize, int transpose,
                                     int dtype, void* out, mpn_stream_t stream) {
    MPN_REQUIRE(ksize == 1 || ksize == 3, MPN_ERR_BAD_SHAPE, "conv pack: ksize must be 1 or 3");
    MPN_REQUIRE(w_hwio && out, MPN_ERR_BAD_ARG, "conv pack: null pointer");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16 || dtype == MPN_F16, MPN_ERR_BAD_DTYPE, "conv pack: dtype");
    const int es = dtype == MPN_F32 ? 4 : 2;
    const int taps = ksize * ksize;
    const PackGeom g = pack_geom(transpose ? Cout : Cin, transpose ? Cin : Cout, taps, es, dtype);
    const long long total = g.total_bytes / es;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE3(dtype, (pack_weights_kernel<T><<<blocks, 256, 0, st>>>(w_hwio, (T*)out, Cin, Cout, taps, transpose, g.BN, g.n_tiles,
                                                                              g.nchunk, g.row_bytes, total)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" size_t mpn_conv_pack_desc_bytes(void) { return sizeof(PackDesc); }

/* Fills ONE host-side descriptor (desc_host, mpn_conv_pack_desc_bytes() bytes) for the batched packer and returns the
 * number of blocks the job wants; block_begin = running sum over the jobs. */
extern "C" int mpn_conv_pack_desc_fill(void* desc_host, const float* w_hwio, int Cin, int Cout, int ksize, int transpose,
                                       int dtype, void* out, int block_begin) {
    if (!(ksize == 1 || ksize == 3) || !desc_host || !w_hwio || !out) return -1;
    const int es = dtype == MPN_F32 ? 4 : 2;
    const int taps = ksize * ksize;
    const PackGeom g = pack_geom(transpose ? Cout : Cin, transpose ? Cin : Cout, taps, es, dtype);
    PackDesc d;
    d.w = w_hwio; d.out = out; d.Cin_o = Cin; d.Cout_o = Cout; d.taps = taps; d.transpose = transpose;
    d.BN = g.BN; d.n_tiles = g.n_tiles; d.nchunk = g.nchunk; d.row_bytes = g.row_bytes;
    d.total_elems = g.total_bytes / es;
    long long blocks = (d.total_elems + 4 * 256 - 1) / (4 * 256);   // 4 elements per thread
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    d.block_begin = block_begin;
    d.block_count = (int)blocks;
    memcpy(desc_host, &d, sizeof(d));
    return (int)blocks;
}

extern "C" int mpn_conv_pack_weights_batched(const void* descs_device, int ndesc, int total_blocks, int dtype,
                                             mpn_stream_t stream) {
    MPN_REQUIRE(descs_device && ndesc > 0 && total_blocks > 0, MPN_ERR_BAD_ARG, "pack batched: bad arguments");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16 || dtype == MPN_F16, MPN_ERR_BAD_DTYPE, "pack batched: dtype");
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE3(dtype, (pack_weights_batched_kernel<T><<<total_blocks, 256, 0, st>>>((const PackDesc*)descs_device, ndesc)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

#ifdef MPN_DIAG
// diagnostic build only (never in the product library): device buffer of 8 u64 per block, or NULL
static void* g_conv_dbg = nullptr;
extern "C" void mpn_diag_set_conv_stamps(void* buf) { g_conv_dbg = buf; }
#endif

extern "C" int mpn_conv_num_parts(int N, int H, int W, int ksize) {
    if (ksize == 3) return N * ((H + 7) / 8) * ((W + 15) / 16);
    return (int)(((long long)N * H * W + 127) / 128);
}

/* Rows of the statistics slab that mpn_conv_fwd[_grouped] / mpn_conv_bwd_data_bn[_grouped] WRITE for one layer of this shape (what
 * the finalize must be told): the persistent 3x3 kernel (16-bit storage, Cin % 64 == 0, Cout % 64 == 0) writes one row per block
 * that has a tile of the layer - the same alone and inside a group -, every other kernel one row per tile (mpn_conv_num_parts, which
 * stays the upper bound to size the slab with). < 0: no device to ask for its compute-unit count. */
extern "C" int mpn_conv_stats_rows(int N, int H, int W, int Cin, int Cout, int ksize, int dtype) {
    MPN_REQUIRE(ksize == 1 || ksize == 3, MPN_ERR_BAD_SHAPE, "conv: ksize must be 1 or 3 (got %d)", ksize);
    MPN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, MPN_ERR_BAD_SHAPE, "conv: bad shape");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16 || dtype == MPN_F16, MPN_ERR_BAD_DTYPE, "conv: dtype %d", dtype);
    const int es = dtype == MPN_F32 ? 4 : 2;
    if (ksize == 3 && pack_geom(Cin, Cout, 9, es, dtype).row_bytes < 0) return mpn_c3::stats_rows(N, H, W, Cout);
    return mpn_conv_num_parts(N, H, W, ksize);
}

template <int TAPS, int BN, int RB> constexpr int conv_smem_bytes() {
    return (TAPS == 9 ? kHaloW * kHaloH : 128) * a_row_stride(RB) + 2 * (2 * BN * 64);
}

template <typename T, int TAPS, int BN, int RB, bool BNR = false>
static int launch_conv_rb(const ConvParams& p, int m_tiles, hipStream_t st) {
    constexpr int smem = conv_smem_bytes<TAPS, BN, RB>();
    static_assert(sizeof(T) != 2 || smem >= 128 * (BN * 2 + 8) + 4 * BN * 4, "the output image of the epilogue fits");
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)conv_mfma_kernel<T, TAPS, BN, RB, BNR>, smem, &attr_mask));
    conv_mfma_kernel<T, TAPS, BN, RB, BNR><<<dim3((unsigned)(m_tiles * p.n_tiles)), dim3(kThreads), smem, st>>>(p);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

template <typename T, int TAPS, int BN>
static int launch_conv(const ConvParams& p, int m_tiles, hipStream_t st) {
    if (p.row_bytes == 256) return launch_conv_rb<T, TAPS, BN, 256>(p, m_tiles, st);
    return launch_conv_rb<T, TAPS, BN, 128>(p, m_tiles, st);
}

static int conv_fill_params(ConvParams& p, const PackGeom& g, const void* x, const void* w_packed, void* y, int N, int H, int W,
                            int Cin, int Cout, int x_stride, int y_stride, int ksize, const float* in_scale,
                            const float* in_shift, int in_act, float* stats_part, const void* up_res) {
    p.x = x; p.wp = w_packed; p.y = y;
    p.in_scale = in_scale; p.in_shift = in_shift; p.in_act = in_act;
    p.stats_part = stats_part; p.up_res = up_res;
    p.bnr_x = nullptr; p.bnr_scale = nullptr; p.bnr_shift = nullptr; p.bnr_act = MPN_ACT_NONE; p.bnr_xs = 0;
#ifdef MPN_DIAG
    p.dbg = (unsigned long long*)g_conv_dbg;
#endif
    // measured: +10..14 % on 1x1 layers with several n-tiles (A rows re-read from the same L2), -4 % on single-n-tile layers
    p.xcd_remap = (ksize == 1 && g.n_tiles > 1);
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.x_stride = x_stride > 0 ? x_stride : Cin;
    p.y_stride = y_stride > 0 ? y_stride : Cout;
    p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 7) / 8;
    p.M = (long long)N * H * W;
    p.row_bytes = g.row_bytes; p.nchunk = g.nchunk; p.n_tiles = g.n_tiles; p.wp_tile_bytes = g.tile_bytes;
    return MPN_OK;
}

static void c3_fill_job(mpn_c3::Job& j, const void* x, const void* w_packed, void* y, int N, int H, int W, int Cin, int Cout,
                        int x_stride, int y_stride, const float* in_scale, const float* in_shift, int in_act, float* stats_part) {
    j.x = x; j.wp = w_packed; j.y = y; j.in_scale = in_scale; j.in_shift = in_shift; j.stats_part = stats_part; j.in_act = in_act;
    j.N = N; j.H = H; j.W = W; j.Cin = Cin; j.Cout = Cout;
    j.xs = x_stride > 0 ? x_stride : Cin; j.ys = y_stride > 0 ? y_stride : Cout;
    j.bnr_x = nullptr; j.bnr_scale = nullptr; j.bnr_shift = nullptr; j.bnr_act = MPN_ACT_NONE; j.bnr_xs = 0;
#ifdef MPN_DIAG
    j.dbg = (unsigned long long*)g_conv_dbg;
#endif
}

extern "C" int mpn_conv_fwd(const void* x, const void* w_packed, void* y, int N, int H, int W, int Cin, int Cout,
                            int x_stride, int y_stride, int ksize, int dtype, const float* in_scale, const float* in_shift,
                            int in_act, float* stats_part, const void* up_res, mpn_stream_t stream) {
    MPN_REQUIRE(ksize == 1 || ksize == 3, MPN_ERR_BAD_SHAPE, "conv: ksize must be 1 or 3 (got %d)", ksize);
    MPN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, MPN_ERR_BAD_SHAPE, "conv: bad shape");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16 || dtype == MPN_F16, MPN_ERR_BAD_DTYPE, "conv: dtype %d", dtype);
    const int es = dtype == MPN_F32 ? 4 : 2;
    const int ve = 16 / es;
    MPN_REQUIRE(Cin % ve == 0 && Cout % ve == 0, MPN_ERR_BAD_SHAPE,
                "conv: Cin (%d) and Cout (%d) must be multiples of %d", Cin, Cout, ve);
    MPN_REQUIRE((x_stride == 0 || (x_stride >= Cin && x_stride % ve == 0)) && (y_stride == 0 || (y_stride >= Cout && y_stride % ve == 0)),
                MPN_ERR_BAD_SHAPE, "conv: pixel strides (%d, %d) must be 0 or multiples of %d not below the channel counts", x_stride,
                y_stride, ve);
    MPN_REQUIRE(x && w_packed && y, MPN_ERR_BAD_ARG, "conv: null pointer");
    MPN_REQUIRE(mpn_aligned16(x) && mpn_aligned16(w_packed) && mpn_aligned16(y), MPN_ERR_BAD_ALIGN,
                "conv: pointers must be 16-byte aligned");
    MPN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), MPN_ERR_BAD_ARG, "conv: scale/shift mismatch");
    MPN_REQUIRE(up_res == nullptr || (ksize == 1 && H % 2 == 0 && W % 2 == 0), MPN_ERR_BAD_ARG,
                "conv: upsample-add epilogue needs ksize 1 and even H, W");
    const PackGeom g = pack_geom(Cin, Cout, ksize * ksize, es, dtype);
    if (g.row_bytes < 0) {    // 3x3, 16-bit storage, Cin % 64 == 0, Cout % 128 == 0: 256-pixel tiles, one 8-wave block per CU
        mpn_c3::Job job;
        c3_fill_job(job, x, w_packed, y, N, H, W, Cin, Cout, x_stride, y_stride, in_scale, in_shift, in_act, stats_part);
        return mpn_c3::launch(&job, 1, dtype, (hipStream_t)stream);
    }
    if (g.row_bytes == 0) {   // deep 1x1 layer: the GEMM kernel (weights packed as [Cout][Cin])
        MPN_REQUIRE(up_res == nullptr, MPN_ERR_BAD_ARG, "conv: the upsample-add epilogue needs Cout < 256 or Cin < 256");
        return pw_gemm_launch(x, w_packed, y, (long long)N * H * W, Cin, Cout, x_stride > 0 ? x_stride : Cin,
                              y_stride > 0 ? y_stride : Cout, in_scale, in_shift, in_act, stats_part, (hipStream_t)stream);
    }
    ConvParams p;
    conv_fill_params(p, g, x, w_packed, y, N, H, W, Cin, Cout, x_stride, y_stride, ksize, in_scale, in_shift, in_act, stats_part, up_res);
    const int m_tiles = mpn_conv_num_parts(N, H, W, ksize);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPN_F32) {
        if (ksize == 3) return g.BN == 128 ? launch_conv<float, 9, 128>(p, m_tiles, st) : launch_conv<float, 9, 64>(p, m_tiles, st);
        return g.BN == 128 ? launch_conv<float, 1, 128>(p, m_tiles, st) : launch_conv<float, 1, 64>(p, m_tiles, st);
    }
    if (dtype == MPN_F16) {
        if (ksize == 3) return g.BN == 128 ? launch_conv<half_t, 9, 128>(p, m_tiles, st) : launch_conv<half_t, 9, 64>(p, m_tiles, st);
        return g.BN == 128 ? launch_conv<half_t, 1, 128>(p, m_tiles, st) : launch_conv<half_t, 1, 64>(p, m_tiles, st);
    }
    if (ksize == 3) return g.BN == 128 ? launch_conv<bf16_t, 9, 128>(p, m_tiles, st) : launch_conv<bf16_t, 9, 64>(p, m_tiles, st);
    return g.BN == 128 ? launch_conv<bf16_t, 1, 128>(p, m_tiles, st) : launch_conv<bf16_t, 1, 64>(p, m_tiles, st);
}

template <int BN, bool BNR = false>
static int launch_conv_grouped(const ConvGroup& grp, int grid, hipStream_t st) {
    constexpr int smem = conv_smem_bytes<9, BN, 128>();
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)conv_mfma_grouped_kernel<bf16_t, 9, BN, 128, BNR>, smem, &attr_mask));
    conv_mfma_grouped_kernel<bf16_t, 9, BN, 128, BNR><<<dim3((unsigned)grid), dim3(kThreads), smem, st>>>(grp);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* Several independent 3x3 convolutions of the same channel geometry in ONE grid (the four pyramid levels of a subnet
 * stage: keypoint_subnet.py:64-91 applies phi_subnet to p2..p5 independently). bf16 3x3 with at most 5 jobs run in one
 * grid; anything else runs as the separate launches it replaces. Results are those of mpn_conv_fwd per job, bit for bit
 * (same kernel body, same tiles). x_stride[j] / y_stride[j] (arrays, may be NULL = dense): pixel strides of job j's input / output. */
extern "C" int mpn_conv_fwd_grouped(int njobs, const void* const* x, const void* const* w_packed, void* const* y, int N,
                                    const int* H, const int* W, int Cin, int Cout, const int* x_stride, const int* y_stride,
                                    int ksize, int dtype,
                                    const float* const* in_scale, const float* const* in_shift, int in_act,
                                    float* const* stats_part, mpn_stream_t stream) {
    MPN_REQUIRE(njobs > 0 && x && w_packed && y && H && W && in_scale && in_shift && stats_part, MPN_ERR_BAD_ARG,
                "conv grouped: bad arguments");
    const int es = dtype == MPN_F32 ? 4 : 2;
    const PackGeom g = pack_geom(Cin, Cout, ksize * ksize, es, dtype);
    bool same_variant = true;   // (affine on load or not: one kernel instance per grid)
    for (int j = 1; j < njobs; ++j) same_variant = same_variant && ((in_scale[j] == nullptr) == (in_scale[0] == nullptr));
    if (g.row_bytes < 0 && njobs <= mpn_c3::kMaxJobs && same_variant) {
        MPN_REQUIRE(N > 0, MPN_ERR_BAD_SHAPE, "conv grouped: bad shape");
        mpn_c3::Job jobs[mpn_c3::kMaxJobs];
        for (int j = 0; j < njobs; ++j) {
            MPN_REQUIRE(x[j] && w_packed[j] && y[j] && H[j] > 0 && W[j] > 0, MPN_ERR_BAD_ARG, "conv grouped: null pointer / bad size");
            MPN_REQUIRE(mpn_aligned16(x[j]) && mpn_aligned16(w_packed[j]) && mpn_aligned16(y[j]), MPN_ERR_BAD_ALIGN,
                        "conv grouped: pointers must be 16-byte aligned");
            MPN_REQUIRE((in_scale[j] == nullptr) == (in_shift[j] == nullptr), MPN_ERR_BAD_ARG, "conv grouped: scale/shift mismatch");
            const int ys = y_stride ? y_stride[j] : 0, xs = x_stride ? x_stride[j] : 0;
            MPN_REQUIRE((ys == 0 || (ys >= Cout && ys % 8 == 0)) && (xs == 0 || (xs >= Cin && xs % 8 == 0)), MPN_ERR_BAD_SHAPE,
                        "conv grouped: bad pixel strides %d, %d", xs, ys);
            c3_fill_job(jobs[j], x[j], w_packed[j], y[j], N, H[j], W[j], Cin, Cout, xs, ys, in_scale[j], in_shift[j], in_act, stats_part[j]);
        }
        return mpn_c3::launch(jobs, njobs, dtype, (hipStream_t)stream);
    }
    const bool fast = dtype == MPN_BF16 && ksize == 3 && njobs <= kMaxGroup && g.row_bytes == 128;
    if (!fast) {
        for (int j = 0; j < njobs; ++j)
            if (int rc = mpn_conv_fwd(x[j], w_packed[j], y[j], N, H[j], W[j], Cin, Cout, x_stride ? x_stride[j] : 0,
                                      y_stride ? y_stride[j] : 0, ksize, dtype,
                                      in_scale[j], in_shift[j], in_act, stats_part[j], nullptr, stream))
                return rc;
        return MPN_OK;
    }
    MPN_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, MPN_ERR_BAD_SHAPE, "conv grouped: bad shape");
    ConvGroup grp = {};
    int begin = 0;
    for (int j = 0; j < njobs; ++j) {
        MPN_REQUIRE(x[j] && w_packed[j] && y[j] && H[j] > 0 && W[j] > 0, MPN_ERR_BAD_ARG, "conv grouped: null pointer / bad size");
        MPN_REQUIRE(mpn_aligned16(x[j]) && mpn_aligned16(w_packed[j]) && mpn_aligned16(y[j]), MPN_ERR_BAD_ALIGN,
                    "conv grouped: pointers must be 16-byte aligned");
        MPN_REQUIRE((in_scale[j] == nullptr) == (in_shift[j] == nullptr), MPN_ERR_BAD_ARG, "conv grouped: scale/shift mismatch");
        const int ys = y_stride ? y_stride[j] : 0, xs = x_stride ? x_stride[j] : 0;
        MPN_REQUIRE((ys == 0 || (ys >= Cout && ys % 8 == 0)) && (xs == 0 || (xs >= Cin && xs % 8 == 0)), MPN_ERR_BAD_SHAPE,
                    "conv grouped: bad pixel strides %d, %d", xs, ys);
        conv_fill_params(grp.p[j], g, x[j], w_packed[j], y[j], N, H[j], W[j], Cin, Cout, xs, ys, 3, in_scale[j], in_shift[j], in_act,
                         stats_part[j], nullptr);
        grp.p[j].xcd_remap = 0;
        grp.begin[j] = begin;
        begin += mpn_conv_num_parts(N, H[j], W[j], 3) * g.n_tiles;
    }
    for (int j = njobs; j <= kMaxGroup; ++j) grp.begin[j] = begin;
    grp.njobs = njobs;
    hipStream_t st = (hipStream_t)stream;
    return g.BN == 128 ? launch_conv_grouped<128>(grp, begin, st) : launch_conv_grouped<64>(grp, begin, st);
}


/* Data gradients of up to five independent 3x3 convolutions (one grid, as mpn_conv_fwd_grouped on the transposed packed weights)
 * that ALSO do the first pass of the batch-norm backward of the layer they feed (keypoint_subnet.py:75-78: conv -> bn -> relu ->
 * conv: the gradient that leaves conv2's data gradient is the gradient w.r.t. relu(bn1(x))): dx[j] is written MASKED
 * (g = dx where the activation passed, computed from bn_x[j] * bn_scale[j] + bn_shift[j]; bn_bwd_apply's own mask is then a
 * no-op) and part[j] ([mpn_conv_num_parts(N,H,W,3)][2][C] floats) receives the partial sums of g and of g * bn_x (RAW x:
 * finish with a mpn_bn_bwd_fin_desc_fill_raw finalize, which forms sum g * xhat = invstd * (sum g x - mean * sum g)).
 * One tensor read (dx) and one launch less than mpn_bn_bwd_reduce afterwards. 16-bit storage, K % 64 == 0, K <= 512,
 * C % 64 == 0, C <= 512: mpn_conv_bwd_data_bn_supported says whether a geometry is covered. */
extern "C" int mpn_conv_bwd_data_bn_supported(int K, int C, int ksize, int dtype) {
    if (ksize == 1) {
        if (dtype != MPN_BF16 || K % 8 != 0 || C % 8 != 0) return 0;
        if (pw_gemm_eligible(K, C, 1, 2)) return 1;                     // the deep 1x1 layers (pointwise.hip)
        return pack_geom(K, C, 1, 2, dtype).row_bytes == 128 ? 1 : 0;   // thin ones: the tiled kernel with 128-byte chunks
    }
    if (ksize != 3) return 0;
    if ((dtype == MPN_BF16 || dtype == MPN_F16) && (mpn_c3::eligible(K, C, 9, 2) || mpn_c3::eligible64(K, C, 9, 2)) && C <= 512) return 1;
    // thin 3x3 data gradients (the detector's output convolutions: 8 / 24 -> 64 channels) on the tiled kernel
    return (dtype == MPN_BF16 && K % 8 == 0 && C % 8 == 0 && pack_geom(K, C, 9, 2, dtype).row_bytes == 128) ? 1 : 0;
}

extern "C" int mpn_conv_bwd_data_bn_grouped(int njobs, const void* const* dy, const void* const* w_packed_t, void* const* dx, int N,
                                            const int* H, const int* W, int K, int C, const int* dy_stride, const int* dx_stride,
                                            int dtype, const void* const* bn_x, const int* bn_x_stride,
                                            const float* const* bn_scale, const float* const* bn_shift, int bn_act,
                                            float* const* part, mpn_stream_t stream) {
    MPN_REQUIRE(njobs > 0 && njobs <= mpn_c3::kMaxJobs && dy && w_packed_t && dx && H && W && bn_x && bn_scale && bn_shift && part,
                MPN_ERR_BAD_ARG, "conv_bwd_data_bn: bad arguments");
    MPN_REQUIRE(mpn_conv_bwd_data_bn_supported(K, C, 3, dtype), MPN_ERR_BAD_SHAPE, "conv_bwd_data_bn: geometry not covered (K %d, C %d)", K, C);
    MPN_REQUIRE(N > 0, MPN_ERR_BAD_SHAPE, "conv_bwd_data_bn: bad shape");
    const PackGeom pg = pack_geom(K, C, 9, 2, dtype);
    if (pg.row_bytes == 128) {   // the tiled kernel (thin K: not a geometry of the persistent 3x3 kernel)
        MPN_REQUIRE(njobs <= kMaxGroup, MPN_ERR_BAD_ARG, "conv_bwd_data_bn: at most %d jobs", kMaxGroup);
        ConvGroup grp = {};
        int begin = 0;
        for (int j = 0; j < njobs; ++j) {
            MPN_REQUIRE(dy[j] && w_packed_t[j] && dx[j] && bn_x[j] && bn_scale[j] && bn_shift[j] && part[j] && H[j] > 0 && W[j] > 0,
                        MPN_ERR_BAD_ARG, "conv_bwd_data_bn: null pointer / bad size");
            MPN_REQUIRE(mpn_aligned16(dy[j]) && mpn_aligned16(w_packed_t[j]) && mpn_aligned16(dx[j]) && mpn_aligned16(bn_x[j]) &&
                            mpn_aligned16(bn_scale[j]) && mpn_aligned16(bn_shift[j]), MPN_ERR_BAD_ALIGN, "conv_bwd_data_bn: pointers must be 16-byte aligned");
            const int ys = dx_stride ? dx_stride[j] : 0, xs = dy_stride ? dy_stride[j] : 0, bs = bn_x_stride ? bn_x_stride[j] : 0;
            MPN_REQUIRE((ys == 0 || (ys >= C && ys % 8 == 0)) && (xs == 0 || (xs >= K && xs % 8 == 0)) && (bs == 0 || (bs >= C && bs % 8 == 0)),
                        MPN_ERR_BAD_SHAPE, "conv_bwd_data_bn: bad pixel strides %d, %d, %d", xs, ys, bs);
            conv_fill_params(grp.p[j], pg, dy[j], w_packed_t[j], dx[j], N, H[j], W[j], K, C, xs, ys, 3, nullptr, nullptr, MPN_ACT_NONE,
                             part[j], nullptr);
            grp.p[j].bnr_x = bn_x[j]; grp.p[j].bnr_scale = bn_scale[j]; grp.p[j].bnr_shift = bn_shift[j]; grp.p[j].bnr_act = bn_act;
            grp.p[j].bnr_xs = bs > 0 ? bs : C;
            grp.p[j].xcd_remap = 0;
            grp.begin[j] = begin;
            begin += mpn_conv_num_parts(N, H[j], W[j], 3) * pg.n_tiles;
        }
        for (int j = njobs; j <= kMaxGroup; ++j) grp.begin[j] = begin;
        grp.njobs = njobs;
        return pg.BN == 128 ? launch_conv_grouped<128, true>(grp, begin, (hipStream_t)stream) : launch_conv_grouped<64, true>(grp, begin, (hipStream_t)stream);
    }
    mpn_c3::Job jobs[mpn_c3::kMaxJobs];
    for (int j = 0; j < njobs; ++j) {
        MPN_REQUIRE(dy[j] && w_packed_t[j] && dx[j] && bn_x[j] && bn_scale[j] && bn_shift[j] && part[j] && H[j] > 0 && W[j] > 0,
                    MPN_ERR_BAD_ARG, "conv_bwd_data_bn: null pointer / bad size");
        MPN_REQUIRE(mpn_aligned16(dy[j]) && mpn_aligned16(w_packed_t[j]) && mpn_aligned16(dx[j]) && mpn_aligned16(bn_x[j]), MPN_ERR_BAD_ALIGN,
                    "conv_bwd_data_bn: pointers must be 16-byte aligned");
        const int ys = dx_stride ? dx_stride[j] : 0, xs = dy_stride ? dy_stride[j] : 0, bs = bn_x_stride ? bn_x_stride[j] : 0;
        MPN_REQUIRE((ys == 0 || (ys >= C && ys % 8 == 0)) && (xs == 0 || (xs >= K && xs % 8 == 0)) && (bs == 0 || (bs >= C && bs % 8 == 0)),
                    MPN_ERR_BAD_SHAPE, "conv_bwd_data_bn: bad pixel strides %d, %d, %d", xs, ys, bs);
        c3_fill_job(jobs[j], dy[j], w_packed_t[j], dx[j], N, H[j], W[j], K, C, xs, ys, nullptr, nullptr, MPN_ACT_NONE, part[j]);
        jobs[j].bnr_x = bn_x[j]; jobs[j].bnr_scale = bn_scale[j]; jobs[j].bnr_shift = bn_shift[j]; jobs[j].bnr_act = bn_act;
        jobs[j].bnr_xs = bs > 0 ? bs : C;
    }
    return mpn_c3::launch(jobs, njobs, dtype, (hipStream_t)stream);
}

/* One layer (1x1 through the GEMM kernel of pointwise.hip: the data gradients of Conv2d_5..13_pointwise, which feed the
 * depthwise layers' batch-norms, mobilenet_v1.py:66-74; or 3x3): dy [N,H,W,K] -> dx [N,H,W,C] masked, part
 * [mpn_conv_num_parts(N,H,W,ksize)][2][C] = partial sums of g and g * bn_x (raw x). */
extern "C" int mpn_conv_bwd_data_bn(const void* dy, const void* w_packed_t, void* dx, int N, int H, int W, int K, int C,
                                    int dy_stride, int dx_stride, int ksize, int dtype, const void* bn_x, int bn_x_stride,
                                    const float* bn_scale, const float* bn_shift, int bn_act, float* part, mpn_stream_t stream) {
    MPN_REQUIRE(mpn_conv_bwd_data_bn_supported(K, C, ksize, dtype), MPN_ERR_BAD_SHAPE, "conv_bwd_data_bn: geometry not covered (K %d, C %d, k %d)", K, C, ksize);
    if (ksize == 3) {
        const int h = H, w = W;
        return mpn_conv_bwd_data_bn_grouped(1, &dy, &w_packed_t, &dx, N, &h, &w, K, C, &dy_stride, &dx_stride, dtype, &bn_x, &bn_x_stride,
                                            &bn_scale, &bn_shift, bn_act, &part, stream);
    }
    MPN_REQUIRE(dy && w_packed_t && dx && bn_x && bn_scale && bn_shift && part && N > 0 && H > 0 && W > 0, MPN_ERR_BAD_ARG, "conv_bwd_data_bn: bad arguments");
    MPN_REQUIRE(mpn_aligned16(dy) && mpn_aligned16(w_packed_t) && mpn_aligned16(dx) && mpn_aligned16(bn_x) && mpn_aligned16(bn_scale) && mpn_aligned16(bn_shift),
                MPN_ERR_BAD_ALIGN, "conv_bwd_data_bn: pointers must be 16-byte aligned");
    MPN_REQUIRE((dy_stride == 0 || (dy_stride >= K && dy_stride % 8 == 0)) && (dx_stride == 0 || (dx_stride >= C && dx_stride % 8 == 0)) &&
                (bn_x_stride == 0 || (bn_x_stride >= C && bn_x_stride % 8 == 0)), MPN_ERR_BAD_SHAPE, "conv_bwd_data_bn: bad pixel strides");
    if (pw_gemm_eligible(K, C, 1, 2))
        return pw_gemm_launch(dy, w_packed_t, dx, (long long)N * H * W, K, C, dy_stride > 0 ? dy_stride : K, dx_stride > 0 ? dx_stride : C, nullptr, nullptr,
                              MPN_ACT_NONE, part, (hipStream_t)stream, bn_x, bn_x_stride > 0 ? bn_x_stride : C, bn_scale, bn_shift, bn_act);
    const PackGeom g = pack_geom(K, C, 1, 2, dtype);
    ConvParams p;
    conv_fill_params(p, g, dy, w_packed_t, dx, N, H, W, K, C, dy_stride, dx_stride, 1, nullptr, nullptr, MPN_ACT_NONE, part, nullptr);
    p.bnr_x = bn_x; p.bnr_scale = bn_scale; p.bnr_shift = bn_shift; p.bnr_act = bn_act; p.bnr_xs = bn_x_stride > 0 ? bn_x_stride : C;
    const int m_tiles = mpn_conv_num_parts(N, H, W, 1);
    return g.BN == 128 ? launch_conv_rb<bf16_t, 1, 128, 128, true>(p, m_tiles, (hipStream_t)stream)
                       : launch_conv_rb<bf16_t, 1, 64, 128, true>(p, m_tiles, (hipStream_t)stream);
}
