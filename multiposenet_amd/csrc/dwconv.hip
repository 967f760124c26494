// K3: depthwise 3x3 convolution (TF 'SAME' padding, stride 1 or 2), NHWC, fwd / dgrad / wgrad.
// Replaces tf.nn.depthwise_conv2d at detector/backbones/mobilenet_v1.py:101 (weights
// `depthwise_weights` [3,3,C,1] == [9][C] in memory) and its gradients.
//
// HBM-bound (9 MAC per 2-4 bytes): every input element is fetched from HBM once per tile with
// 16-byte channel-vector loads that are coalesced across the lanes of a pixel (NHWC), the
// producer's batch-norm affine + ReLU6 is applied on the way into an LDS halo tile (f32), and
// each thread then produces 16 bytes of output channels for a few pixels out of LDS.
// The following batch-norm's partial statistics come out of the same pass (wave shuffles over
// the lanes that share a channel vector, then one LDS hop across the 4 waves).
#include <type_traits>
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int STRIDE> struct DwTile;
template <> struct DwTile<1> { static constexpr int TH = 8, TW = 16, HH = 10, HW = 18; };
template <> struct DwTile<2> { static constexpr int TH = 4, TW = 8, HH = 9, HW = 17; };

struct DwParams {
    const void* x;      // input  [N,H,W,C]
    const float* w;     // [9][C]
    void* y;            // output [N,OH,OW,C]
    const void* dy;     // wgrad: output gradient [N,OH,OW,C]
    float* part;        // fwd: stats partials [nparts][2][C]; wgrad: [nblk][9][C]
    const float* in_scale;
    const float* in_shift;
    int in_act;
    int flip;           // use w[8-t] (stride-1 dgrad)
    int N, H, W, C, OH, OW;
    int pad_t, pad_l;
    int tiles_x, tiles_y;
    int nvg;            // channel vectors handled per block (<= 8)
    int cblocks;        // channel blocks
    // data-gradient kernels with the batch-norm backward reduction of the layer they feed fused in (BNR): the output is
    // dA of that layer, bnr_x its raw conv output; part then receives sum(g), sum(g * xhat) instead of the forward statistics
    const void* bnr_x;
    const float* bnr_scale;
    const float* bnr_shift;
    const float* bnr_mean;
    const float* bnr_invstd;
    int bnr_act;
    const void* addend; // stride-2 sliding-window data gradient: a tensor of dx's shape added before the store (and before BNR)
    int xcd_remap;      // sliding-window kernels: XCD-aware block -> strip map
    float* wpart;       // fused backward (dwconv_bwd_sw2_kernel): the weight gradient's partial slab [units][9][C] (part: the fused reduction's)
    int swr;            // forward sliding window: output rows per strip (0: sw_rows(OH)); shorter strips where a launch without
                        // statistics would leave CUs idle (batch-1 inference: 13 launches of ~16 us each at 640 x 640)
};

// 4-channel (one LDS float4) accessors of the storage type: the COMPUTE granule. 72 weight registers per thread
// (9 taps x 8 channels) pushed the 8-channel version to 2 waves/SIMD; 4 channels per lane need 36 and keep every
// global access a contiguous 8/16-byte piece of a fully used line.
__device__ __forceinline__ void load4(const float* p, float (&f)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&f)[4]) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xffff0000u);
    f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xffff0000u);
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

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// two packed channel pairs -> 4 consecutive channels of the storage type (one v_cvt_pk_bf16_f32 per pair)
__device__ __forceinline__ void store4x2(float* p, f32x2_t a, f32x2_t b) {
    *reinterpret_cast<float4*>(p) = make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void store4x2(bf16_t* p, f32x2_t a, f32x2_t b) {
    const bf16x2_t lo = __builtin_convertvector(a, bf16x2_t), hi = __builtin_convertvector(b, bf16x2_t);
    uint2 q;
    q.x = __builtin_bit_cast(unsigned, lo);
    q.y = __builtin_bit_cast(unsigned, hi);
    *reinterpret_cast<uint2*>(p) = q;
}

// XCD-aware work id: the dispatcher deals consecutive block ids round-robin over the 8 XCDs (each with its own L2), so
// blocks that share an XCD (same id % 8) get a contiguous range of work ids - neighbouring strips, which share halo
// columns and rows, then hit in one L2 instead of fetching the halo once per XCD. Bijective for any grid size.
__device__ __forceinline__ int xcd_work_id(int remap) {
    const int wid = blockIdx.x;
    if (!remap) return wid;
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wid >> 3);
}

// the value a consumer reads back after the store (bf16 storage rounds, f32 does not)
template <typename T> __device__ __forceinline__ f32x2_t round_storage(f32x2_t a);
template <> __device__ __forceinline__ f32x2_t round_storage<float>(f32x2_t a) { return a; }
template <> __device__ __forceinline__ f32x2_t round_storage<bf16_t>(f32x2_t a) {
    return __builtin_convertvector(__builtin_convertvector(a, bf16x2_t), f32x2_t);
}

// reduce over the lanes/waves that share this thread's channel vector; result valid in threads
// with pt == 0 (pixel lane 0). nvg is a power of two <= 8.
template <int NV>
__device__ __forceinline__ void reduce_same_vg(float (&v)[NV], int nvg, float* smem /*[4][16][NV]*/) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
        for (int o = nvg; o < 64; o <<= 1) v[k] += __shfl_xor(v[k], o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane < nvg)
#pragma unroll
        for (int k = 0; k < NV; ++k) smem[(wave * 16 + lane) * NV + k] = v[k];
    __syncthreads();
    if ((int)threadIdx.x < nvg)
#pragma unroll
        for (int k = 0; k < NV; ++k)
            v[k] = smem[(0 * 16 + lane) * NV + k] + smem[(1 * 16 + lane) * NV + k] + smem[(2 * 16 + lane) * NV + k] +
                   smem[(3 * 16 + lane) * NV + k];
}

// stage the (affine+activated) input halo tile of one (image, tile, channel block) into LDS as f32
template <typename T> struct HaloAffine {
    float sc[Vec16<T>::N], sh[Vec16<T>::N];
    bool on;
};

// per-thread affine of the staging lane (the lane's channel vector is the same for every vector it stages,
// because kThreads % nvg == 0) - loaded once per kernel
template <typename T>
__device__ __forceinline__ HaloAffine<T> load_halo_affine(const DwParams& p, int c0, int cb_vecs) {
    constexpr int VE = Vec16<T>::N;
    HaloAffine<T> a;
    const int vg = threadIdx.x % p.nvg;
    a.on = p.in_scale != nullptr;
#pragma unroll
    for (int j = 0; j < VE; ++j) {
        const bool ok = a.on && vg < cb_vecs;
        a.sc[j] = ok ? p.in_scale[c0 + vg * VE + j] : 1.f;
        a.sh[j] = ok ? p.in_shift[c0 + vg * VE + j] : 0.f;
    }
    return a;
}

// Halo staging split in two so that a persistent block can PREFETCH: `load` issues all of a tile's global loads into
// registers (before the previous tile is computed), `commit` applies the affine + activation and writes the f32 LDS
// tile (after the previous tile's LDS reads are done). The HBM round trip hides under the previous tile's math.
template <typename T, int STRIDE> struct HaloRegs {
    using TL = DwTile<STRIDE>;
    static constexpr int MAXV = (TL::HH * TL::HW * 8 + kThreads - 1) / kThreads;
    Vec16<T> v[MAXV];
    unsigned okmask;

    __device__ __forceinline__ void load(const DwParams& p, int img, int oy0, int ox0, int c0, int cb_vecs) {
        constexpr int VE = Vec16<T>::N;
        const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
        const int iy0 = oy0 * STRIDE - p.pad_t, ix0 = ox0 * STRIDE - p.pad_l;
        // decode from an OPAQUE copy of the thread index: hoisted out of the tile loop the per-vector pixel decode
        // costs ~20 registers that hipcc spills (and a scratch reload waits on vmcnt, i.e. on the prefetch)
        int tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const int vg = tid_ % p.nvg;
        const int hp0 = tid_ / p.nvg, hstep = kThreads / p.nvg;
        okmask = 0u;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int hp = hp0 + k * hstep;
            const int hy = hp / TL::HW, hx = hp - hy * TL::HW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = hp < TL::HH * TL::HW && vg < cb_vecs && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            // UNPREDICATED load from a clamped address (zeroed at commit time): `if (ok) load else zero` is lowered to
            // load + select, which waits for the data right here - the prefetch then hides nothing
            v[k].load(x + (ok ? (((long long)img * p.H + iy) * p.W + ix) * p.C + c0 + vg * VE : 0));
            okmask |= (ok ? 1u : 0u) << k;
        }
    }

    // affine + activation on in-image elements only (the padding is zeros of the ACTIVATED tensor) -> LDS
    __device__ __forceinline__ void commit(const DwParams& p, const HaloAffine<T>& aff, float* tile) {
        constexpr int VE = Vec16<T>::N;
        int tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const int vg = tid_ % p.nvg;
        const int hp0 = tid_ / p.nvg, hstep = kThreads / p.nvg;
        const float lo = (p.in_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
        const float hi = (p.in_act == MPN_ACT_RELU6) ? 6.f : INFINITY;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int hp = hp0 + k * hstep;
            if (hp < TL::HH * TL::HW) {
                float f[VE];
                v[k].unpack(f);
                const bool ok = (okmask >> k) & 1u;
                if (aff.on) {
#pragma unroll
                    for (int j = 0; j < VE; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * aff.sc[j] + aff.sh[j], lo, hi);
                }
                if (!ok) {
#pragma unroll
                    for (int j = 0; j < VE; ++j) f[j] = 0.f;
                }
                float* dst = tile + (hp * p.nvg + vg) * VE;
#pragma unroll
                for (int j = 0; j < VE; j += 4)
                    *reinterpret_cast<float4*>(dst + j) = make_float4(f[j], f[j + 1], f[j + 2], f[j + 3]);
            }
        }
    }
};

__device__ __forceinline__ void dw_tile_origin(const DwParams& p, int t, int th, int tw, int& img, int& oy0, int& ox0) {
    const int tx = t % p.tiles_x;
    const int t2 = t / p.tiles_x;
    const int ty = t2 % p.tiles_y;
    img = t2 / p.tiles_y;
    oy0 = ty * th;
    ox0 = tx * tw;
}

// ---------------------------------------------------------------------------------------------------------------
// Forward, register sliding window ("sw"): no LDS tile, no barrier in the loop. A thread owns 4 channels of ONE output
// column and walks down a strip of sw_rows() output rows; the 3x3 window of activated inputs lives in registers, each step
// loads the 3 (stride 1) or 6 (stride 2) new 8-byte pieces straight from global memory (neighbouring columns overlap
// and hit in L1), applies the producer's batch-norm affine + activation, and emits one 4-channel output. Memory latency
// is hidden by occupancy (~70 registers) and by requesting the next row's pieces before the current row is multiplied.
// Lanes run over (column, 4-channel group) with channels fastest, so every wave access is a contiguous run of pixels.
// output rows per thread: 32 on maps of 64 rows and more (the two primed rows of a strip are 6 % instead of 12 % extra
// loads; the step -0.8 %), 16 below (strips of 32 would leave the 32x32 and 16x16 maps with too few blocks)
__host__ __device__ inline int sw_rows(int OH, int stride = 2) {
    (void)stride;   // (64-row strips for stride 1 on the 128 / 256-row maps: no gain in the forward)
    return OH >= 64 ? 32 : 16;
}

template <typename T> struct Raw4;
template <> struct Raw4<float> { float4 v; };
template <> struct Raw4<bf16_t> { uint2 v; };
__device__ __forceinline__ void raw_load(Raw4<float>& r, const float* p) { r.v = *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void raw_load(Raw4<bf16_t>& r, const bf16_t* p) { r.v = *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ void raw_unpack(const Raw4<float>& r, float (&f)[4]) { f[0] = r.v.x; f[1] = r.v.y; f[2] = r.v.z; f[3] = r.v.w; }
__device__ __forceinline__ void raw_unpack(const Raw4<bf16_t>& r, float (&f)[4]) {
    f[0] = __uint_as_float(r.v.x << 16); f[1] = __uint_as_float(r.v.x & 0xffff0000u);
    f[2] = __uint_as_float(r.v.y << 16); f[3] = __uint_as_float(r.v.y & 0xffff0000u);
}

// NOAFF: no producer batch-norm on the input (the kernel as a data gradient over dY): the affine + clamp of every loaded
// element compiles away (24 VALU operations and 8 registers per row)
template <typename T, int STRIDE, bool BNR = false, bool NOAFF = false>
__global__ __launch_bounds__(kThreads) void dwconv_fwd_sw_kernel(const DwParams p, int ncg, int cols, int xblocks, int yblocks) {
    static_assert(!BNR || (STRIDE == 1 && NOAFF), "the fused batch-norm backward reduction rides on the stride-1 data gradient");
    __shared__ float red[kThreads * 8];
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    T* __restrict__ y = reinterpret_cast<T*>(p.y);
    int b = xcd_work_id(p.xcd_remap);
    const int xb = b % xblocks; b /= xblocks;
    const int yb = b % yblocks; b /= yblocks;
    const int cgb = b % p.cblocks;
    const int img = b / p.cblocks;
    const int cgl = threadIdx.x % ncg, col = threadIdx.x / ncg;       // 4-channel group inside the block, column
    const int c = (cgb * ncg + cgl) * 4;
    const int ox = xb * cols + col;
    const bool lane_ok = c < p.C && ox < p.OW && col < cols;
    const int cc = lane_ok ? c : 0;
    float wr[9][4], sc[4], sh[4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[t][j] = p.w[(p.flip ? 8 - t : t) * p.C + cc + j];
    const bool aff = p.in_scale != nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = aff ? p.in_scale[cc + j] : 1.f; sh[j] = aff ? p.in_shift[cc + j] : 0.f; }
    const float lo = (aff && p.in_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (aff && p.in_act == MPN_ACT_RELU6) ? 6.f : INFINITY;

    const int swr = p.swr > 0 ? p.swr : sw_rows(p.OH, STRIDE);
    const int oy_begin = yb * swr, oy_end = min(oy_begin + swr, p.OH);
    const int ix0 = ox * STRIDE - p.pad_l;                            // leftmost input column of the window
    const T* ximg = x + (long long)img * p.H * p.W * p.C + cc;
    bool xok[3];
    int xoff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ix = ix0 + k;
        xok[k] = lane_ok && ix >= 0 && ix < p.W;
        xoff[k] = (xok[k] ? ix : 0) * p.C;
    }
    // one input row: 3 raw pieces (unpredicated loads from clamped addresses) -> activated f32, zero outside the image
    auto row_load = [&](Raw4<T> (&r)[3], int iy) {
        const int iyc = min(max(iy, 0), p.H - 1);
        const T* rowp = ximg + (long long)iyc * p.W * p.C;
#pragma unroll
        for (int k = 0; k < 3; ++k) raw_load(r[k], rowp + xoff[k]);
    };
    // (row validity is uniform over the block: a scalar branch; only the outer columns of the window can fall outside
    //  the image in x, so the centre column of a stride-1 window needs no select at all)
    auto row_act = [&](const Raw4<T> (&r)[3], int iy, f32x2_t (&a)[3][2]) {
        if (iy < 0 || iy >= p.H) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { a[k][0] = (f32x2_t){0.f, 0.f}; a[k][1] = (f32x2_t){0.f, 0.f}; }
            return;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float f[4];
            raw_unpack(r[k], f);
            if constexpr (!NOAFF) {
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * sc[j] + sh[j], lo, hi);
            }
            if (STRIDE != 1 || k != 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = xok[k] ? f[j] : 0.f;
            }
            a[k][0] = (f32x2_t){f[0], f[1]};
            a[k][1] = (f32x2_t){f[2], f[3]};
        }
    };
    f32x2_t s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
    f32x2_t w01[9], w23[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { w01[t] = (f32x2_t){wr[t][0], wr[t][1]}; w23[t] = (f32x2_t){wr[t][2], wr[t][3]}; }

    f32x2_t r0[3][2], r1[3][2], r2[3][2];                             // window rows (activated), [column][channel pair]
    T* yp = y + (((long long)img * p.OH + oy_begin) * p.OW + ox) * p.C + cc;
    const long long ystep = (long long)p.OW * p.C;
    // BNR: per-channel constants of the batch-norm whose input gradient this kernel produces
    f32x2_t bsc01 = {0.f, 0.f}, bsc23 = {0.f, 0.f}, bsh01 = {0.f, 0.f}, bsh23 = {0.f, 0.f};
    f32x2_t bis01 = {0.f, 0.f}, bis23 = {0.f, 0.f}, bnm01 = {0.f, 0.f}, bnm23 = {0.f, 0.f};
    float blo = -INFINITY, bhi = INFINITY;
    const T* bxp = nullptr;
    if constexpr (BNR) {
        const float4 s4 = *reinterpret_cast<const float4*>(p.bnr_scale + cc), h4 = *reinterpret_cast<const float4*>(p.bnr_shift + cc);
        const float4 m4 = *reinterpret_cast<const float4*>(p.bnr_mean + cc), i4 = *reinterpret_cast<const float4*>(p.bnr_invstd + cc);
        bsc01 = (f32x2_t){s4.x, s4.y}; bsc23 = (f32x2_t){s4.z, s4.w};
        bsh01 = (f32x2_t){h4.x, h4.y}; bsh23 = (f32x2_t){h4.z, h4.w};
        bis01 = (f32x2_t){i4.x, i4.y}; bis23 = (f32x2_t){i4.z, i4.w};
        bnm01 = (f32x2_t){-m4.x * i4.x, -m4.y * i4.y}; bnm23 = (f32x2_t){-m4.z * i4.z, -m4.w * i4.w};
        blo = (p.bnr_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
        bhi = (p.bnr_act == MPN_ACT_RELU6) ? 6.f : INFINITY;
        bxp = reinterpret_cast<const T*>(p.bnr_x) + ((long long)img * p.OH * p.OW + (lane_ok ? ox : 0)) * p.C + cc;
    }
    auto bnr_load = [&](Raw4<T>& r, int oy) {
        if constexpr (BNR) raw_load(r, bxp + (long long)min(oy, p.OH - 1) * p.OW * p.C);
    };
    auto emit = [&](const f32x2_t (&a)[3][2], const f32x2_t (&bb)[3][2], const f32x2_t (&cr)[3][2], const Raw4<T>& yr) {
        f32x2_t a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a01 += a[k][0] * w01[k];      a23 += a[k][1] * w23[k];
            a01 += bb[k][0] * w01[3 + k]; a23 += bb[k][1] * w23[3 + k];
            a01 += cr[k][0] * w01[6 + k]; a23 += cr[k][1] * w23[6 + k];
        }
        if (lane_ok) {
            if constexpr (BNR) {
                // g = dA * act'(x * scale + shift), computed from the ROUNDED dA the separate reduction would read
                float f[4];
                raw_unpack(yr, f);
                const f32x2_t x01 = {f[0], f[1]}, x23 = {f[2], f[3]};
                const f32x2_t p01 = x01 * bsc01 + bsh01, p23 = x23 * bsc23 + bsh23;
                const f32x2_t d01 = round_storage<T>(a01), d23 = round_storage<T>(a23);
                f32x2_t g01, g23;
                g01.x = (p01.x > blo && p01.x < bhi) ? d01.x : 0.f; g01.y = (p01.y > blo && p01.y < bhi) ? d01.y : 0.f;
                g23.x = (p23.x > blo && p23.x < bhi) ? d23.x : 0.f; g23.y = (p23.y > blo && p23.y < bhi) ? d23.y : 0.f;
                s01 += g01; s23 += g23;
                q01 += g01 * (x01 * bis01 + bnm01); q23 += g23 * (x23 * bis23 + bnm23);
            } else {
                s01 += a01; s23 += a23;
                q01 += a01 * a01; q23 += a23 * a23;
            }
            store4x2(yp, a01, a23);
        }
        yp += ystep;
    };
    if (STRIDE == 1) {
        // rows iy = oy-1, oy, oy+1: prime two rows, then one new row per output row (unrolled by 3: the window registers
        // rotate roles). The load of the following row is issued right after the current one is activated and flies
        // while the output row is multiplied and stored.
        // THREE raw row buffers rotate with the window: a buffer is re-requested (3 rows ahead) as soon as it has been
        // activated, so three rows (9 pieces = 72 bytes per thread, ~70 KB per CU) are always in flight - with one row in
        // flight the kernel ran at 3.4 TB/s, exactly what its bytes in flight allow.
        Raw4<T> ra[3], rb[3], rc[3], ya, yb, yc;
        int iy = oy_begin - p.pad_t;
        row_load(ra, iy);
        row_load(rb, iy + 1);
        row_load(rc, iy + 2);
        bnr_load(ya, oy_begin);
        bnr_load(yb, oy_begin + 1);
        bnr_load(yc, oy_begin + 2);
        row_act(ra, iy, r0);
        row_load(ra, iy + 3);
        row_act(rb, iy + 1, r1);
        row_load(rb, iy + 4);
        iy += 2;
        for (int oy = oy_begin; oy < oy_end; oy += 3, iy += 3) {
            row_act(rc, iy, r2);
            row_load(rc, iy + 3);
            emit(r0, r1, r2, ya);
            bnr_load(ya, oy + 3);
            if (oy + 1 < oy_end) {
                row_act(ra, iy + 1, r0);
                row_load(ra, iy + 4);
                emit(r1, r2, r0, yb);
                bnr_load(yb, oy + 4);
            }
            if (oy + 2 < oy_end) {
                row_act(rb, iy + 2, r1);
                row_load(rb, iy + 5);
                emit(r2, r0, r1, yc);
                bnr_load(yc, oy + 5);
            }
        }
    } else {
        // rows iy = 2*oy - pad_t + {0,1,2}; consecutive outputs share one row (the third becomes the first)
        Raw4<T> ra[3], rb[3];
        int iy = oy_begin * 2 - p.pad_t;
        row_load(ra, iy);
        row_act(ra, iy, r0);
        row_load(ra, iy + 1);
        row_load(rb, iy + 2);
        for (int oy = oy_begin; oy < oy_end; oy += 2, iy += 4) {
            row_act(ra, iy + 1, r1);
            row_act(rb, iy + 2, r2);
            row_load(ra, iy + 3);
            row_load(rb, iy + 4);
            emit(r0, r1, r2, ra[0]);
            if (oy + 1 < oy_end) {
                row_act(ra, iy + 3, r1);
                row_act(rb, iy + 4, r0);
                row_load(ra, iy + 5);
                row_load(rb, iy + 6);
                emit(r2, r1, r0, ra[0]);
            }
        }
    }
    if (p.part != nullptr) {
        // partial row of this block: sum over the block's columns per 4-channel group (fixed order)
        float st[8] = {s01.x, s01.y, s23.x, s23.y, q01.x, q01.y, q23.x, q23.y};
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = lane_ok ? st[j] : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < ncg && (cgb * ncg + (int)threadIdx.x) * 4 < p.C) {
            float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int cidx = 0; cidx < cols; ++cidx)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[j] += red[(cidx * ncg + threadIdx.x) * 8 + j];
            const int prow = (img * yblocks + yb) * xblocks + xb;
            float* dst = p.part + (long long)prow * 2 * p.C + (cgb * ncg + threadIdx.x) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { dst[j] = acc8[j]; dst[p.C + j] = acc8[4 + j]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-1 sliding window with TWO output columns per thread: the window is 3 rows x 4 columns, so an input element is
// unpacked and activated by 2 threads instead of 3 and loaded 4/2 instead of 3/1 times per output (the one-column kernel
// is bound by exactly that vector work: 3.7 TB/s with the producer's affine, 4.9 without). Same walk, same rotation of
// three raw row buffers, same statistics / fused batch-norm reduction (BNR) as dwconv_fwd_sw_kernel.
template <typename T, bool BNR, bool NOAFF>
__global__ __launch_bounds__(kThreads) void dwconv_fwd_sw2_kernel(const DwParams p, int ncg, int cols, int xblocks, int yblocks) {
    static_assert(!BNR || NOAFF, "the fused batch-norm backward reduction rides on the data gradient");
    __shared__ float red[kThreads * 8];
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    T* __restrict__ y = reinterpret_cast<T*>(p.y);
    int b = xcd_work_id(p.xcd_remap);
    const int xb = b % xblocks; b /= xblocks;
    const int yb = b % yblocks; b /= yblocks;
    const int cgb = b % p.cblocks;
    const int img = b / p.cblocks;
    const int cgl = threadIdx.x % ncg, col = threadIdx.x / ncg;       // 4-channel group inside the block, column PAIR
    const int c = (cgb * ncg + cgl) * 4;
    const int ox = (xb * cols + col) * 2;                             // first of the two output columns
    const bool ok0 = c < p.C && ox < p.OW && col < cols;
    const bool ok1 = ok0 && ox + 1 < p.OW;
    const int cc = ok0 ? c : 0;
    float sc[4], sh[4];
    f32x2_t w01[9], w23[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 q = *reinterpret_cast<const float4*>(p.w + (p.flip ? 8 - t : t) * p.C + cc);
        w01[t] = (f32x2_t){q.x, q.y};
        w23[t] = (f32x2_t){q.z, q.w};
    }
    const bool aff = p.in_scale != nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = aff ? p.in_scale[cc + j] : 1.f; sh[j] = aff ? p.in_shift[cc + j] : 0.f; }
    const float lo = (aff && p.in_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (aff && p.in_act == MPN_ACT_RELU6) ? 6.f : INFINITY;

    const int swr = p.swr > 0 ? p.swr : sw_rows(p.OH, 1);
    const int oy_begin = yb * swr, oy_end = min(oy_begin + swr, p.OH);
    const int ix0 = ox - p.pad_l;                                     // leftmost input column of the 4-column window
    const T* ximg = x + (long long)img * p.H * p.W * p.C + cc;
    bool xok[4];
    int xoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ix = ix0 + k;
        xok[k] = ok0 && ix >= 0 && ix < p.W;
        xoff[k] = (xok[k] ? ix : 0) * p.C;
    }
    // every lane of the wave has its whole 4-column window inside the image (all but the waves that touch the left / right
    // border): the twelve zero-selects per row are skipped - the walk is co-limited by its vector instruction stream (115 per row
    // step: profiles/r04_dwconv_traffic.json) - on a wave-uniform branch around the whole walk
    const bool interior = __all(xok[0] && xok[2] && xok[3]) != 0;
    // (measured and not adopted: buffer loads with the image as a wave-uniform descriptor, the row as a scalar offset and a 32-bit
    //  per-lane offset - seven 64-bit vector address instructions per row less, and 2-6 % SLOWER on cold tensors)
    auto row_load = [&](Raw4<T> (&r)[4], int iy) {
        const int iyc = min(max(iy, 0), p.H - 1);
        const T* rowp = ximg + (long long)iyc * p.W * p.C;
#pragma unroll
        for (int k = 0; k < 4; ++k) raw_load(r[k], rowp + xoff[k]);
    };
    auto row_act = [&](auto edge, const Raw4<T> (&r)[4], int iy, f32x2_t (&a)[4][2]) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge)::value;
        if (iy < 0 || iy >= p.H) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k][0] = (f32x2_t){0.f, 0.f}; a[k][1] = (f32x2_t){0.f, 0.f}; }
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[4];
            raw_unpack(r[k], f);
            if constexpr (!NOAFF) {
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * sc[j] + sh[j], lo, hi);
            }
            if (k != 1 && EDGE) {   // (column 1 = the first output's own column: inside the image whenever the lane is)
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = xok[k] ? f[j] : 0.f;
            }
            a[k][0] = (f32x2_t){f[0], f[1]};
            a[k][1] = (f32x2_t){f[2], f[3]};
        }
    };
    f32x2_t s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
    f32x2_t r0[4][2], r1[4][2], r2[4][2];
    T* yp = y + (((long long)img * p.OH + oy_begin) * p.OW + (ok0 ? ox : 0)) * p.C + cc;
    const long long ystep = (long long)p.OW * p.C;
    f32x2_t bsc01 = {0.f, 0.f}, bsc23 = {0.f, 0.f}, bsh01 = {0.f, 0.f}, bsh23 = {0.f, 0.f};
    f32x2_t bis01 = {0.f, 0.f}, bis23 = {0.f, 0.f}, bnm01 = {0.f, 0.f}, bnm23 = {0.f, 0.f};
    float blo = -INFINITY, bhi = INFINITY;
    const T* bxp = nullptr;
    if constexpr (BNR) {
        const float4 s4 = *reinterpret_cast<const float4*>(p.bnr_scale + cc), h4 = *reinterpret_cast<const float4*>(p.bnr_shift + cc);
        const float4 m4 = *reinterpret_cast<const float4*>(p.bnr_mean + cc), i4 = *reinterpret_cast<const float4*>(p.bnr_invstd + cc);
        bsc01 = (f32x2_t){s4.x, s4.y}; bsc23 = (f32x2_t){s4.z, s4.w};
        bsh01 = (f32x2_t){h4.x, h4.y}; bsh23 = (f32x2_t){h4.z, h4.w};
        bis01 = (f32x2_t){i4.x, i4.y}; bis23 = (f32x2_t){i4.z, i4.w};
        bnm01 = (f32x2_t){-m4.x * i4.x, -m4.y * i4.y}; bnm23 = (f32x2_t){-m4.z * i4.z, -m4.w * i4.w};
        blo = (p.bnr_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
        bhi = (p.bnr_act == MPN_ACT_RELU6) ? 6.f : INFINITY;
        bxp = reinterpret_cast<const T*>(p.bnr_x) + ((long long)img * p.OH * p.OW + (ok0 ? ox : 0)) * p.C + cc;
    }
    const int off1 = ok1 ? p.C : 0;                                   // second output column (clamped for the loads)
    auto bnr_load = [&](Raw4<T> (&r)[2], int oy) {
        if constexpr (BNR) {
            const T* q = bxp + (long long)min(oy, p.OH - 1) * p.OW * p.C;
            raw_load(r[0], q);
            raw_load(r[1], q + off1);
        }
    };
    auto account = [&](f32x2_t a01, f32x2_t a23, const Raw4<T>& yr) {
        if constexpr (BNR) {
            float f[4];
            raw_unpack(yr, f);
            const f32x2_t x01 = {f[0], f[1]}, x23 = {f[2], f[3]};
            const f32x2_t p01 = x01 * bsc01 + bsh01, p23 = x23 * bsc23 + bsh23;
            const f32x2_t d01 = round_storage<T>(a01), d23 = round_storage<T>(a23);
            f32x2_t g01, g23;
            g01.x = (p01.x > blo && p01.x < bhi) ? d01.x : 0.f; g01.y = (p01.y > blo && p01.y < bhi) ? d01.y : 0.f;
            g23.x = (p23.x > blo && p23.x < bhi) ? d23.x : 0.f; g23.y = (p23.y > blo && p23.y < bhi) ? d23.y : 0.f;
            s01 += g01; s23 += g23;
            q01 += g01 * (x01 * bis01 + bnm01); q23 += g23 * (x23 * bis23 + bnm23);
        } else {
            s01 += a01; s23 += a23;
            q01 += a01 * a01; q23 += a23 * a23;
        }
    };
    auto emit = [&](const f32x2_t (&a)[4][2], const f32x2_t (&bb)[4][2], const f32x2_t (&cr)[4][2], const Raw4<T> (&yr)[2]) {
        f32x2_t a01 = {0.f, 0.f}, a23 = {0.f, 0.f}, b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a01 += a[k][0] * w01[k];          a23 += a[k][1] * w23[k];
            a01 += bb[k][0] * w01[3 + k];     a23 += bb[k][1] * w23[3 + k];
            a01 += cr[k][0] * w01[6 + k];     a23 += cr[k][1] * w23[6 + k];
            b01 += a[k + 1][0] * w01[k];      b23 += a[k + 1][1] * w23[k];
            b01 += bb[k + 1][0] * w01[3 + k]; b23 += bb[k + 1][1] * w23[3 + k];
            b01 += cr[k + 1][0] * w01[6 + k]; b23 += cr[k + 1][1] * w23[6 + k];
        }
        if (ok0) {
            account(a01, a23, yr[0]);
            store4x2(yp, a01, a23);
        }
        if (ok1) {
            account(b01, b23, yr[1]);
            store4x2(yp + p.C, b01, b23);
        }
        yp += ystep;
    };
    // the walk, once per variant of the row activation (a wave-uniform branch around the whole loop: inside it hipcc turns the
    // condition back into selects)
    auto walk = [&](auto edge) __attribute__((always_inline)) {
        Raw4<T> ra[4], rb[4], rc[4], ya[2], yb2[2], yc[2];
        int iy = oy_begin - p.pad_t;
        row_load(ra, iy);
        row_load(rb, iy + 1);
        row_load(rc, iy + 2);
        bnr_load(ya, oy_begin);
        bnr_load(yb2, oy_begin + 1);
        bnr_load(yc, oy_begin + 2);
        row_act(edge, ra, iy, r0);
        row_load(ra, iy + 3);
        row_act(edge, rb, iy + 1, r1);
        row_load(rb, iy + 4);
        iy += 2;
        for (int oy = oy_begin; oy < oy_end; oy += 3, iy += 3) {
            row_act(edge, rc, iy, r2);
            row_load(rc, iy + 3);
            emit(r0, r1, r2, ya);
            bnr_load(ya, oy + 3);
            if (oy + 1 < oy_end) {
                row_act(edge, ra, iy + 1, r0);
                row_load(ra, iy + 4);
                emit(r1, r2, r0, yb2);
                bnr_load(yb2, oy + 4);
            }
            if (oy + 2 < oy_end) {
                row_act(edge, rb, iy + 2, r1);
                row_load(rb, iy + 5);
                emit(r2, r0, r1, yc);
                bnr_load(yc, oy + 5);
            }
        }
    };
    if (interior) walk(std::false_type{}); else walk(std::true_type{});
    if (p.part != nullptr) {
        float st[8] = {s01.x, s01.y, s23.x, s23.y, q01.x, q01.y, q23.x, q23.y};
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = ok0 ? st[j] : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < ncg && (cgb * ncg + (int)threadIdx.x) * 4 < p.C) {
            float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int cidx = 0; cidx < cols; ++cidx)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[j] += red[(cidx * ncg + threadIdx.x) * 8 + j];
            const int prow = (img * yblocks + yb) * xblocks + xb;
            float* dst = p.part + (long long)prow * 2 * p.C + (cgb * ncg + threadIdx.x) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { dst[j] = acc8[j]; dst[p.C + j] = acc8[4 + j]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient, register sliding window: the forward kernel's walk (a thread = 4 channels of one output column,
// the 3x3 window of activated inputs in registers, three raw rows in flight) with the output row's dY piece in place
// of the store: acc[tap] += window[tap] * dY, 18 packed FMAs per output pixel, 36 accumulators per thread. One block
// = one (image, row strip, column block, channel block) unit = one row of the partial slab; blocks are at most 128
// channels wide (ncg <= 32) so that small maps with many channels still make hundreds of blocks with small slabs.
template <typename T, int STRIDE>
__global__ __launch_bounds__(kThreads) void dwconv_wgrad_sw_kernel(const DwParams p, int ncg, int cols, int xblocks, int yblocks,
                                                                  int rows) {
    __shared__ __attribute__((aligned(16))) float red[9 * kThreads * 4];   // [tap][thread][4 channels]
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
    int b = xcd_work_id(p.xcd_remap);
    const int cgb = b % p.cblocks; b /= p.cblocks;
    const int unit = b;                                               // partial-slab row
    const int xb = b % xblocks; b /= xblocks;
    const int yb = b % yblocks;
    const int img = b / yblocks;
    const int cgl = threadIdx.x % ncg, col = threadIdx.x / ncg;
    const int c = (cgb * ncg + cgl) * 4;
    const int ox = xb * cols + col;
    const bool lane_ok = c < p.C && ox < p.OW && col < cols;
    const int cc = lane_ok ? c : 0;
    const int oxc = lane_ok ? ox : 0;
    float sc[4], sh[4];
    const bool aff = p.in_scale != nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = aff ? p.in_scale[cc + j] : 1.f; sh[j] = aff ? p.in_shift[cc + j] : 0.f; }
    const float lo = (aff && p.in_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (aff && p.in_act == MPN_ACT_RELU6) ? 6.f : INFINITY;

    const int oy_begin = yb * rows, oy_end = min(oy_begin + rows, p.OH);
    const int ix0 = oxc * STRIDE - p.pad_l;
    const T* ximg = x + (long long)img * p.H * p.W * p.C + cc;
    const T* dyimg = dy + ((long long)img * p.OH * p.OW + oxc) * p.C + cc;
    bool xok[3];
    int xoff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ix = ix0 + k;
        xok[k] = lane_ok && ix >= 0 && ix < p.W;
        xoff[k] = (xok[k] ? ix : 0) * p.C;
    }
    auto row_load = [&](Raw4<T> (&r)[3], int iy) {
        const int iyc = min(max(iy, 0), p.H - 1);
        const T* rowp = ximg + (long long)iyc * p.W * p.C;
#pragma unroll
        for (int k = 0; k < 3; ++k) raw_load(r[k], rowp + xoff[k]);
    };
    auto dy_load = [&](Raw4<T>& r, int oy) {
        const int oyc = min(oy, p.OH - 1);
        raw_load(r, dyimg + (long long)oyc * p.OW * p.C);
    };
    auto row_act = [&](const Raw4<T> (&r)[3], int iy, f32x2_t (&a)[3][2]) {
        if (iy < 0 || iy >= p.H) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { a[k][0] = (f32x2_t){0.f, 0.f}; a[k][1] = (f32x2_t){0.f, 0.f}; }
            return;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float f[4];
            raw_unpack(r[k], f);
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * sc[j] + sh[j], lo, hi);
            if (STRIDE != 1 || k != 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = xok[k] ? f[j] : 0.f;
            }
            a[k][0] = (f32x2_t){f[0], f[1]};
            a[k][1] = (f32x2_t){f[2], f[3]};
        }
    };
    f32x2_t a01[9], a23[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { a01[t] = (f32x2_t){0.f, 0.f}; a23[t] = (f32x2_t){0.f, 0.f}; }
    auto accum = [&](const f32x2_t (&ra_)[3][2], const f32x2_t (&rb_)[3][2], const f32x2_t (&rc_)[3][2], const Raw4<T>& d) {
        float g[4];
        raw_unpack(d, g);
        const f32x2_t g01 = {g[0], g[1]}, g23 = {g[2], g[3]};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a01[k] += ra_[k][0] * g01;     a23[k] += ra_[k][1] * g23;
            a01[3 + k] += rb_[k][0] * g01; a23[3 + k] += rb_[k][1] * g23;
            a01[6 + k] += rc_[k][0] * g01; a23[6 + k] += rc_[k][1] * g23;
        }
    };
    f32x2_t r0[3][2], r1[3][2], r2[3][2];
    if (STRIDE == 1) {
        Raw4<T> ra[3], rb[3], rc[3], da, db, dc;
        int iy = oy_begin - p.pad_t;
        row_load(ra, iy);
        row_load(rb, iy + 1);
        row_load(rc, iy + 2);
        dy_load(da, oy_begin);
        dy_load(db, oy_begin + 1);
        dy_load(dc, oy_begin + 2);
        row_act(ra, iy, r0);
        row_load(ra, iy + 3);
        row_act(rb, iy + 1, r1);
        row_load(rb, iy + 4);
        iy += 2;
        for (int oy = oy_begin; oy < oy_end; oy += 3, iy += 3) {
            row_act(rc, iy, r2);
            row_load(rc, iy + 3);
            accum(r0, r1, r2, da);
            dy_load(da, oy + 3);
            if (oy + 1 < oy_end) {
                row_act(ra, iy + 1, r0);
                row_load(ra, iy + 4);
                accum(r1, r2, r0, db);
                dy_load(db, oy + 4);
            }
            if (oy + 2 < oy_end) {
                row_act(rb, iy + 2, r1);
                row_load(rb, iy + 5);
                accum(r2, r0, r1, dc);
                dy_load(dc, oy + 5);
            }
        }
    } else {
        Raw4<T> ra[3], rb[3], da, db;
        int iy = oy_begin * 2 - p.pad_t;
        row_load(ra, iy);
        dy_load(da, oy_begin);
        dy_load(db, oy_begin + 1);
        row_act(ra, iy, r0);
        row_load(ra, iy + 1);
        row_load(rb, iy + 2);
        for (int oy = oy_begin; oy < oy_end; oy += 2, iy += 4) {
            row_act(ra, iy + 1, r1);
            row_act(rb, iy + 2, r2);
            row_load(ra, iy + 3);
            row_load(rb, iy + 4);
            accum(r0, r1, r2, da);
            dy_load(da, oy + 2);
            if (oy + 1 < oy_end) {
                row_act(ra, iy + 3, r1);
                row_act(rb, iy + 4, r0);
                row_load(ra, iy + 5);
                row_load(rb, iy + 6);
                accum(r2, r1, r0, db);
                dy_load(db, oy + 3);
            }
        }
    }
    // block reduction over the columns that share a 4-channel group (fixed order), one partial-slab row per block
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float4 v = make_float4(a01[t].x, a01[t].y, a23[t].x, a23[t].y);
        if (!lane_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&red[(t * kThreads + threadIdx.x) * 4]) = v;
    }
    __syncthreads();
    const int nch = ncg * 4;
    float* dst = p.part + (long long)unit * 9 * p.C + cgb * nch;
    for (int o = threadIdx.x; o < 9 * nch; o += kThreads) {
        const int t = o / nch, cj = o - t * nch;
        if (cgb * nch + cj < p.C) {
            float sum = 0.f;
            for (int cidx = 0; cidx < cols; ++cidx) sum += red[(t * kThreads + cidx * ncg) * 4 + cj];
            dst[t * p.C + cj] = sum;
        }
    }
}

// stride-1 weight gradient with TWO output columns per thread (3 x 4 window, as dwconv_fwd_sw2_kernel): the 36
// accumulators take both columns' products, an input element is activated by 2 threads instead of 3.
template <typename T>
__global__ __launch_bounds__(kThreads) void dwconv_wgrad_sw2_kernel(const DwParams p, int ncg, int cols, int xblocks, int yblocks,
                                                                   int rows) {
    __shared__ __attribute__((aligned(16))) float red[9 * kThreads * 4];   // [tap][thread][4 channels]
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
    int b = xcd_work_id(p.xcd_remap);
    const int cgb = b % p.cblocks; b /= p.cblocks;
    const int unit = b;                                               // partial-slab row
    const int xb = b % xblocks; b /= xblocks;
    const int yb = b % yblocks;
    const int img = b / yblocks;
    const int cgl = threadIdx.x % ncg, col = threadIdx.x / ncg;       // column PAIR
    const int c = (cgb * ncg + cgl) * 4;
    const int ox = (xb * cols + col) * 2;
    const bool ok0 = c < p.C && ox < p.OW && col < cols;
    const bool ok1 = ok0 && ox + 1 < p.OW;
    const int cc = ok0 ? c : 0;
    const int oxc = ok0 ? ox : 0;
    float sc[4], sh[4];
    const bool aff = p.in_scale != nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = aff ? p.in_scale[cc + j] : 1.f; sh[j] = aff ? p.in_shift[cc + j] : 0.f; }
    const float lo = (aff && p.in_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (aff && p.in_act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    const int oy_begin = yb * rows, oy_end = min(oy_begin + rows, p.OH);
    const int ix0 = oxc - p.pad_l;
    const T* ximg = x + (long long)img * p.H * p.W * p.C + cc;
    const T* dyimg = dy + ((long long)img * p.OH * p.OW + oxc) * p.C + cc;
    const int dyoff1 = ok1 ? p.C : 0;
    bool xok[4];
    int xoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ix = ix0 + k;
        xok[k] = ok0 && ix >= 0 && ix < p.W;
        xoff[k] = (xok[k] ? ix : 0) * p.C;
    }
    auto row_load = [&](Raw4<T> (&r)[4], int iy) {
        const int iyc = min(max(iy, 0), p.H - 1);
        const T* rowp = ximg + (long long)iyc * p.W * p.C;
#pragma unroll
        for (int k = 0; k < 4; ++k) raw_load(r[k], rowp + xoff[k]);
    };
    auto dy_load = [&](Raw4<T> (&r)[2], int oy) {
        const T* q = dyimg + (long long)min(oy, p.OH - 1) * p.OW * p.C;
        raw_load(r[0], q);
        raw_load(r[1], q + dyoff1);
    };
    auto row_act = [&](const Raw4<T> (&r)[4], int iy, f32x2_t (&a)[4][2]) {
        if (iy < 0 || iy >= p.H) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k][0] = (f32x2_t){0.f, 0.f}; a[k][1] = (f32x2_t){0.f, 0.f}; }
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[4];
            raw_unpack(r[k], f);
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * sc[j] + sh[j], lo, hi);
            if (k != 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = xok[k] ? f[j] : 0.f;
            }
            a[k][0] = (f32x2_t){f[0], f[1]};
            a[k][1] = (f32x2_t){f[2], f[3]};
        }
    };
    f32x2_t a01[9], a23[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { a01[t] = (f32x2_t){0.f, 0.f}; a23[t] = (f32x2_t){0.f, 0.f}; }
    auto accum = [&](const f32x2_t (&ra_)[4][2], const f32x2_t (&rb_)[4][2], const f32x2_t (&rc_)[4][2], const Raw4<T> (&d)[2]) {
        float g[4], h[4];
        raw_unpack(d[0], g);
        raw_unpack(d[1], h);
        const float m1 = ok1 ? 1.f : 0.f;                             // the second column may be outside the image
        const f32x2_t g01 = {g[0], g[1]}, g23 = {g[2], g[3]};
        const f32x2_t h01 = {h[0] * m1, h[1] * m1}, h23 = {h[2] * m1, h[3] * m1};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a01[k] += ra_[k][0] * g01 + ra_[k + 1][0] * h01;     a23[k] += ra_[k][1] * g23 + ra_[k + 1][1] * h23;
            a01[3 + k] += rb_[k][0] * g01 + rb_[k + 1][0] * h01; a23[3 + k] += rb_[k][1] * g23 + rb_[k + 1][1] * h23;
            a01[6 + k] += rc_[k][0] * g01 + rc_[k + 1][0] * h01; a23[6 + k] += rc_[k][1] * g23 + rc_[k + 1][1] * h23;
        }
    };
    f32x2_t r0[4][2], r1[4][2], r2[4][2];
    Raw4<T> ra[4], rb[4], rc[4], da[2], db[2], dc[2];
    int iy = oy_begin - p.pad_t;
    row_load(ra, iy);
    row_load(rb, iy + 1);
    row_load(rc, iy + 2);
    dy_load(da, oy_begin);
    dy_load(db, oy_begin + 1);
    dy_load(dc, oy_begin + 2);
    row_act(ra, iy, r0);
    row_load(ra, iy + 3);
    row_act(rb, iy + 1, r1);
    row_load(rb, iy + 4);
    iy += 2;
    for (int oy = oy_begin; oy < oy_end; oy += 3, iy += 3) {
        row_act(rc, iy, r2);
        row_load(rc, iy + 3);
        accum(r0, r1, r2, da);
        dy_load(da, oy + 3);
        if (oy + 1 < oy_end) {
            row_act(ra, iy + 1, r0);
            row_load(ra, iy + 4);
            accum(r1, r2, r0, db);
            dy_load(db, oy + 4);
        }
        if (oy + 2 < oy_end) {
            row_act(rb, iy + 2, r1);
            row_load(rb, iy + 5);
            accum(r2, r0, r1, dc);
            dy_load(dc, oy + 5);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float4 v = make_float4(a01[t].x, a01[t].y, a23[t].x, a23[t].y);
        if (!ok0) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&red[(t * kThreads + threadIdx.x) * 4]) = v;
    }
    __syncthreads();
    const int nch = ncg * 4;
    float* dst = p.part + (long long)unit * 9 * p.C + cgb * nch;
    for (int o = threadIdx.x; o < 9 * nch; o += kThreads) {
        const int t = o / nch, cj = o - t * nch;
        if (cgb * nch + cj < p.C) {
            float sum = 0.f;
            for (int cidx = 0; cidx < cols; ++cidx) sum += red[(t * kThreads + cidx * ncg) * 4 + cj];
            dst[t * p.C + cj] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-1 BACKWARD in one walk: the data gradient (dwconv_fwd_sw2_kernel over dY with the flipped kernel), the batch-norm backward
// reduction of the layer it feeds (BNR) and the weight gradient (dwconv_wgrad_sw2_kernel). As launches of their own the two
// gradients read dY twice and the layer's raw input twice (once for the weight gradient's activated window, once for the
// reduction's mask and xhat): 5 tensor passes; here dY, the raw input and dA move once each: 3. Same thread map (4 channels x 2
// columns, walking down a strip), ONE window in registers: three rows x four columns of dY. The data gradient of row r gathers
// from it as before; the weight gradient is organised by INPUT row - the activated input row r (four columns, transient) meets
// dY rows r+1, r, r-1 (ky = 0, 1, 2) at this thread's two columns: acc[ky][kx] += in[r][c+j+kx-1] * dY[r-ky+1][c+j] - so the
// second 3 x 4 window the weight-gradient kernel keeps is not needed; the strip's input rows are exactly its output rows, every
// (input row, ky) pair is counted once over the grid. The reduction's raw input at the thread's own columns is the same row r.
// p.x: raw input of the depthwise conv (= the fed layer's raw conv output) with its batch-norm affine in_scale / in_shift /
// in_act (the mask test of the reduction is the same expression); p.dy: dY; p.y: dA; p.wpart: weight partials [units][9][C];
// BNR: p.part [units][2][C] with p.bnr_mean / p.bnr_invstd.
template <typename T, bool BNR>
__global__ __launch_bounds__(kThreads) void dwconv_bwd_sw2_kernel(const DwParams p, int ncg, int cols, int xblocks, int yblocks, int rows) {
    __shared__ __attribute__((aligned(16))) float red[9 * kThreads * 4];   // [tap][thread][4 channels]; afterwards [thread][8] of the reduction
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
    T* __restrict__ y = reinterpret_cast<T*>(p.y);
    int b = xcd_work_id(p.xcd_remap);
    const int cgb = b % p.cblocks; b /= p.cblocks;
    const int unit = b;                                               // partial-slab row (both slabs)
    const int xb = b % xblocks; b /= xblocks;
    const int yb = b % yblocks;
    const int img = b / yblocks;
    const int cgl = threadIdx.x % ncg, col = threadIdx.x / ncg;       // column PAIR
    const int c = (cgb * ncg + cgl) * 4;
    const int ox = (xb * cols + col) * 2;
    const bool ok0 = c < p.C && ox < p.W && col < cols;
    const bool ok1 = ok0 && ox + 1 < p.W;
    const int cc = ok0 ? c : 0;
    f32x2_t sc01, sc23, sh01, sh23;
    f32x2_t w01[9], w23[9];                                            // FLIPPED: the data gradient correlates dY with w[8 - t]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 q = *reinterpret_cast<const float4*>(p.w + (8 - t) * p.C + cc);
        w01[t] = (f32x2_t){q.x, q.y};
        w23[t] = (f32x2_t){q.z, q.w};
    }
    {
        const float4 s4 = *reinterpret_cast<const float4*>(p.in_scale + cc), h4 = *reinterpret_cast<const float4*>(p.in_shift + cc);
        sc01 = (f32x2_t){s4.x, s4.y}; sc23 = (f32x2_t){s4.z, s4.w};
        sh01 = (f32x2_t){h4.x, h4.y}; sh23 = (f32x2_t){h4.z, h4.w};
    }
    const float lo = (p.in_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (p.in_act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    f32x2_t bis01 = {0.f, 0.f}, bis23 = {0.f, 0.f}, bnm01 = {0.f, 0.f}, bnm23 = {0.f, 0.f};
    if constexpr (BNR) {
        const float4 m4 = *reinterpret_cast<const float4*>(p.bnr_mean + cc), i4 = *reinterpret_cast<const float4*>(p.bnr_invstd + cc);
        bis01 = (f32x2_t){i4.x, i4.y}; bis23 = (f32x2_t){i4.z, i4.w};
        bnm01 = (f32x2_t){-m4.x * i4.x, -m4.y * i4.y}; bnm23 = (f32x2_t){-m4.z * i4.z, -m4.w * i4.w};
    }
    const int oy_begin = yb * rows, oy_end = min(oy_begin + rows, p.H);
    const int ix0 = ox - 1;                                           // leftmost column of the 4-column windows
    const long long img_off = (long long)img * p.H * p.W * p.C + cc;
    const T* ximg = x + img_off;
    const T* dimg = dy + img_off;
    bool xok[4];
    int xoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ix = ix0 + k;
        xok[k] = ok0 && ix >= 0 && ix < p.W;
        xoff[k] = (xok[k] ? ix : 0) * p.C;
    }
    const long long rstep = (long long)p.W * p.C;
    auto row_load = [&](Raw4<T> (&r)[4], const T* base, int iy) {
        const T* rowp = base + (long long)min(max(iy, 0), p.H - 1) * rstep;
#pragma unroll
        for (int k = 0; k < 4; ++k) raw_load(r[k], rowp + xoff[k]);
    };
    // a dY row into the window (zeros outside the image)
    auto dy_row = [&](const Raw4<T> (&r)[4], int iy, f32x2_t (&a)[4][2]) {
        const bool rowok = iy >= 0 && iy < p.H;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[4];
            raw_unpack(r[k], f);
            const bool ok = rowok && xok[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = ok ? f[j] : 0.f;
            a[k][0] = (f32x2_t){f[0], f[1]};
            a[k][1] = (f32x2_t){f[2], f[3]};
        }
    };
    // the activated input row (always inside the image: the strip's own rows; columns outside are the activated tensor's zero padding)
    auto in_row = [&](const Raw4<T> (&r)[4], f32x2_t (&a)[4][2]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[4];
            raw_unpack(r[k], f);
            f32x2_t v01 = (f32x2_t){f[0], f[1]} * sc01 + sh01, v23 = (f32x2_t){f[2], f[3]} * sc23 + sh23;
            v01.x = __builtin_amdgcn_fmed3f(v01.x, lo, hi); v01.y = __builtin_amdgcn_fmed3f(v01.y, lo, hi);
            v23.x = __builtin_amdgcn_fmed3f(v23.x, lo, hi); v23.y = __builtin_amdgcn_fmed3f(v23.y, lo, hi);
            if (k != 1) {   // (column 1 = the thread's first own column: inside the image whenever the lane is)
                v01.x = xok[k] ? v01.x : 0.f; v01.y = xok[k] ? v01.y : 0.f;
                v23.x = xok[k] ? v23.x : 0.f; v23.y = xok[k] ? v23.y : 0.f;
            }
            a[k][0] = v01;
            a[k][1] = v23;
        }
    };
    f32x2_t a01[9], a23[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { a01[t] = (f32x2_t){0.f, 0.f}; a23[t] = (f32x2_t){0.f, 0.f}; }
    f32x2_t s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
    T* yp = y + img_off + ((long long)oy_begin * p.W + (ok0 ? ox : 0)) * p.C;
    auto account = [&](f32x2_t d01, f32x2_t d23, const Raw4<T>& xr) {
        if constexpr (BNR) {
            float f[4];
            raw_unpack(xr, f);
            const f32x2_t x01 = {f[0], f[1]}, x23 = {f[2], f[3]};
            const f32x2_t p01 = x01 * sc01 + sh01, p23 = x23 * sc23 + sh23;
            const f32x2_t r01 = round_storage<T>(d01), r23 = round_storage<T>(d23);
            f32x2_t g01, g23;
            g01.x = (p01.x > lo && p01.x < hi) ? r01.x : 0.f; g01.y = (p01.y > lo && p01.y < hi) ? r01.y : 0.f;
            g23.x = (p23.x > lo && p23.x < hi) ? r23.x : 0.f; g23.y = (p23.y > lo && p23.y < hi) ? r23.y : 0.f;
            s01 += g01; s23 += g23;
            q01 += g01 * (x01 * bis01 + bnm01); q23 += g23 * (x23 * bis23 + bnm23);
        }
    };
    // one row r: window rows (r - 1, r, r + 1) = (d0, d1, d2) of dY, the input row `xin` (activated) with its raw pieces `xr`
    auto step = [&](const f32x2_t (&d0)[4][2], const f32x2_t (&d1)[4][2], const f32x2_t (&d2)[4][2], const f32x2_t (&xin)[4][2],
                    const Raw4<T> (&xr)[4]) {
        // data gradient of the two columns (the forward kernel's emit over dY, flipped weights)
        f32x2_t e01 = {0.f, 0.f}, e23 = {0.f, 0.f}, g01 = {0.f, 0.f}, g23 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            e01 += d0[k][0] * w01[k];          e23 += d0[k][1] * w23[k];
            e01 += d1[k][0] * w01[3 + k];      e23 += d1[k][1] * w23[3 + k];
            e01 += d2[k][0] * w01[6 + k];      e23 += d2[k][1] * w23[6 + k];
            g01 += d0[k + 1][0] * w01[k];      g23 += d0[k + 1][1] * w23[k];
            g01 += d1[k + 1][0] * w01[3 + k];  g23 += d1[k + 1][1] * w23[3 + k];
            g01 += d2[k + 1][0] * w01[6 + k];  g23 += d2[k + 1][1] * w23[6 + k];
        }
        if (ok0) {
            account(e01, e23, xr[1]);
            store4x2(yp, e01, e23);
        }
        if (ok1) {
            account(g01, g23, xr[2]);
            store4x2(yp + p.C, g01, g23);
        }
        yp += rstep;
        // weight gradient: input row r against dY rows r + 1 (ky = 0), r (ky = 1), r - 1 (ky = 2) at the own columns (window 1, 2)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a01[k] += xin[k][0] * d2[1][0] + xin[k + 1][0] * d2[2][0];     a23[k] += xin[k][1] * d2[1][1] + xin[k + 1][1] * d2[2][1];
            a01[3 + k] += xin[k][0] * d1[1][0] + xin[k + 1][0] * d1[2][0]; a23[3 + k] += xin[k][1] * d1[1][1] + xin[k + 1][1] * d1[2][1];
            a01[6 + k] += xin[k][0] * d0[1][0] + xin[k + 1][0] * d0[2][0]; a23[6 + k] += xin[k][1] * d0[1][1] + xin[k + 1][1] * d0[2][1];
        }
    };
    f32x2_t r0[4][2], r1[4][2], r2[4][2], xin[4][2];
    Raw4<T> da[4], db[4], dc[4], xa[4], xb2[4], xc[4];
    // dY rows oy_begin - 1, oy_begin, oy_begin + 1 and input rows oy_begin, +1, +2 are requested up front; afterwards every consumed
    // buffer is re-requested three rows ahead (as in the forward walk)
    row_load(da, dimg, oy_begin - 1);
    row_load(db, dimg, oy_begin);
    row_load(dc, dimg, oy_begin + 1);
    row_load(xa, ximg, oy_begin);
    row_load(xb2, ximg, oy_begin + 1);
    row_load(xc, ximg, oy_begin + 2);
    dy_row(da, oy_begin - 1, r0);
    row_load(da, dimg, oy_begin + 2);
    dy_row(db, oy_begin, r1);
    row_load(db, dimg, oy_begin + 3);
    for (int r = oy_begin; r < oy_end; r += 3) {
        dy_row(dc, r + 1, r2);
        row_load(dc, dimg, r + 4);
        in_row(xa, xin);
        step(r0, r1, r2, xin, xa);
        row_load(xa, ximg, r + 3);
        if (r + 1 < oy_end) {
            dy_row(da, r + 2, r0);
            row_load(da, dimg, r + 5);
            in_row(xb2, xin);
            step(r1, r2, r0, xin, xb2);
            row_load(xb2, ximg, r + 4);
        }
        if (r + 2 < oy_end) {
            dy_row(db, r + 3, r1);
            row_load(db, dimg, r + 6);
            in_row(xc, xin);
            step(r2, r0, r1, xin, xc);
            row_load(xc, ximg, r + 5);
        }
    }
    // weight partials: the block's columns summed in a fixed order
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float4 v = make_float4(a01[t].x, a01[t].y, a23[t].x, a23[t].y);
        if (!ok0) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&red[(t * kThreads + threadIdx.x) * 4]) = v;
    }
    __syncthreads();
    const int nch = ncg * 4;
    {
        float* dst = p.wpart + (long long)unit * 9 * p.C + cgb * nch;
        for (int o = threadIdx.x; o < 9 * nch; o += kThreads) {
            const int t = o / nch, cj = o - t * nch;
            if (cgb * nch + cj < p.C) {
                float sum = 0.f;
                for (int cidx = 0; cidx < cols; ++cidx) sum += red[(t * kThreads + cidx * ncg) * 4 + cj];
                dst[t * p.C + cj] = sum;
            }
        }
    }
    if constexpr (BNR) {
        __syncthreads();   // the weight partials have been read
        float st[8] = {s01.x, s01.y, s23.x, s23.y, q01.x, q01.y, q23.x, q23.y};
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = ok0 ? st[j] : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < ncg && (cgb * ncg + (int)threadIdx.x) * 4 < p.C) {
            float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int cidx = 0; cidx < cols; ++cidx)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[j] += red[(cidx * ncg + threadIdx.x) * 8 + j];
            float* dst = p.part + (long long)unit * 2 * p.C + (cgb * ncg + threadIdx.x) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { dst[j] = acc8[j]; dst[p.C + j] = acc8[4 + j]; }
        }
    }
}

// stride-2 data gradient (gather form): dx[iy,ix,c] = sum_{ky,kx} dy[(iy+pt-ky)/2,(ix+pl-kx)/2,c]*w[ky,kx,c]
// over the taps for which the division is exact. One thread = one input pixel x 16 bytes of channels.
template <typename T>
__global__ __launch_bounds__(kThreads) void dwconv_dgrad_s2_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                                   T* __restrict__ dx, int N, int H, int W, int C, int OH,
                                                                   int OW, int pad_t, int pad_l, long long total_vec) {
    constexpr int VE = Vec16<T>::N;
    const int cvec = C / VE;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total_vec;
         i += (long long)gridDim.x * kThreads) {
        const int vg = (int)(i % cvec);
        long long r = i / cvec;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int n = (int)(r / H);
        float acc[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) acc[j] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + pad_t - ky;
            if (ty < 0 || (ty & 1) || (ty >> 1) >= OH) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + pad_l - kx;
                if (tx < 0 || (tx & 1) || (tx >> 1) >= OW) continue;
                Vec16<T> v;
                v.load(dy + (((long long)n * OH + (ty >> 1)) * OW + (tx >> 1)) * C + vg * VE);
                float f[VE];
                v.unpack(f);
#pragma unroll
                for (int j = 0; j < VE; ++j) acc[j] += f[j] * w[(ky * 3 + kx) * C + vg * VE + j];
            }
        }
        Vec16<T> ov;
        ov.pack(acc);
        ov.store(dx + i * VE);
    }
}

// weight gradient: dw[t][c] = sum a[n, oy*S+ky-pt, ox*S+kx-pl, c] * dy[n,oy,ox,c].
// Blocks walk many tiles of one channel block with the 9 x VE accumulators in registers and reduce once.
template <typename T, int STRIDE>
__global__ __launch_bounds__(kThreads, 2) void dwconv_wgrad_kernel(const DwParams p, int nsplit) {
    using TL = DwTile<STRIDE>;
    constexpr int VE = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* tile = smem_f;
    float* red = smem_f + TL::HH * TL::HW * p.nvg * VE;  // [4][16][36]

    const int cb = blockIdx.x % p.cblocks;
    const int split = blockIdx.x / p.cblocks;
    const int c0 = cb * p.nvg * VE;
    const int cb_vecs = min(p.nvg, (p.C - c0) / VE);
    const int cstride = p.nvg * VE;
    const int ncg = cstride / 4;
    const int cg = threadIdx.x % ncg;
    const int pt = threadIdx.x / ncg;
    const int npt = kThreads / ncg;
    const bool cg_ok = cg * 4 < cb_vecs * VE;
    const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);

    f32x2_t acc2[18];   // [tap][channel pair]: packed FP32 FMAs
#pragma unroll
    for (int j = 0; j < 18; ++j) acc2[j] = (f32x2_t){0.f, 0.f};

    const HaloAffine<T> aff = load_halo_affine<T>(p, c0, cb_vecs);
    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    HaloRegs<T, STRIDE> hr;
    int img, oy0, ox0;
    // same loop shape as the forward kernel; the tile's dY values (one 4-channel piece per owned output pixel) are
    // ALL requested before the first barrier instead of one exposed global load per pixel inside the loop
    constexpr int NOP = TL::TH * TL::TW / (kThreads / 16);   // owned output pixels per thread at most (ncg <= 16)
    if (split < ntiles) {
        dw_tile_origin(p, split, TL::TH, TL::TW, img, oy0, ox0);
        hr.load(p, img, oy0, ox0, c0, cb_vecs);
        hr.commit(p, aff, tile);
    }
    for (int t = split; t < ntiles; t += nsplit) {
        dw_tile_origin(p, t, TL::TH, TL::TW, img, oy0, ox0);
        float g_all[NOP][4];
        unsigned gmask = 0u;
#pragma unroll
        for (int k = 0; k < NOP; ++k) {
            const int op = pt + k * npt;
            const int oyl = op / TL::TW, oxl = op - oyl * TL::TW;
            const int oy = oy0 + oyl, ox = ox0 + oxl;
            const bool ok = op < TL::TH * TL::TW && cg_ok && oy < p.OH && ox < p.OW;
            load4(dy + (ok ? (((long long)img * p.OH + oy) * p.OW + ox) * p.C + c0 + cg * 4 : 0), g_all[k]);
            gmask |= (ok ? 1u : 0u) << k;
        }
        const bool more = t + nsplit < ntiles;
        if (more) {   // prefetch the next tile: its loads fly while this tile is computed
            int img2, oy2, ox2;
            dw_tile_origin(p, t + nsplit, TL::TH, TL::TW, img2, oy2, ox2);
            hr.load(p, img2, oy2, ox2, c0, cb_vecs);
        }
        __syncthreads();  // this tile's halo image is complete
#pragma unroll
        for (int k = 0; k < NOP; ++k) {
            const int op = pt + k * npt;
            const int oyl = op / TL::TW, oxl = op - oyl * TL::TW;
            if (!((gmask >> k) & 1u)) continue;
            const float (&g)[4] = g_all[k];
            // all 9 LDS reads before the first FMA (fenced): see the forward kernel
            float4 q[9];
            const float* base = tile + ((oyl * STRIDE) * TL::HW + oxl * STRIDE) * cstride + cg * 4;
#pragma unroll
            for (int t = 0; t < 9; ++t) q[t] = *reinterpret_cast<const float4*>(base + ((t / 3) * TL::HW + (t % 3)) * cstride);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                acc2[t * 2 + 0] += (f32x2_t){q[t].x, q[t].y} * (f32x2_t){g[0], g[1]};
                acc2[t * 2 + 1] += (f32x2_t){q[t].z, q[t].w} * (f32x2_t){g[2], g[3]};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // everybody is done reading this tile
        if (more) hr.commit(p, aff, tile);
    }
    float acc[36];
#pragma unroll
    for (int j = 0; j < 18; ++j) { acc[2 * j] = acc2[j].x; acc[2 * j + 1] = acc2[j].y; }
    reduce_same_vg<36>(acc, ncg, red);
    if ((int)threadIdx.x < ncg && cg_ok) {
        float* dst = p.part + (long long)split * 9 * p.C + c0 + cg * 4;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[t * p.C + j] = acc[t * 4 + j];
    }
}

void tf_same_pad(int size, int stride, int* out, int* pad_before) {
    *out = (size + stride - 1) / stride;
    int total = (*out - 1) * stride + 3 - size;
    if (total < 0) total = 0;
    *pad_before = total / 2;
}

int fill_params(DwParams& p, int N, int H, int W, int C, int stride, int dtype) {
    MPN_REQUIRE(stride == 1 || stride == 2, MPN_ERR_BAD_SHAPE, "dwconv: stride must be 1 or 2");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "dwconv: dtype %d", dtype);
    const int ve = dtype == MPN_F32 ? 4 : 8;
    MPN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % ve == 0, MPN_ERR_BAD_SHAPE,
                "dwconv: C (%d) must be a multiple of %d", C, ve);
    p.N = N; p.H = H; p.W = W; p.C = C;
    p.xcd_remap = 1;
    tf_same_pad(H, stride, &p.OH, &p.pad_t);
    tf_same_pad(W, stride, &p.OW, &p.pad_l);
    const int th = stride == 1 ? 8 : 4, tw = stride == 1 ? 16 : 8;
    p.tiles_y = (p.OH + th - 1) / th;
    p.tiles_x = (p.OW + tw - 1) / tw;
    int nvg = 8;
    while (nvg > C / ve) nvg >>= 1;
    p.nvg = nvg;
    p.cblocks = (C / ve + nvg - 1) / nvg;
    return MPN_OK;
}

template <int STRIDE> size_t dw_smem(int ve, int nred, int nvg) {
    using TL = DwTile<STRIDE>;
    return (size_t)(TL::HH * TL::HW * nvg * ve + 4 * 16 * nred) * sizeof(float);   // halo tile + reduction scratch
}

template <typename K> int set_smem(K kernel, size_t bytes) {
    static mpn_attr_mask_t attr_mask{0};   // (one instance per kernel type K; the sizes used are <= the first one's)
    if (bytes > 48 * 1024) MPN_HIP(mpn_ensure_dynamic_lds((const void*)kernel, (int)bytes, &attr_mask));
    return MPN_OK;
}

}  // namespace

extern "C" int mpn_dwconv_out_size(int size, int stride) { return (size + stride - 1) / stride; }

namespace {
// stride-2 data gradient, sliding-window form for even H, W (pad_t = pad_l = 0): a thread owns 4 channels of one dY column
// b and walks down the dY rows a; with the previous row's two pieces in registers, dY[a-1..a][b-1..b] gives the 2x2 block
// dx[2a..2a+1][2b..2b+1] (4 + 2 + 2 + 1 taps): 2 loads and 4 stores per step, no parity branches (the gather kernel above
// issues all 9 tap loads with a quarter of the lanes active each).
// dY rows per thread: 32 on maps of 64 dY rows and more (256x256 input: 81 -> 72 us), 16 at 32, 8 below (16-row maps
// would otherwise be one strip per column block)
__host__ __device__ inline int dg_rows(int OH) { return OH >= 64 ? 32 : (OH >= 32 ? 16 : 8); }
// ADD: dx = gradient + addend (the FPN lateral's gradient into a backbone feature map: one read instead of the separate
// read-modify-write pass, a single rounding of the sum, and the sum is what the fused batch-norm reduction sees)
template <typename T, bool BNR, bool ADD>
__global__ __launch_bounds__(kThreads) void dwconv_dgrad_s2_sw_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                                      T* __restrict__ dx, int H, int W, int C, int OH, int OW,
                                                                      int ncg, int cols, int xblocks, int yblocks, int cblocks,
                                                                      const DwParams p) {
    __shared__ float red[BNR ? kThreads * 8 : 1];
    int bi = xcd_work_id(p.xcd_remap);
    const int xb = bi % xblocks; bi /= xblocks;
    const int yb = bi % yblocks; bi /= yblocks;
    const int cgb = bi % cblocks;
    const int img = bi / cblocks;
    const int cgl = threadIdx.x % ncg, col = threadIdx.x / ncg;
    const int c = (cgb * ncg + cgl) * 4;
    const int b = xb * cols + col;
    const bool lane_ok = c < C && b < OW && col < cols;
    const int cc = lane_ok ? c : 0, bc = lane_ok ? b : 0;
    f32x2_t w01[9], w23[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 q = *reinterpret_cast<const float4*>(w + t * C + cc);
        w01[t] = (f32x2_t){q.x, q.y};
        w23[t] = (f32x2_t){q.z, q.w};
    }
    const int dgr = dg_rows(OH);
    const int a_begin = yb * dgr, a_end = min(a_begin + dgr, OH);
    const bool left_ok = bc > 0;
    const T* dyimg = dy + ((long long)img * OH * OW) * C + cc;
    const int off_c = bc * C, off_l = (left_ok ? bc - 1 : 0) * C;
    auto row_load = [&](Raw4<T> (&r)[2], int a) {
        const int ac = min(max(a, 0), OH - 1);
        const T* rowp = dyimg + (long long)ac * OW * C;
        raw_load(r[0], rowp + off_l);
        raw_load(r[1], rowp + off_c);
    };
    auto row_cvt = [&](const Raw4<T> (&r)[2], int a, f32x2_t (&v)[2][2]) {   // [left, centre][channel pair]; zero outside
        if (a < 0) {
            v[0][0] = v[0][1] = v[1][0] = v[1][1] = (f32x2_t){0.f, 0.f};
            return;
        }
        float f[4];
        raw_unpack(r[0], f);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = left_ok ? f[j] : 0.f;
        v[0][0] = (f32x2_t){f[0], f[1]}; v[0][1] = (f32x2_t){f[2], f[3]};
        raw_unpack(r[1], f);
        v[1][0] = (f32x2_t){f[0], f[1]}; v[1][1] = (f32x2_t){f[2], f[3]};
    };
    T* xp = dx + (((long long)img * H + 2 * a_begin) * W + 2 * bc) * C + cc;
    const long long xrow = (long long)W * C;
    // BNR: batch-norm backward reduction of the layer whose input gradient dx is (see dwconv_fwd_sw_kernel)
    f32x2_t s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
    f32x2_t bsc01 = {0.f, 0.f}, bsc23 = {0.f, 0.f}, bsh01 = {0.f, 0.f}, bsh23 = {0.f, 0.f};
    f32x2_t bis01 = {0.f, 0.f}, bis23 = {0.f, 0.f}, bnm01 = {0.f, 0.f}, bnm23 = {0.f, 0.f};
    float blo = -INFINITY, bhi = INFINITY;
    const T* bxp = nullptr;
    if constexpr (BNR) {
        const float4 s4 = *reinterpret_cast<const float4*>(p.bnr_scale + cc), h4 = *reinterpret_cast<const float4*>(p.bnr_shift + cc);
        const float4 m4 = *reinterpret_cast<const float4*>(p.bnr_mean + cc), i4 = *reinterpret_cast<const float4*>(p.bnr_invstd + cc);
        bsc01 = (f32x2_t){s4.x, s4.y}; bsc23 = (f32x2_t){s4.z, s4.w};
        bsh01 = (f32x2_t){h4.x, h4.y}; bsh23 = (f32x2_t){h4.z, h4.w};
        bis01 = (f32x2_t){i4.x, i4.y}; bis23 = (f32x2_t){i4.z, i4.w};
        bnm01 = (f32x2_t){-m4.x * i4.x, -m4.y * i4.y}; bnm23 = (f32x2_t){-m4.z * i4.z, -m4.w * i4.w};
        blo = (p.bnr_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
        bhi = (p.bnr_act == MPN_ACT_RELU6) ? 6.f : INFINITY;
        bxp = reinterpret_cast<const T*>(p.bnr_x) + (((long long)img * H + 2 * a_begin) * W + 2 * bc) * C + cc;
    }
    Raw4<T> yq[4];   // the raw outputs of that layer at the 2x2 block of the NEXT emit (requested one step ahead)
    auto bnr_load = [&](int a) {
        if constexpr (BNR) {
            const T* q = bxp + (long long)(2 * (min(a, OH - 1) - a_begin)) * xrow;
            raw_load(yq[0], q); raw_load(yq[1], q + C); raw_load(yq[2], q + xrow); raw_load(yq[3], q + xrow + C);
        }
    };
    auto bnr_acc = [&](const Raw4<T>& yr, f32x2_t o01, f32x2_t o23) {
        float f[4];
        raw_unpack(yr, f);
        const f32x2_t x01 = {f[0], f[1]}, x23 = {f[2], f[3]};
        const f32x2_t p01 = x01 * bsc01 + bsh01, p23 = x23 * bsc23 + bsh23;
        const f32x2_t d01 = round_storage<T>(o01), d23 = round_storage<T>(o23);
        f32x2_t g01, g23;
        g01.x = (p01.x > blo && p01.x < bhi) ? d01.x : 0.f; g01.y = (p01.y > blo && p01.y < bhi) ? d01.y : 0.f;
        g23.x = (p23.x > blo && p23.x < bhi) ? d23.x : 0.f; g23.y = (p23.y > blo && p23.y < bhi) ? d23.y : 0.f;
        s01 += g01; s23 += g23;
        q01 += g01 * (x01 * bis01 + bnm01); q23 += g23 * (x23 * bis23 + bnm23);
    };
    const T* adp = nullptr;
    if constexpr (ADD) adp = reinterpret_cast<const T*>(p.addend) + (((long long)img * H + 2 * a_begin) * W + 2 * bc) * C + cc;
    Raw4<T> aq[4];   // the addend at the 2x2 block of the NEXT emit (requested one step ahead, like yq)
    auto add_load = [&](int a) {
        if constexpr (ADD) {
            const T* q = adp + (long long)(2 * (min(a, OH - 1) - a_begin)) * xrow;
            raw_load(aq[0], q); raw_load(aq[1], q + C); raw_load(aq[2], q + xrow); raw_load(aq[3], q + xrow + C);
        }
    };
    auto add_acc = [&](const Raw4<T>& r, f32x2_t& o01, f32x2_t& o23) {
        float f[4];
        raw_unpack(r, f);
        o01 += (f32x2_t){f[0], f[1]};
        o23 += (f32x2_t){f[2], f[3]};
    };
    int a_cur = a_begin;
    auto emit = [&](const f32x2_t (&pv)[2][2], const f32x2_t (&cv)[2][2]) {   // previous row a-1, current row a
        // taps t = ky*3 + kx
        f32x2_t e01 = cv[1][0] * w01[0] + pv[1][0] * w01[6] + cv[0][0] * w01[2] + pv[0][0] * w01[8];
        f32x2_t e23 = cv[1][1] * w23[0] + pv[1][1] * w23[6] + cv[0][1] * w23[2] + pv[0][1] * w23[8];
        f32x2_t f01 = cv[1][0] * w01[1] + pv[1][0] * w01[7];
        f32x2_t f23 = cv[1][1] * w23[1] + pv[1][1] * w23[7];
        f32x2_t g01 = cv[1][0] * w01[3] + cv[0][0] * w01[5];
        f32x2_t g23 = cv[1][1] * w23[3] + cv[0][1] * w23[5];
        f32x2_t h01 = cv[1][0] * w01[4];
        f32x2_t h23 = cv[1][1] * w23[4];
        if constexpr (ADD) {
            add_acc(aq[0], e01, e23);
            add_acc(aq[1], f01, f23);
            add_acc(aq[2], g01, g23);
            add_acc(aq[3], h01, h23);
        }
        if (lane_ok) {
            store4x2(xp, e01, e23);
            store4x2(xp + C, f01, f23);
            store4x2(xp + xrow, g01, g23);
            store4x2(xp + xrow + C, h01, h23);
            if constexpr (BNR) {
                bnr_acc(yq[0], e01, e23);
                bnr_acc(yq[1], f01, f23);
                bnr_acc(yq[2], g01, g23);
                bnr_acc(yq[3], h01, h23);
            }
        }
        xp += 2 * xrow;
        ++a_cur;
        bnr_load(a_cur);
        add_load(a_cur);
    };
    bnr_load(a_begin);
    add_load(a_begin);
    Raw4<T> ra[2], rb[2], rc[2];
    f32x2_t v0[2][2], v1[2][2];
    row_load(ra, a_begin - 1);
    row_load(rb, a_begin);
    row_load(rc, a_begin + 1);
    row_cvt(ra, a_begin - 1, v0);
    row_load(ra, a_begin + 2);
    // the three raw buffers rotate; the converted rows alternate between v0 and v1
    for (int a = a_begin; a < a_end; a += 6) {
        row_cvt(rb, a, v1);     row_load(rb, a + 3); emit(v0, v1);
        if (a + 1 < a_end) { row_cvt(rc, a + 1, v0); row_load(rc, a + 4); emit(v1, v0); }
        if (a + 2 < a_end) { row_cvt(ra, a + 2, v1); row_load(ra, a + 5); emit(v0, v1); }
        if (a + 3 < a_end) { row_cvt(rb, a + 3, v0); row_load(rb, a + 6); emit(v1, v0); }
        if (a + 4 < a_end) { row_cvt(rc, a + 4, v1); row_load(rc, a + 7); emit(v0, v1); }
        if (a + 5 < a_end) { row_cvt(ra, a + 5, v0); row_load(ra, a + 8); emit(v1, v0); }
    }
    if constexpr (BNR) {
        float st[8] = {s01.x, s01.y, s23.x, s23.y, q01.x, q01.y, q23.x, q23.y};
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = lane_ok ? st[j] : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < ncg && (cgb * ncg + (int)threadIdx.x) * 4 < C) {
            float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int cidx = 0; cidx < cols; ++cidx)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[j] += red[(cidx * ncg + threadIdx.x) * 8 + j];
            const int prow = (img * yblocks + yb) * xblocks + xb;
            float* dst = p.part + (long long)prow * 2 * C + (cgb * ncg + threadIdx.x) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { dst[j] = acc8[j]; dst[C + j] = acc8[4 + j]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-2 BACKWARD in one walk (even H, W: pad 0), the counterpart of dwconv_bwd_sw2_kernel: dwconv_dgrad_s2_sw_kernel's walk (a
// thread = 4 channels of one dY column b, rows a down a strip; dY[a-1..a][b-1..b] gives the 2 x 2 block dA[2a..2a+1][2b..2b+1], the
// addend and the reduction of the batch-norm below ride on it) that ALSO keeps the activated input rows 2a, 2a+1, 2a+2 at columns
// 2b..2b+2 - the weight gradient's window for dY[a][b]: acc[ky][kx] += in[2a+ky][2b+kx] * dY[a][b]. Row 2a+2 is the next step's row
// 2a, so a step loads two new input rows (six pieces). The reduction's raw x at the 2 x 2 block are four of those pieces: the
// separate launches read the input once for the weight gradient and once more for the reduction, and dY twice.
// p.x: raw input with its batch-norm affine in_scale / in_shift / in_act; p.dy: dY [N,OH,OW,C]; p.y: dA [N,H,W,C]; p.wpart: weight
// partials [units][9][C]; BNR: p.part [units][2][C] with p.bnr_mean / p.bnr_invstd; ADD: p.addend (dA's shape).
template <typename T, bool BNR, bool ADD>
__global__ __launch_bounds__(kThreads) void dwconv_bwd_s2_kernel(const DwParams p, int ncg, int cols, int xblocks, int yblocks, int rows) {
    __shared__ __attribute__((aligned(16))) float red[9 * kThreads * 4];   // [tap][thread][4 channels]; afterwards [thread][8] of the reduction
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
    T* __restrict__ dxo = reinterpret_cast<T*>(p.y);
    const int H = p.H, W = p.W, C = p.C, OH = p.OH, OW = p.OW;
    int bi = xcd_work_id(p.xcd_remap);
    const int cgb = bi % p.cblocks; bi /= p.cblocks;
    const int unit = bi;                                              // partial-slab row (both slabs)
    const int xb = bi % xblocks; bi /= xblocks;
    const int yb = bi % yblocks;
    const int img = bi / yblocks;
    const int cgl = threadIdx.x % ncg, col = threadIdx.x / ncg;
    const int c = (cgb * ncg + cgl) * 4;
    const int b = xb * cols + col;
    const bool lane_ok = c < C && b < OW && col < cols;
    const int cc = lane_ok ? c : 0, bc = lane_ok ? b : 0;
    f32x2_t w01[9], w23[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 q = *reinterpret_cast<const float4*>(p.w + t * C + cc);
        w01[t] = (f32x2_t){q.x, q.y};
        w23[t] = (f32x2_t){q.z, q.w};
    }
    f32x2_t sc01, sc23, sh01, sh23;
    {
        const float4 s4 = *reinterpret_cast<const float4*>(p.in_scale + cc), h4 = *reinterpret_cast<const float4*>(p.in_shift + cc);
        sc01 = (f32x2_t){s4.x, s4.y}; sc23 = (f32x2_t){s4.z, s4.w};
        sh01 = (f32x2_t){h4.x, h4.y}; sh23 = (f32x2_t){h4.z, h4.w};
    }
    const float lo = (p.in_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (p.in_act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    f32x2_t bis01 = {0.f, 0.f}, bis23 = {0.f, 0.f}, bnm01 = {0.f, 0.f}, bnm23 = {0.f, 0.f};
    if constexpr (BNR) {
        const float4 m4 = *reinterpret_cast<const float4*>(p.bnr_mean + cc), i4 = *reinterpret_cast<const float4*>(p.bnr_invstd + cc);
        bis01 = (f32x2_t){i4.x, i4.y}; bis23 = (f32x2_t){i4.z, i4.w};
        bnm01 = (f32x2_t){-m4.x * i4.x, -m4.y * i4.y}; bnm23 = (f32x2_t){-m4.z * i4.z, -m4.w * i4.w};
    }
    const int a_begin = yb * rows, a_end = min(a_begin + rows, OH);
    const bool left_ok = bc > 0, right_ok = lane_ok && 2 * bc + 2 < W;
    const T* dyimg = dy + ((long long)img * OH * OW) * C + cc;
    const int off_c = bc * C, off_l = (left_ok ? bc - 1 : 0) * C;
    const T* ximg = x + (long long)img * H * W * C + cc;
    const int xo0 = 2 * bc * C, xo1 = xo0 + C, xo2 = right_ok ? xo0 + 2 * C : xo0;
    const long long xrow = (long long)W * C;
    auto dy_load = [&](Raw4<T> (&r)[2], int a) {
        const T* rowp = dyimg + (long long)min(max(a, 0), OH - 1) * OW * C;
        raw_load(r[0], rowp + off_l);
        raw_load(r[1], rowp + off_c);
    };
    auto dy_cvt = [&](const Raw4<T> (&r)[2], int a, f32x2_t (&v)[2][2]) {   // [left, centre][channel pair]; zero outside
        if (a < 0) {
            v[0][0] = v[0][1] = v[1][0] = v[1][1] = (f32x2_t){0.f, 0.f};
            return;
        }
        float f[4];
        raw_unpack(r[0], f);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = left_ok ? f[j] : 0.f;
        v[0][0] = (f32x2_t){f[0], f[1]}; v[0][1] = (f32x2_t){f[2], f[3]};
        raw_unpack(r[1], f);
        v[1][0] = (f32x2_t){f[0], f[1]}; v[1][1] = (f32x2_t){f[2], f[3]};
    };
    auto x_load = [&](Raw4<T> (&r)[3], int iy) {
        const T* rowp = ximg + (long long)min(iy, H - 1) * xrow;
        raw_load(r[0], rowp + xo0);
        raw_load(r[1], rowp + xo1);
        raw_load(r[2], rowp + xo2);
    };
    // an input row, activated (zeros below the image / right of it: the activated tensor's padding)
    auto x_act = [&](const Raw4<T> (&r)[3], int iy, f32x2_t (&a)[3][2]) {
        const bool rowok = iy < H;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float f[4];
            raw_unpack(r[k], f);
            f32x2_t v01 = (f32x2_t){f[0], f[1]} * sc01 + sh01, v23 = (f32x2_t){f[2], f[3]} * sc23 + sh23;
            v01.x = __builtin_amdgcn_fmed3f(v01.x, lo, hi); v01.y = __builtin_amdgcn_fmed3f(v01.y, lo, hi);
            v23.x = __builtin_amdgcn_fmed3f(v23.x, lo, hi); v23.y = __builtin_amdgcn_fmed3f(v23.y, lo, hi);
            const bool ok = rowok && (k < 2 || right_ok);
            v01.x = ok ? v01.x : 0.f; v01.y = ok ? v01.y : 0.f;
            v23.x = ok ? v23.x : 0.f; v23.y = ok ? v23.y : 0.f;
            a[k][0] = v01;
            a[k][1] = v23;
        }
    };
    T* xp = dxo + (((long long)img * H + 2 * a_begin) * W + 2 * bc) * C + cc;
    f32x2_t s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
    auto bnr_acc = [&](const Raw4<T>& yr, f32x2_t o01, f32x2_t o23) {
        float f[4];
        raw_unpack(yr, f);
        const f32x2_t x01 = {f[0], f[1]}, x23 = {f[2], f[3]};
        const f32x2_t p01 = x01 * sc01 + sh01, p23 = x23 * sc23 + sh23;
        const f32x2_t d01 = round_storage<T>(o01), d23 = round_storage<T>(o23);
        f32x2_t g01, g23;
        g01.x = (p01.x > lo && p01.x < hi) ? d01.x : 0.f; g01.y = (p01.y > lo && p01.y < hi) ? d01.y : 0.f;
        g23.x = (p23.x > lo && p23.x < hi) ? d23.x : 0.f; g23.y = (p23.y > lo && p23.y < hi) ? d23.y : 0.f;
        s01 += g01; s23 += g23;
        q01 += g01 * (x01 * bis01 + bnm01); q23 += g23 * (x23 * bis23 + bnm23);
    };
    const T* adp = nullptr;
    if constexpr (ADD) adp = reinterpret_cast<const T*>(p.addend) + (((long long)img * H + 2 * a_begin) * W + 2 * bc) * C + cc;
    Raw4<T> aq[ADD ? 4 : 1];   // the addend at the 2x2 block of the NEXT emit (requested one step ahead)
    auto add_load = [&](int a) {
        if constexpr (ADD) {
            const T* q = adp + (long long)(2 * (min(a, OH - 1) - a_begin)) * xrow;
            raw_load(aq[0], q); raw_load(aq[1], q + C); raw_load(aq[2], q + xrow); raw_load(aq[3], q + xrow + C);
        }
    };
    auto add_acc = [&](const Raw4<T>& r, f32x2_t& o01, f32x2_t& o23) {
        float f[4];
        raw_unpack(r, f);
        o01 += (f32x2_t){f[0], f[1]};
        o23 += (f32x2_t){f[2], f[3]};
    };
    f32x2_t a01[9], a23[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { a01[t] = (f32x2_t){0.f, 0.f}; a23[t] = (f32x2_t){0.f, 0.f}; }
    int a_cur = a_begin;
    // the 2 x 2 block of dA from dY rows a - 1 (pv) and a (cv); e0 / e1: the raw input pieces of rows 2a / 2a + 1 at columns 2b, 2b + 1
    auto emit = [&](const f32x2_t (&pv)[2][2], const f32x2_t (&cv)[2][2], const Raw4<T> (&e0)[2], const Raw4<T> (&e1)[2]) {
        f32x2_t e01 = cv[1][0] * w01[0] + pv[1][0] * w01[6] + cv[0][0] * w01[2] + pv[0][0] * w01[8];
        f32x2_t e23 = cv[1][1] * w23[0] + pv[1][1] * w23[6] + cv[0][1] * w23[2] + pv[0][1] * w23[8];
        f32x2_t f01 = cv[1][0] * w01[1] + pv[1][0] * w01[7];
        f32x2_t f23 = cv[1][1] * w23[1] + pv[1][1] * w23[7];
        f32x2_t g01 = cv[1][0] * w01[3] + cv[0][0] * w01[5];
        f32x2_t g23 = cv[1][1] * w23[3] + cv[0][1] * w23[5];
        f32x2_t h01 = cv[1][0] * w01[4];
        f32x2_t h23 = cv[1][1] * w23[4];
        if constexpr (ADD) {
            add_acc(aq[0], e01, e23);
            add_acc(aq[1], f01, f23);
            add_acc(aq[2], g01, g23);
            add_acc(aq[3], h01, h23);
        }
        if (lane_ok) {
            store4x2(xp, e01, e23);
            store4x2(xp + C, f01, f23);
            store4x2(xp + xrow, g01, g23);
            store4x2(xp + xrow + C, h01, h23);
            if constexpr (BNR) {
                bnr_acc(e0[0], e01, e23);
                bnr_acc(e0[1], f01, f23);
                bnr_acc(e1[0], g01, g23);
                bnr_acc(e1[1], h01, h23);
            }
        }
        xp += 2 * xrow;
        ++a_cur;
        add_load(a_cur);
    };
    // weight gradient of dY[a][b] against the activated input rows 2a (r0), 2a + 1 (r1), 2a + 2 (r2) at columns 2b .. 2b + 2
    auto wacc = [&](const f32x2_t (&r0)[3][2], const f32x2_t (&r1)[3][2], const f32x2_t (&r2)[3][2], const f32x2_t (&cv)[2][2]) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a01[k] += r0[k][0] * cv[1][0];     a23[k] += r0[k][1] * cv[1][1];
            a01[3 + k] += r1[k][0] * cv[1][0]; a23[3 + k] += r1[k][1] * cv[1][1];
            a01[6 + k] += r2[k][0] * cv[1][0]; a23[6 + k] += r2[k][1] * cv[1][1];
        }
    };
    add_load(a_begin);
    Raw4<T> ra[2], rb[2], rc[2];       // dY rows in flight (three rotate)
    Raw4<T> xa[3], xb2[3];             // the two new input rows of the next step
    Raw4<T> ke[2], ko[2], k1[2];       // raw pieces (columns 2b, 2b + 1) of the even row carried over / the odd row
    f32x2_t v0[2][2], v1[2][2];
    f32x2_t E0[3][2], E1[3][2], E2[3][2];
    dy_load(ra, a_begin - 1);
    dy_load(rb, a_begin);
    dy_load(rc, a_begin + 1);
    x_load(xa, 2 * a_begin);
    dy_cvt(ra, a_begin - 1, v0);
    dy_load(ra, a_begin + 2);
    x_act(xa, 2 * a_begin, E0);
    ke[0] = xa[0]; ke[1] = xa[1];
    x_load(xa, 2 * a_begin + 1);
    x_load(xb2, 2 * a_begin + 2);
    // one step: dY row a from `rd` (then re-requested three rows ahead) into `cv`; input rows 2a + 1, 2a + 2 from xa / xb2 (then
    // re-requested for the next step); even rows alternate between (Ein, kin) and (Eout, kout)
    auto step = [&](int a, Raw4<T> (&rd)[2], const f32x2_t (&pv)[2][2], f32x2_t (&cv)[2][2], const f32x2_t (&Ein)[3][2], const Raw4<T> (&kin)[2],
                    f32x2_t (&Eout)[3][2], Raw4<T> (&kout)[2]) __attribute__((always_inline)) {
        dy_cvt(rd, a, cv);
        dy_load(rd, a + 3);
        x_act(xa, 2 * a + 1, E1);
        k1[0] = xa[0]; k1[1] = xa[1];
        x_act(xb2, 2 * a + 2, Eout);
        kout[0] = xb2[0]; kout[1] = xb2[1];
        x_load(xa, 2 * a + 3);
        x_load(xb2, 2 * a + 4);
        emit(pv, cv, kin, k1);
        wacc(Ein, E1, Eout, cv);
    };
    for (int a = a_begin; a < a_end; a += 6) {
        step(a, rb, v0, v1, E0, ke, E2, ko);
        if (a + 1 < a_end) step(a + 1, rc, v1, v0, E2, ko, E0, ke);
        if (a + 2 < a_end) step(a + 2, ra, v0, v1, E0, ke, E2, ko);
        if (a + 3 < a_end) step(a + 3, rb, v1, v0, E2, ko, E0, ke);
        if (a + 4 < a_end) step(a + 4, rc, v0, v1, E0, ke, E2, ko);
        if (a + 5 < a_end) step(a + 5, ra, v1, v0, E2, ko, E0, ke);
    }
    // weight partials: the block's columns summed in a fixed order
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float4 v = make_float4(a01[t].x, a01[t].y, a23[t].x, a23[t].y);
        if (!lane_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&red[(t * kThreads + threadIdx.x) * 4]) = v;
    }
    __syncthreads();
    const int nch = ncg * 4;
    {
        float* dst = p.wpart + (long long)unit * 9 * C + cgb * nch;
        for (int o = threadIdx.x; o < 9 * nch; o += kThreads) {
            const int t = o / nch, cj = o - t * nch;
            if (cgb * nch + cj < C) {
                float sum = 0.f;
                for (int cidx = 0; cidx < cols; ++cidx) sum += red[(t * kThreads + cidx * ncg) * 4 + cj];
                dst[t * C + cj] = sum;
            }
        }
    }
    if constexpr (BNR) {
        __syncthreads();   // the weight partials have been read
        float st[8] = {s01.x, s01.y, s23.x, s23.y, q01.x, q01.y, q23.x, q23.y};
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = lane_ok ? st[j] : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < ncg && (cgb * ncg + (int)threadIdx.x) * 4 < C) {
            float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int cidx = 0; cidx < cols; ++cidx)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[j] += red[(cidx * ncg + threadIdx.x) * 8 + j];
            float* dst = p.part + (long long)unit * 2 * C + (cgb * ncg + threadIdx.x) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { dst[j] = acc8[j]; dst[C + j] = acc8[4 + j]; }
        }
    }
}
}  // namespace

// stride-1 launches use the two-column kernel (forward, plain data gradient and data gradient with the fused batch-norm
// reduction alike), stride 2 the one-column kernel
static int dw_xt(const DwParams& p) { return (p.H == p.OH && p.W == p.OW) ? 2 : 1; }
struct DwSwGeom { int ncg, cols, xblocks, yblocks, cblocks; };
static int dw_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus;
}
// Strip height of the stride-1 forward walk WITH statistics (the training forward). The kernel holds three 256-thread blocks per
// CU; with the fixed 32-row strips the two largest layers made 1 024 blocks for 768 slots - a second round a third full, i.e. a
// third of the launch at a third of the chip. Pick the height whose block count costs the fewest rounds x (rows + 2 primed rows):
// 43-row strips = 768 blocks on those layers (Conv2d_1_depthwise 85 -> 64 us in the step).
static int dw_fwd_rows(const DwParams& p, int blocks_per_row_strip) {
    const int def = sw_rows(p.OH, 1);
    if (!(p.H == p.OH && p.W == p.OW) || p.part == nullptr || p.bnr_x != nullptr) return def;
    const long long slots = 3ll * dw_cu_count();
    int best = def;
    long long best_cost = -1;
    for (int yb = (p.OH + 63) / 64; yb <= (p.OH + 15) / 16; ++yb) {
        const int rows = (p.OH + yb - 1) / yb;
        const long long blocks = (long long)blocks_per_row_strip * ((p.OH + rows - 1) / rows);
        const long long cost = ((blocks + slots - 1) / slots) * (rows + 2);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = rows; }
    }
    return best;
}
static DwSwGeom dw_sw_geom(const DwParams& p) {
    DwSwGeom g;
    const int cg_total = p.C / 4;
    const int cap = 32;   // blocks of <= 128 channels: more columns per block, 4x fewer statistics rows on the 512 / 1024-channel maps (21.6 -> 18.8, 15.4 -> 13.0 us)
    g.ncg = cg_total < cap ? cg_total : cap;
    g.cols = kThreads / g.ncg;
    g.cblocks = (cg_total + g.ncg - 1) / g.ncg;
    g.xblocks = (p.OW + g.cols * dw_xt(p) - 1) / (g.cols * dw_xt(p));
    const int swr = p.swr > 0 ? p.swr : dw_fwd_rows(p, p.N * g.cblocks * g.xblocks);
    g.yblocks = (p.OH + swr - 1) / swr;
    return g;
}

static int dw_fwd_nsplit(const DwParams& p) { const DwSwGeom g = dw_sw_geom(p); return p.N * g.yblocks * g.xblocks; }

/* rows of the stats partial slab written by mpn_dwconv_fwd */
extern "C" int mpn_dwconv_num_parts(int N, int H, int W, int C, int stride, int dtype) {
    DwParams p = {};
    if (fill_params(p, N, H, W, C, stride, dtype)) return 0;
    float probe;
    p.part = &probe;   // geometry of the forward launch that writes statistics
    return dw_fwd_nsplit(p);
}

extern "C" int mpn_dwconv_fwd(const void* x, const float* w, void* y, int N, int H, int W, int C, int stride,
                              int dtype, const float* in_scale, const float* in_shift, int in_act, int flip,
                              float* stats_part, mpn_stream_t stream) {
    DwParams p = {};
    if (int rc = fill_params(p, N, H, W, C, stride, dtype)) return rc;
    MPN_REQUIRE(x && w && y, MPN_ERR_BAD_ARG, "dwconv_fwd: null pointer");
    MPN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), MPN_ERR_BAD_ARG, "dwconv_fwd: scale/shift mismatch");
    p.x = x; p.w = w; p.y = y; p.part = stats_part;
    p.in_scale = in_scale; p.in_shift = in_shift; p.in_act = in_act; p.flip = flip;
    hipStream_t st = (hipStream_t)stream;
    DwSwGeom g = dw_sw_geom(p);
    if (stats_part != nullptr && stride == 1 && dw_xt(p) == 2) p.swr = (p.OH + g.yblocks - 1) / g.yblocks;   // (the height dw_sw_geom chose)
    if (stats_part == nullptr) {   // (with statistics the strip height fixes the slab rows: mpn_dwconv_num_parts)
        int swr = sw_rows(p.OH, stride);
        while (swr > 4 && (long long)p.N * g.cblocks * g.yblocks * g.xblocks < 512) {
            swr >>= 1;
            p.swr = swr;
            g = dw_sw_geom(p);
        }
    }
    p.cblocks = g.cblocks;
    const long long blocks = (long long)p.N * g.cblocks * g.yblocks * g.xblocks;
    MPN_REQUIRE(blocks < (1ll << 31), MPN_ERR_BAD_SHAPE, "dwconv_fwd: grid too large");
    MPN_DISPATCH_DTYPE(dtype, {
        if (stride == 1 && dw_xt(p) == 2 && in_scale == nullptr) dwconv_fwd_sw2_kernel<T, false, true><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks);
        else if (stride == 1 && dw_xt(p) == 2) dwconv_fwd_sw2_kernel<T, false, false><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks);
        else if (stride == 1 && in_scale == nullptr) dwconv_fwd_sw_kernel<T, 1, false, true><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks);
        else if (stride == 1) dwconv_fwd_sw_kernel<T, 1><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks);
        else dwconv_fwd_sw_kernel<T, 2><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks);
    });
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

// stride-2 sliding-window data gradient: geometry (even H, W only; else the gather kernel)
struct DwDgS2Geom { bool ok; int ncg, cols, cblocks, xblocks, yblocks; };
static DwDgS2Geom dw_dg_s2_geom(const DwParams& p) {
    DwDgS2Geom g = {};
    const int cg_total = p.C / 4;
    g.ncg = cg_total < 32 ? cg_total : 32;
    g.ok = p.pad_t == 0 && p.pad_l == 0 && p.H == 2 * p.OH && p.W == 2 * p.OW && (g.ncg & (g.ncg - 1)) == 0;
    g.cols = kThreads / g.ncg;
    g.cblocks = (cg_total + g.ncg - 1) / g.ncg;
    g.xblocks = (p.OW + g.cols - 1) / g.cols;
    g.yblocks = (p.OH + dg_rows(p.OH) - 1) / dg_rows(p.OH);
    return g;
}

// bnr: fuse the batch-norm backward reduction of the layer that dx feeds (p.bnr_* and p.part set); needs the sliding-window kernels
static int dw_bwd_data_impl(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int stride, int dtype,
                            const DwParams* bnr, mpn_stream_t stream, const void* addend = nullptr) {
    DwParams p = {};
    if (int rc = fill_params(p, N, H, W, C, stride, dtype)) return rc;
    MPN_REQUIRE(dy && w && dx, MPN_ERR_BAD_ARG, "dwconv_bwd_data: null pointer");
    hipStream_t st = (hipStream_t)stream;
    MPN_REQUIRE(addend == nullptr || (stride == 2 && dw_dg_s2_geom(p).ok), MPN_ERR_BAD_SHAPE,
                "dwconv_bwd_data_add: needs stride 2, even H and W and a power-of-two channel block (mpn_dwconv_bwd_data_add_supported)");
    MPN_REQUIRE(addend != dx, MPN_ERR_BAD_ARG, "dwconv_bwd_data_add: addend must not alias dx");
    p.addend = addend;
    if (stride == 1) {   // correlation with the flipped kernel, pad 1
        if (bnr == nullptr)
            return mpn_dwconv_fwd(dy, w, dx, N, H, W, C, 1, dtype, nullptr, nullptr, MPN_ACT_NONE, 1, nullptr, stream);
        p.x = dy; p.w = w; p.y = dx; p.flip = 1; p.in_act = MPN_ACT_NONE;
        p.part = bnr->part; p.bnr_x = bnr->bnr_x; p.bnr_scale = bnr->bnr_scale; p.bnr_shift = bnr->bnr_shift;
        p.bnr_mean = bnr->bnr_mean; p.bnr_invstd = bnr->bnr_invstd; p.bnr_act = bnr->bnr_act;
        const DwSwGeom g = dw_sw_geom(p);
        p.cblocks = g.cblocks;
        const long long blocks = (long long)p.N * g.cblocks * g.yblocks * g.xblocks;
        MPN_REQUIRE(blocks < (1ll << 31), MPN_ERR_BAD_SHAPE, "dwconv_bwd_data: grid too large");
        MPN_DISPATCH_DTYPE(dtype, {
            if (dw_xt(p) == 2) dwconv_fwd_sw2_kernel<T, true, true><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks);
            else dwconv_fwd_sw_kernel<T, 1, true, true><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks);
        });
        MPN_LAUNCH_CHECK();
        return MPN_OK;
    }
    const DwDgS2Geom g = dw_dg_s2_geom(p);
    if (g.ok) {
        const long long blocks = (long long)N * g.cblocks * g.yblocks * g.xblocks;
        MPN_REQUIRE(blocks < (1ll << 31), MPN_ERR_BAD_SHAPE, "dwconv_bwd_data: grid too large");
        if (bnr != nullptr) {
            p.part = bnr->part; p.bnr_x = bnr->bnr_x; p.bnr_scale = bnr->bnr_scale; p.bnr_shift = bnr->bnr_shift;
            p.bnr_mean = bnr->bnr_mean; p.bnr_invstd = bnr->bnr_invstd; p.bnr_act = bnr->bnr_act;
            MPN_DISPATCH_DTYPE(dtype, {
                if (addend) dwconv_dgrad_s2_sw_kernel<T, true, true><<<(unsigned)blocks, kThreads, 0, st>>>(
                                (const T*)dy, w, (T*)dx, H, W, C, p.OH, p.OW, g.ncg, g.cols, g.xblocks, g.yblocks, g.cblocks, p);
                else dwconv_dgrad_s2_sw_kernel<T, true, false><<<(unsigned)blocks, kThreads, 0, st>>>(
                                (const T*)dy, w, (T*)dx, H, W, C, p.OH, p.OW, g.ncg, g.cols, g.xblocks, g.yblocks, g.cblocks, p);
            });
        } else {
            MPN_DISPATCH_DTYPE(dtype, {
                if (addend) dwconv_dgrad_s2_sw_kernel<T, false, true><<<(unsigned)blocks, kThreads, 0, st>>>(
                                (const T*)dy, w, (T*)dx, H, W, C, p.OH, p.OW, g.ncg, g.cols, g.xblocks, g.yblocks, g.cblocks, p);
                else dwconv_dgrad_s2_sw_kernel<T, false, false><<<(unsigned)blocks, kThreads, 0, st>>>(
                                (const T*)dy, w, (T*)dx, H, W, C, p.OH, p.OW, g.ncg, g.cols, g.xblocks, g.yblocks, g.cblocks, p);
            });
        }
        MPN_LAUNCH_CHECK();
        return MPN_OK;
    }
    MPN_REQUIRE(bnr == nullptr, MPN_ERR_BAD_SHAPE, "dwconv_bwd_data_bn: shape not supported (mpn_dwconv_bwd_data_bn_num_parts == 0)");
    const int ve = dtype == MPN_F32 ? 4 : 8;
    const long long total_vec = (long long)N * H * W * (C / ve);
    long long blocks = (total_vec + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    MPN_DISPATCH_DTYPE(dtype, (dwconv_dgrad_s2_kernel<T><<<(int)blocks, kThreads, 0, st>>>(
                                  (const T*)dy, w, (T*)dx, N, H, W, C, p.OH, p.OW, p.pad_t, p.pad_l, total_vec)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* data gradient: dy [N,OH,OW,C] -> dx [N,H,W,C] (H, W = the forward INPUT size) */
extern "C" int mpn_dwconv_bwd_data(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int stride,
                                   int dtype, mpn_stream_t stream) {
    return dw_bwd_data_impl(dy, w, dx, N, H, W, C, stride, dtype, nullptr, stream);
}

/* rows of the partial slab mpn_dwconv_bwd_data_bn writes ([rows][2][C], the layout of mpn_bn_bwd_reduce: finish with
 * mpn_bn_bwd_finalize(part, rows, C, N*H*W, ...)); 0 = the fused form is not available for this shape */
extern "C" int mpn_dwconv_bwd_data_bn_num_parts(int N, int H, int W, int C, int stride, int dtype) {
    DwParams p = {};
    if (fill_params(p, N, H, W, C, stride, dtype)) return 0;
    if (C % 4 != 0) return 0;
    if (stride == 1) {
        p.bnr_x = &p;   // geometry of the fused launch
        const DwSwGeom g = dw_sw_geom(p);
        if (kThreads % g.ncg != 0) return 0;
        return p.N * g.yblocks * g.xblocks;
    }
    const DwDgS2Geom g = dw_dg_s2_geom(p);
    return g.ok ? N * g.yblocks * g.xblocks : 0;
}

/* mpn_dwconv_bwd_data + the batch-norm backward REDUCTION of the layer whose activated output the depthwise conv read:
 * dx is that layer's dA, x_bn its raw conv output [N,H,W,C]; part receives sum(g), sum(g*xhat) per block
 * (g = dA * act'(x_bn*scale+shift), xhat = (x_bn-mean)*invstd) - one tensor read and one launch less than
 * mpn_bn_bwd_reduce(dx, x_bn, ...) after the fact. */
extern "C" int mpn_dwconv_bwd_data_bn(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int stride,
                                      int dtype, const void* x_bn, const float* scale, const float* shift, const float* mean,
                                      const float* invstd, int act, float* part, mpn_stream_t stream) {
    MPN_REQUIRE(x_bn && scale && shift && mean && invstd && part, MPN_ERR_BAD_ARG, "dwconv_bwd_data_bn: null pointer");
    MPN_REQUIRE(mpn_dwconv_bwd_data_bn_num_parts(N, H, W, C, stride, dtype) > 0, MPN_ERR_BAD_SHAPE,
                "dwconv_bwd_data_bn: shape not supported (mpn_dwconv_bwd_data_bn_num_parts == 0)");
    DwParams b = {};
    b.part = part; b.bnr_x = x_bn; b.bnr_scale = scale; b.bnr_shift = shift; b.bnr_mean = mean; b.bnr_invstd = invstd;
    b.bnr_act = act;
    return dw_bwd_data_impl(dy, w, dx, N, H, W, C, stride, dtype, &b, stream);
}

/* 1 when mpn_dwconv_bwd_data_add takes this shape (the stride-2 sliding-window kernel: even H and W, C % 4 == 0) */
extern "C" int mpn_dwconv_bwd_data_add_supported(int N, int H, int W, int C, int stride, int dtype) {
    DwParams p = {};
    if (fill_params(p, N, H, W, C, stride, dtype)) return 0;
    return (stride == 2 && C % 4 == 0 && dw_dg_s2_geom(p).ok) ? 1 : 0;
}

/* dx = data gradient + addend (addend: a tensor of dx's shape and type, e.g. the gradient an FPN lateral sends into the
 * same backbone feature map), optionally with the batch-norm backward reduction of mpn_dwconv_bwd_data_bn over that SUM
 * (x_bn == NULL: no reduction; then scale .. part are ignored). */
extern "C" int mpn_dwconv_bwd_data_add(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int stride,
                                       int dtype, const void* addend, const void* x_bn, const float* scale,
                                       const float* shift, const float* mean, const float* invstd, int act, float* part,
                                       mpn_stream_t stream) {
    MPN_REQUIRE(addend, MPN_ERR_BAD_ARG, "dwconv_bwd_data_add: null addend");
    MPN_REQUIRE(mpn_dwconv_bwd_data_add_supported(N, H, W, C, stride, dtype), MPN_ERR_BAD_SHAPE,
                "dwconv_bwd_data_add: shape not supported (mpn_dwconv_bwd_data_add_supported == 0)");
    if (x_bn == nullptr) return dw_bwd_data_impl(dy, w, dx, N, H, W, C, stride, dtype, nullptr, stream, addend);
    MPN_REQUIRE(scale && shift && mean && invstd && part, MPN_ERR_BAD_ARG, "dwconv_bwd_data_add: null batch-norm pointer");
    DwParams b = {};
    b.part = part; b.bnr_x = x_bn; b.bnr_scale = scale; b.bnr_shift = shift; b.bnr_mean = mean; b.bnr_invstd = invstd;
    b.bnr_act = act;
    return dw_bwd_data_impl(dy, w, dx, N, H, W, C, stride, dtype, &b, stream, addend);
}

// sliding-window weight gradient: blocks of at most 128 channels; strips of 32 output rows on the large maps, 16 below
struct DwWgSwGeom { int ncg, cols, xblocks, yblocks, cblocks, rows, units, xt; };
static DwWgSwGeom dw_wg_sw_geom(const DwParams& p) {
    DwWgSwGeom g;
    const int cg_total = p.C / 4;
    g.ncg = cg_total < 32 ? cg_total : 32;
    g.cols = kThreads / g.ncg;
    g.cblocks = (cg_total + g.ncg - 1) / g.ncg;
    // strip height by map size (measured per layer): stride 1: 64 / 32 / 16 output rows for maps of >= 128 / >= 64 / fewer
    // rows (128ch @128x128: 74 -> 62 us with 64); stride 2: 32 / 16 / 8
    const bool s1 = p.H == p.OH;
    g.rows = s1 ? (p.OH >= 128 ? 64 : (p.OH >= 64 ? 32 : 16)) : (p.OH >= 64 ? 32 : (p.OH >= 32 ? 16 : 8));
    g.xt = s1 ? 2 : 1;
    g.xblocks = (p.OW + g.cols * g.xt - 1) / (g.cols * g.xt);
    g.yblocks = (p.OH + g.rows - 1) / g.rows;
    g.units = p.N * g.yblocks * g.xblocks;
    return g;
}
static bool dw_wg_use_sw(const DwParams& p) {
    const int cg_total = p.C / 4;
    // (the thread map needs a power-of-two group count per block; other channel counts keep the LDS-tile kernel)
    const int ncg = cg_total < 32 ? cg_total : 32;
    return (ncg & (ncg - 1)) == 0 && kThreads % ncg == 0;
}

extern "C" int mpn_dwconv_wgrad_num_parts(int N, int H, int W, int C, int stride, int dtype) {
    DwParams p = {};
    if (fill_params(p, N, H, W, C, stride, dtype)) return 0;
    if (dw_wg_use_sw(p)) return dw_wg_sw_geom(p).units;
    const int ntiles = N * p.tiles_y * p.tiles_x;
    int nsplit = 2048 / p.cblocks;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > ntiles) nsplit = ntiles;
    return nsplit;
}

/* weight gradient partials: part [mpn_dwconv_wgrad_num_parts][9][C]; reduce with mpn_reduce_partials */
extern "C" int mpn_dwconv_bwd_weight(const void* x, const void* dy, float* part, int N, int H, int W, int C,
                                     int stride, int dtype, const float* in_scale, const float* in_shift, int in_act,
                                     mpn_stream_t stream) {
    DwParams p = {};
    if (int rc = fill_params(p, N, H, W, C, stride, dtype)) return rc;
    MPN_REQUIRE(x && dy && part, MPN_ERR_BAD_ARG, "dwconv_bwd_weight: null pointer");
    p.x = x; p.dy = dy; p.part = part;
    p.in_scale = in_scale; p.in_shift = in_shift; p.in_act = in_act;
    const int ve = dtype == MPN_F32 ? 4 : 8;
    const int nsplit = mpn_dwconv_wgrad_num_parts(N, H, W, C, stride, dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dw_wg_use_sw(p)) {
        const DwWgSwGeom g = dw_wg_sw_geom(p);
        p.cblocks = g.cblocks;
        const long long blocks = (long long)g.units * g.cblocks;
        MPN_REQUIRE(blocks < (1ll << 31), MPN_ERR_BAD_SHAPE, "dwconv_bwd_weight: grid too large");
        MPN_DISPATCH_DTYPE(dtype, {
            if (stride == 1 && g.xt == 2) dwconv_wgrad_sw2_kernel<T><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks, g.rows);
            else if (stride == 1) dwconv_wgrad_sw_kernel<T, 1><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks, g.rows);
            else dwconv_wgrad_sw_kernel<T, 2><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks, g.rows);
        });
        MPN_LAUNCH_CHECK();
        return MPN_OK;
    }
    const int grid = nsplit * p.cblocks;
    if (stride == 1) {
        const size_t sm = dw_smem<1>(ve, 36, p.nvg);
        MPN_DISPATCH_DTYPE(dtype, {
            if (int rc = set_smem(dwconv_wgrad_kernel<T, 1>, sm)) return rc;
            dwconv_wgrad_kernel<T, 1><<<grid, kThreads, sm, st>>>(p, nsplit);
        });
    } else {
        const size_t sm = dw_smem<2>(ve, 36, p.nvg);
        MPN_DISPATCH_DTYPE(dtype, {
            if (int rc = set_smem(dwconv_wgrad_kernel<T, 2>, sm)) return rc;
            dwconv_wgrad_kernel<T, 2><<<grid, kThreads, sm, st>>>(p, nsplit);
        });
    }
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* 1 when mpn_dwconv_bwd_fused takes this shape: stride 1, the sliding-window geometry of the weight gradient (its partial slab
 * has mpn_dwconv_wgrad_num_parts rows, and so has the fused reduction's) */
extern "C" int mpn_dwconv_bwd_fused_supported(int N, int H, int W, int C, int stride, int dtype) {
    DwParams p = {};
    if ((stride != 1 && stride != 2) || fill_params(p, N, H, W, C, stride, dtype)) return 0;
    if (C % 4 != 0 || !dw_wg_use_sw(p)) return 0;
    if (stride == 1) return dw_wg_sw_geom(p).xt == 2 ? 1 : 0;
    // stride 2 (mpn_dwconv_bwd_fused_s2): even H and W, and the two separate walks' strips coincide (they do by construction)
    const DwDgS2Geom d = dw_dg_s2_geom(p);
    const DwWgSwGeom g = dw_wg_sw_geom(p);
    return (d.ok && d.ncg == g.ncg && d.cols == g.cols && d.xblocks == g.xblocks && d.yblocks == g.yblocks && dg_rows(p.OH) == g.rows) ? 1 : 0;
}

/* Stride-1 depthwise backward in ONE pass over dy, x and dx (tf.nn.depthwise_conv2d's two gradients, mobilenet_v1.py:101, + the
 * batch-norm backward reduction of the layer that produced x): dx = data gradient [N,H,W,C]; wpart
 * [mpn_dwconv_wgrad_num_parts][9][C] = weight-gradient partials over act(x * in_scale + in_shift) (finish with mpn_reduce_partials);
 * bn_part (NULL: no reduction) [mpn_dwconv_wgrad_num_parts][2][C] = sum(g), sum(g * xhat) with g = dx where the activation passes,
 * xhat = (x - mean) * invstd (finish with mpn_bn_bwd_finalize). Replaces mpn_dwconv_bwd_weight + mpn_dwconv_bwd_data_bn: three
 * tensor passes instead of five. */
extern "C" int mpn_dwconv_bwd_fused(const void* x, const void* dy, const float* w, void* dx, float* wpart, int N, int H, int W, int C,
                                    int dtype, const float* in_scale, const float* in_shift, int in_act, const float* mean,
                                    const float* invstd, float* bn_part, mpn_stream_t stream) {
    MPN_REQUIRE(mpn_dwconv_bwd_fused_supported(N, H, W, C, 1, dtype), MPN_ERR_BAD_SHAPE,
                "dwconv_bwd_fused: shape not supported (mpn_dwconv_bwd_fused_supported == 0)");
    MPN_REQUIRE(x && dy && w && dx && wpart && in_scale && in_shift, MPN_ERR_BAD_ARG, "dwconv_bwd_fused: null pointer");
    MPN_REQUIRE(bn_part == nullptr || (mean && invstd), MPN_ERR_BAD_ARG, "dwconv_bwd_fused: the reduction needs mean and invstd");
    MPN_REQUIRE(dx != dy && dx != x, MPN_ERR_BAD_ARG, "dwconv_bwd_fused: dx must not alias an input");
    DwParams p = {};
    if (int rc = fill_params(p, N, H, W, C, 1, dtype)) return rc;
    p.x = x; p.dy = dy; p.w = w; p.y = dx; p.wpart = wpart; p.part = bn_part;
    p.in_scale = in_scale; p.in_shift = in_shift; p.in_act = in_act; p.bnr_mean = mean; p.bnr_invstd = invstd;
    const DwWgSwGeom g = dw_wg_sw_geom(p);
    p.cblocks = g.cblocks;
    const long long blocks = (long long)g.units * g.cblocks;
    MPN_REQUIRE(blocks < (1ll << 31), MPN_ERR_BAD_SHAPE, "dwconv_bwd_fused: grid too large");
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, {
        if (bn_part) dwconv_bwd_sw2_kernel<T, true><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks, g.rows);
        else dwconv_bwd_sw2_kernel<T, false><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks, g.rows);
    });
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* The stride-2 counterpart of mpn_dwconv_bwd_fused (even H and W; mpn_dwconv_bwd_fused_supported(.., stride 2, ..) == 1): x [N,H,W,C] the
 * conv's raw input, dy [N,H/2,W/2,C]; dx [N,H,W,C] = data gradient (+ addend when not NULL: the FPN lateral's gradient into the same
 * feature map, added before the store and before the reduction, as mpn_dwconv_bwd_data_add does); wpart
 * [mpn_dwconv_wgrad_num_parts(.., 2, ..)][9][C]; bn_part (NULL: no reduction) [same rows][2][C]. Replaces mpn_dwconv_bwd_weight +
 * mpn_dwconv_bwd_data_add / _bn: the input and dy are read once instead of twice. dx must not alias x, dy or addend. */
extern "C" int mpn_dwconv_bwd_fused_s2(const void* x, const void* dy, const float* w, void* dx, float* wpart, int N, int H, int W, int C,
                                       int dtype, const float* in_scale, const float* in_shift, int in_act, const float* mean,
                                       const float* invstd, float* bn_part, const void* addend, mpn_stream_t stream) {
    MPN_REQUIRE(mpn_dwconv_bwd_fused_supported(N, H, W, C, 2, dtype), MPN_ERR_BAD_SHAPE,
                "dwconv_bwd_fused_s2: shape not supported (mpn_dwconv_bwd_fused_supported == 0)");
    MPN_REQUIRE(x && dy && w && dx && wpart && in_scale && in_shift, MPN_ERR_BAD_ARG, "dwconv_bwd_fused_s2: null pointer");
    MPN_REQUIRE(bn_part == nullptr || (mean && invstd), MPN_ERR_BAD_ARG, "dwconv_bwd_fused_s2: the reduction needs mean and invstd");
    MPN_REQUIRE(dx != dy && dx != x && dx != addend, MPN_ERR_BAD_ARG, "dwconv_bwd_fused_s2: dx must not alias an input");
    DwParams p = {};
    if (int rc = fill_params(p, N, H, W, C, 2, dtype)) return rc;
    p.x = x; p.dy = dy; p.w = w; p.y = dx; p.wpart = wpart; p.part = bn_part; p.addend = addend;
    p.in_scale = in_scale; p.in_shift = in_shift; p.in_act = in_act; p.bnr_mean = mean; p.bnr_invstd = invstd;
    const DwWgSwGeom g = dw_wg_sw_geom(p);
    p.cblocks = g.cblocks;
    const long long blocks = (long long)g.units * g.cblocks;
    MPN_REQUIRE(blocks < (1ll << 31), MPN_ERR_BAD_SHAPE, "dwconv_bwd_fused_s2: grid too large");
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, {
        if (bn_part && addend) dwconv_bwd_s2_kernel<T, true, true><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks, g.rows);
        else if (bn_part) dwconv_bwd_s2_kernel<T, true, false><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks, g.rows);
        else if (addend) dwconv_bwd_s2_kernel<T, false, true><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks, g.rows);
        else dwconv_bwd_s2_kernel<T, false, false><<<(unsigned)blocks, kThreads, 0, st>>>(p, g.ncg, g.cols, g.xblocks, g.yblocks, g.rows);
    });
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
