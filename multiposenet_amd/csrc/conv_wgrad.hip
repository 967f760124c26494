// Weight gradients of the dense 1x1 / 3x3 convolutions on the MFMA matrix cores, NHWC.
//   dW[tap][ci][co] = sum over pixels  act(bn(x))[pixel+tap][ci] * dY[pixel][co]
// i.e. a GEMM whose reduction dimension is the PIXEL axis (hundreds of thousands long) and whose
// output is tiny, so: split-K over pixel tiles, register accumulation across a long tile loop, one
// partial slab per split, deterministic reduction afterwards (mpn_reduce_partials).
//
// Both operands are pixel-major in memory (channels contiguous) while the MFMA wants 8
// consecutive k (= pixels) per lane. bf16: the LDS images stay row-major [pixel][channel] exactly
// as loaded (16-byte coalesced) and the fragments are fetched with ds_read_b64_tr_b16, the gfx950
// transposing LDS read (4 pixels x 16 channels per 16-lane group). f32 parity build: one
// ds_read_b32 per element and v_mfma_f32_16x16x4_f32.
//   3x3: a block owns 32 input channels (64-byte rows) x all 9 taps x 128 output channels; the
//        10x18 halo image of the input patch serves all 9 taps (tap = address offset);
//   1x1: a block owns a 128 x 128 (ci x co) tile, classic 2x2 wave layout.
// LDS rows are XOR-swizzled in 32-byte segments so the 8 pixel rows one half-wave touches per
// transposed read land in 8 different bank groups.
#include "common.h"
#include <type_traits>

namespace {

// fragment types / MFMA / transposing read of the 16-bit storage type T (bf16 or fp16; a float T never reaches them)
template <typename T> using HT = H16<std::conditional_t<sizeof(T) == 2, T, bf16_t>>;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kHaloW = 18, kHaloH = 10;

struct WgradParams {
    const void* x;    // [N,H,W,Cin]
    const void* dy;   // [N,H,W,Cout]
    float* part;      // [nsplit][TAPS][Cin][Cout]
    const float* in_scale;
    const float* in_shift;
    int in_act;
    int N, H, W, Cin, Cout;
    int xs, dys;      // elements between consecutive pixels of x / dy (Cin / Cout when dense)
    int tiles_x, tiles_y;
    long long M;
    int ntiles;   // pixel tiles
    int nsplit;
    int n_cg, n_cb;
    int xcd_remap;    // XCD-aware block -> work map inside the job (needs the job's first block on XCD 0)
    // fused thin pointwise backward (mpn_conv1x1_bwd_fused; the DG variant of the kernel): the layer's f32 kernel [Cin][Cout], the
    // data gradient [M][Cin] (pixel stride dxs, written MASKED by the activation of the batch-norm that produced x) and the partial
    // rows [nsplit][2][Cin] of that batch-norm's backward reduction (sums of g and of g * x with the RAW x)
    const float* wf;
    void* dx;
    int dxs;
    float* bn_part;
    // ... with the batch-norm backward APPLY pass of the layer's OWN batch-norm folded into the dY staging (DAPPLY): dy is then the
    // gradient w.r.t. the layer's activated output, ap_x the layer's raw output (pixel stride ap_xs), ap_* that batch-norm's affine,
    // saved statistics and the k1 / k2 of mpn_bn_bwd_finalize
    const void* ap_x;
    int ap_xs;
    const float* ap_scale; const float* ap_shift; const float* ap_mean; const float* ap_invstd; const float* ap_k1; const float* ap_k2;
    int ap_act;
#ifdef MPN_DIAG
    unsigned long long* dbg;   // diagnostic build only (mpn_diag_set_wgrad_stamps): per-wave phase times, else NULL
#endif
};

__device__ __forceinline__ int fsw256(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
// 64-byte rows stay linear: their transposed reads are at worst 2-way conflicted (rows r and r+8 of a
// half-wave), which is noise next to the MFMAs, and linear rows make every tap a constant address offset.
__device__ __forceinline__ int fsw64(int) { return 0; }
// 128-byte rows (4 segments, two rows per 256-byte bank row): row parity separates neighbours, (r>>1)&1 and (r>>3)&1
// separate the rest of the 8 rows (r..r+3, r+8..r+11) a half-wave touches per transposed read.
__device__ __forceinline__ int fsw128(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); }

template <int RB> __device__ __forceinline__ int lds_off(int row, int seg, int within) {
    const int f = (RB == 64) ? fsw64(row) : (RB == 128 ? fsw128(row) : fsw256(row));
    return row * RB + ((seg ^ f) << 5) + within;
}

// four storage elements <-> f32 (the fused data gradient's 8-byte pieces)
template <typename T> struct Raw4h { uint2 v; };
__device__ __forceinline__ void raw4_unpack(const Raw4h<bf16_t>& r, float (&f)[4]) {
    f[0] = __uint_as_float(r.v.x << 16); f[1] = __uint_as_float(r.v.x & 0xffff0000u);
    f[2] = __uint_as_float(r.v.y << 16); f[3] = __uint_as_float(r.v.y & 0xffff0000u);
}
__device__ __forceinline__ void raw4_unpack(const Raw4h<half_t>& r, float (&f)[4]) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t a = __builtin_bit_cast(h2_t, r.v.x), b = __builtin_bit_cast(h2_t, r.v.y);
    f[0] = (float)a[0]; f[1] = (float)a[1]; f[2] = (float)b[0]; f[3] = (float)b[1];
}
template <typename T> __device__ __forceinline__ float round_to_storage(float v);
template <> __device__ __forceinline__ float round_to_storage<bf16_t>(float v) { return to_f32((bf16_t)v); }
template <> __device__ __forceinline__ float round_to_storage<half_t>(float v) { return (float)(_Float16)v; }
template <typename T> __device__ __forceinline__ uint2 pack4_storage(const float (&g)[4]);
template <> __device__ __forceinline__ uint2 pack4_storage<bf16_t>(const float (&g)[4]) {
    return make_uint2(pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3]));
}
template <> __device__ __forceinline__ uint2 pack4_storage<half_t>(const float (&g)[4]) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t a = {(_Float16)g[0], (_Float16)g[1]}, b = {(_Float16)g[2], (_Float16)g[3]};
    return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}

template <typename H> __device__ __forceinline__ typename H::x4 tr_read(const unsigned char* base, int off) {
    return H::tr_read(base + off);
}

// RBA / RBD: bytes per pixel row of the A (input-channel) and dY (output-channel) LDS images. Default geometry:
// 3x3 -> (64, 256) = 32 ci x 128 co per block; 1x1 -> (256, 256). Layers with Cout <= 64 (final_conv3x3) use
// (128, 128) = 64 ci x 64 co so that no MFMA column is spent on zero padding.
template <typename T, int TAPS, int RBA, int RBD>
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int ES = (int)sizeof(T);
    constexpr int VE = 16 / ES;
    constexpr int CG = RBA / ES;                   // input channels per block
    constexpr int BNW = RBD / ES;                  // output channels per block
    constexpr int DSLOTS = RBD / 16;
    constexpr int NPIXA = TAPS == 9 ? kHaloW * kHaloH : 128;
    constexpr int MT_TOTAL = CG / 16, NT_TOTAL = BNW / 16;
    // wave layout: 2x2 over (m-tiles, n-tiles) when there are >= 2 m-tiles, else 1x4
    constexpr int WM = MT_TOTAL >= 2 ? 2 : 1;
    constexpr int WN = 4 / WM;
    constexpr int MTW = MT_TOTAL / WM;
    constexpr int NTW = NT_TOTAL / WN;
    constexpr int KPIX = ES == 2 ? 32 : 4;         // pixels per MFMA k-step
    constexpr int KSTEPS = 128 / KPIX;
    constexpr int ASLOTS = RBA / 16;
    constexpr int AVEC = (NPIXA * ASLOTS + kThreads - 1) / kThreads;
    constexpr int DVEC = 128 * DSLOTS / kThreads;  // 8 (or 4)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                      // [NPIXA][RBA]
    unsigned char* Ds = smem + NPIXA * RBA;        // [128][RBD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;

    // XCD-aware remap: hardware deals blocks round-robin over the 8 XCDs (block b and b+8 share an L2); work items
    // are ordered ci-group fastest, so giving each XCD a contiguous run of work items puts the ci-groups that re-read
    // one dY tile behind the same L2 (speed only; bijective for any grid size).
    int b;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int cg = b % p.n_cg; b /= p.n_cg;
    const int cb = b % p.n_cb;
    const int split = b / p.n_cb;
    const int ci0 = cg * CG, co0 = cb * BNW;

    const int mt0 = (wave / WN) * MTW;
    const int nt0 = (wave % WN) * NTW;

    f32x4_t acc[TAPS][MTW][NTW];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);

    // per-thread constants of the A staging: 16-byte slot -> channels, affine
    const int aslot = tid % ASLOTS;
    const int ace = ci0 + aslot * VE;
    const bool acvalid = ace < p.Cin;
    const bool affine = p.in_scale != nullptr;
    const int dslot = tid % DSLOTS;
    const int dce = co0 + dslot * VE;
    const bool dcvalid = dce < p.Cout;

    for (int tile = split; tile < p.ntiles; tile += p.nsplit) {
        int img = 0, oy0 = 0, ox0 = 0;
        long long m0 = 0;
        if (TAPS == 9) {
            const int tx = tile % p.tiles_x;
            const int t2 = tile / p.tiles_x;
            const int ty = t2 % p.tiles_y;
            img = t2 / p.tiles_y;
            oy0 = ty * 8;
            ox0 = tx * 16;
        } else {
            m0 = (long long)tile * 128;
        }
        __syncthreads();  // previous tile's fragment reads are done
        // ---------------- stage A (activated input, halo for 3x3) and the dY tile: ALL global loads of the tile are
        // issued before the first one is consumed (one memory round trip per tile instead of three)
        {
            Vec16<T> v[AVEC];
            bool inb[AVEC];
            Vec16<T> dv[DVEC];
#pragma unroll
            for (int i = 0; i < AVEC; ++i) {
                const int vi = tid + i * kThreads;
                const int pix = vi / ASLOTS;
                bool ok = acvalid && vi < NPIXA * ASLOTS;
                long long off = 0;
                if (TAPS == 9) {
                    const int hy = pix / kHaloW, hx = pix - hy * kHaloW;
                    const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                    ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    off = (((long long)img * p.H + iy) * p.W + ix) * p.xs + ace;
                } else {
                    const long long m = m0 + pix;
                    ok = ok && m < p.M;
                    off = m * p.xs + ace;
                }
                inb[i] = ok;
                if (ok) v[i].load(x + off); else v[i].zero();
            }
#pragma unroll
            for (int i = 0; i < DVEC; ++i) {
                const int r = (tid / DSLOTS) + i * (kThreads / DSLOTS);
                bool ok = dcvalid;
                long long off = 0;
                if (TAPS == 9) {
                    const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
                    ok = ok && oy < p.H && ox < p.W;
                    off = (((long long)img * p.H + oy) * p.W + ox) * p.dys + dce;
                } else {
                    const long long m = m0 + r;
                    ok = ok && m < p.M;
                    off = m * p.dys + dce;
                }
                if (ok) dv[i].load(dy + off); else dv[i].zero();
            }
#pragma unroll
            for (int i = 0; i < AVEC; ++i) {
                const int vi = tid + i * kThreads;
                if (vi < NPIXA * ASLOTS) {
                    const int pix = vi / ASLOTS;
                    if (affine && inb[i]) {
                        float f[VE];
                        v[i].unpack(f);
#pragma unroll
                        for (int j = 0; j < VE; ++j) {
                            float t = f[j] * p.in_scale[ace + j] + p.in_shift[ace + j];
                            if (p.in_act != MPN_ACT_NONE) t = fmaxf(t, 0.f);
                            if (p.in_act == MPN_ACT_RELU6) t = fminf(t, 6.f);
                            f[j] = t;
                        }
                        v[i].pack(f);
                    }
                    *reinterpret_cast<uint4*>(As + lds_off<RBA>(pix, aslot >> 1, (aslot & 1) * 16)) =
                        *reinterpret_cast<const uint4*>(&v[i].raw);
                }
            }
#pragma unroll
            for (int i = 0; i < DVEC; ++i) {
                const int r = (tid / DSLOTS) + i * (kThreads / DSLOTS);
                *reinterpret_cast<uint4*>(Ds + lds_off<RBD>(r, dslot >> 1, (dslot & 1) * 16)) =
                    *reinterpret_cast<const uint4*>(&dv[i].raw);
            }
        }
        __syncthreads();

        // ---------------- MFMA over the 128 pixels of the tile
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if constexpr (ES == 2) {
                using H = HT<T>;
                using X8 = typename H::x8;
                using X4 = typename H::x4;
                const int qq = l15 >> 2, pp = l15 & 3;
                const int r0 = ks * 32 + 8 * lq + qq;  // this lane's pixel row for the first 4-row block
                X8 bfr[NTW];
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const X4 lo = tr_read<H>(Ds, lds_off<RBD>(r0, nt0 + j, pp * 8));
                    const X4 hi = tr_read<H>(Ds, lds_off<RBD>(r0 + 4, nt0 + j, pp * 8));
                    bfr[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                // software-pipelined over (tap, m-tile): the fragment of step n+1 is fetched while the
                // MFMAs of step n issue; sched_group_barrier pins that interleave so hipcc does not hoist all
                // 18 fragment reads (72 VGPRs) above the MFMAs and spill the accumulators.
                auto a_row = [&](int t) -> int {
                    if (TAPS == 9) {
                        const int ty = 2 * ks + (lq >> 1), tx0 = 8 * (lq & 1);
                        return (ty + t / 3) * kHaloW + tx0 + (t % 3) + qq;
                    }
                    return r0;
                };
                auto a_load = [&](int step) -> X8 {
                    const int t = step / MTW, i = step % MTW;
                    const int arow = a_row(t);
                    const X4 lo = tr_read<H>(As, lds_off<RBA>(arow, mt0 + i, pp * 8));
                    const X4 hi = tr_read<H>(As, lds_off<RBA>(arow + 4, mt0 + i, pp * 8));
                    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                X8 a_next = a_load(0);
#pragma unroll
                for (int step = 0; step < TAPS * MTW; ++step) {
                    const X8 afr = a_next;
                    if (step + 1 < TAPS * MTW) a_next = a_load(step + 1);
                    const int t = step / MTW, i = step % MTW;
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[t][i][j] = H::mfma(afr, bfr[j], acc[t][i][j]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // 2 DS reads (next fragment)
                    __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);  // NTW MFMAs (this fragment)
                }
            } else {
                const int r = ks * 4 + lq;  // pixel of this lane's k index
                float bfr[NTW];
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const int c = (nt0 + j) * 16 + l15;  // channel (float index) in the 64-float row
                    bfr[j] = *reinterpret_cast<const float*>(Ds + lds_off<RBD>(r, c >> 3, (c & 7) * 4));
                }
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int arow = (TAPS == 9) ? (((r >> 4) + t / 3) * kHaloW + (r & 15) + (t % 3)) : r;
#pragma unroll
                    for (int i = 0; i < MTW; ++i) {
                        const int c = (mt0 + i) * 16 + l15;
                        const float afr = *reinterpret_cast<const float*>(As + lds_off<RBA>(arow, c >> 3, (c & 7) * 4));
#pragma unroll
                        for (int j = 0; j < NTW; ++j)
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr, bfr[j], acc[t][i][j], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---------------- write this split's partial [TAPS][Cin][Cout]
    float* __restrict__ dst = p.part + (long long)split * TAPS * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + (mt0 + i) * 16 + lq * 4 + r;
                    const int co = co0 + (nt0 + j) * 16 + l15;
                    if (ci < p.Cin && co < p.Cout)
                        dst[((long long)t * p.Cin + ci) * p.Cout + co] = acc[t][i][j][r];
                }
}

// ------------------------------------------------------------------------------------------------------------
// bf16 throughput kernel: 8 waves, ONE block per CU, software-pipelined over pixel tiles.
// The 4-wave kernel above stalls a whole memory round trip per tile (load -> LDS -> barrier -> MFMA, two blocks per CU
// cannot cover it: measured 21 % MFMA utilisation on the 128x128-map 3x3 layers). Here a block owns twice the input
// channels (fewer re-reads of every dY tile), the NEXT tile's global loads are issued into registers before the
// current tile's MFMAs start and are committed (batch-norm affine + activation) into the OTHER half of a
// double-buffered LDS image afterwards, so that one barrier per tile remains and the round trip hides under
// ~4.6k MFMA cycles per SIMD.  Geometries (RBA, RBD bytes per LDS pixel row; WM x WN waves over m/n tiles):
//   3x3:              (128, 256, 4x2) = 64 ci x 9 taps x 128 co, 144 accumulator registers per lane (a wave: one m-tile x four n-tiles,
//                     0.72 fragment reads per MFMA; as 2 x 2 tiles 1.11 and 0.3 % of the step slower)
//   3x3, Cout <= 64:  (256, 128, 4x2) = 128 ci x 9 taps x 64 co
//   1x1:              (256, 256, 4x2) = 128 ci x 128 co
// (a device function: the plain kernel and the grouped kernel - several independent layers of one channel geometry, e.g. the
//  four pyramid levels of a subnet stage, in ONE grid of one block per CU - share it; blk / nblk = this job's block index and count)
// DG (thin 1x1 layers, one block = all input and output channels): the DATA gradient of the same tile rides along. Both of the weight
// gradient's LDS images are what it needs - dA^T[ci][px] = W[ci][:] . dY[px][:] takes the dY rows as they lie (plain 16-byte reads, no
// transpose) and the layer's kernel from registers (bf16 fragments, loaded once) - plus the RAW input rows for the mask and the
// sums of the batch-norm reduction (a third image, written by the same commit). Separately the two gradients read dY twice and x
// twice; here once each: 536 -> 402... MB on Conv2d_1_pointwise. A wave owns 16 of the tile's 128 pixels for this part.
template <typename T, int TAPS, int RBA, int RBD, int WM, bool STAGGER, bool DG = false, bool DAPPLY = false>
__device__ __forceinline__ void conv_wgrad_bf16_body(const WgradParams& p, const int blk, const int nblk) {
    using H = H16<T>;
    using X8 = typename H::x8;
    using X4 = typename H::x4;
    constexpr int NT = 512;
    constexpr int VE = 8;
    constexpr int CG = RBA / 2, BNW = RBD / 2;
    constexpr int ASLOTS = RBA / 16, DSLOTS = RBD / 16;
    constexpr int NPIXA = TAPS == 9 ? kHaloW * kHaloH : 128;
    constexpr int MT_TOTAL = CG / 16, NT_TOTAL = BNW / 16;
    constexpr int WN = 8 / WM;
    constexpr int MTW = MT_TOTAL / WM, NTW = NT_TOTAL / WN;
    static_assert(MTW * WM == MT_TOTAL && NTW * WN == NT_TOTAL, "wave layout must tile the block");
    constexpr int AVEC = (NPIXA * ASLOTS + NT - 1) / NT;
    constexpr int DVEC = 128 * DSLOTS / NT;
    // LDS images: padded rows (stride = row bytes + 32, an odd number of 32-byte bank groups), NO swizzle: the 8
    // consecutive pixel rows a half-wave touches per transposed read fall into 8 different bank groups, and every
    // (tap, k-step, m-tile) of a fragment read is an IMMEDIATE offset from one per-lane base address. (XOR-swizzled
    // rows cost ~10 VALU instructions per fragment read: 2 waves x 250 VALU x 4 cycles per k-step against 1152 MFMA
    // cycles - the MFMA phase was VALU-bound.)
    constexpr int SA = RBA + 32, SD = RBD + 32;
    constexpr int ABYTES = NPIXA * SA, DBYTES = 128 * SD;
    static_assert(!DG || (TAPS == 1 && !STAGGER), "the fused data gradient: 1x1, waves in step");
    constexpr int OS = RBA + 16;                       // DG: pixel-row stride of the output image
    constexpr int RBYTES = DG ? 128 * SA : 0, OBYTES = DG ? 128 * OS : 0;
    constexpr int KS2 = BNW / 32;                      // DG: k-steps of the data gradient (over the output channels)

    // Waves in step (no STAGGER) commit the next tile only behind the barrier that ends this tile's reads: ONE image of each operand
    // would do. The weight-gradient launches keep two (nothing else wants the LDS); DG needs the room for its two extra images.
    constexpr int NBUF = DG ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [NBUF][ABYTES] A images, [NBUF][DBYTES] dY images, [2][CG] f32 scale / shift; DG: + raw input image, output image, [8][2][CG] sums
    float* aff = reinterpret_cast<float*>(smem + NBUF * ABYTES + NBUF * DBYTES);
    unsigned char* Rs = smem + NBUF * ABYTES + NBUF * DBYTES + 2 * CG * (int)sizeof(float);
    unsigned char* Os = Rs + RBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;

    int b = blk;
    if (p.xcd_remap) {   // (the job's first block sits on XCD 0: see the launchers)
        const int nwg = nblk, q = nwg >> 3, r = nwg & 7, xcd = blk & 7, k = blk >> 3;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int cg = b % p.n_cg; b /= p.n_cg;
    const int cb = b % p.n_cb;
    const int split = b / p.n_cb;
    const int ci0 = cg * CG, co0 = cb * BNW;
    const int mt0 = (wave / WN) * MTW, nt0 = (wave % WN) * NTW;

    f32x4_t acc[TAPS][MTW][NTW];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
    const bool affine = p.in_scale != nullptr;
    const float lo = (affine && p.in_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (affine && p.in_act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    for (int c = tid; c < CG; c += NT) {
        const bool ok = affine && ci0 + c < p.Cin;
        aff[c] = ok ? p.in_scale[ci0 + c] : 1.f;
        aff[CG + c] = ok ? p.in_shift[ci0 + c] : 0.f;
    }

    const int aslot = tid % ASLOTS;
    const int dslot = tid % DSLOTS;

    Vec16<T> v[AVEC], dv[DVEC];
    Vec16<T> ev[DAPPLY ? DVEC : 1];            // DAPPLY: the layer's raw output at the dY pieces
    unsigned inb = 0u, inbd = 0u;
    static_assert(!DAPPLY || DG, "the apply pass folds into the fused thin backward");
    // DAPPLY: dY = sc * g + (cb * y + cc), g = the incoming gradient where lo < y * sc + sh < hi, rounded to storage - the expression
    // and the rounding of bn_bwd_apply_body (bn.hip); a thread's channels (its 16-byte slot) are fixed, the constants fold once
    // (kept in LDS, [4][BNW] floats behind the sums' scratch, and read per piece: as 32 registers per thread they spill on the 64 x 128 tile)
    float* apc = reinterpret_cast<float*>(Os + OBYTES + (DG ? 8 * 2 * CG * (int)sizeof(float) : 0));
    float alo = -INFINITY, ahi = INFINITY;
    if constexpr (DAPPLY) {
        for (int c = tid; c < BNW; c += NT) {
            const bool ok = co0 + c < p.Cout;
            const float s_ = ok ? p.ap_scale[co0 + c] : 0.f, h_ = ok ? p.ap_shift[co0 + c] : 0.f, m_ = ok ? p.ap_mean[co0 + c] : 0.f;
            const float i_ = ok ? p.ap_invstd[co0 + c] : 0.f, a_ = ok ? p.ap_k1[co0 + c] : 0.f, b_ = ok ? p.ap_k2[co0 + c] : 0.f;
            apc[c] = s_; apc[BNW + c] = h_;
            apc[2 * BNW + c] = -s_ * b_ * i_;
            apc[3 * BNW + c] = -s_ * (a_ - m_ * i_ * b_);
        }
        alo = (p.ap_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
        ahi = (p.ap_act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    }

    // Per-thread invariants of the staging (32-bit element offsets relative to the tile origin; the host checks that
    // both tensors have < 2^31 elements). Halo position packed as hy << 8 | hx.
    int relA[AVEC], hyx[AVEC], relD[DVEC];
    {
        const int ace = ci0 + aslot * VE, dce = co0 + dslot * VE;
#pragma unroll
        for (int i = 0; i < AVEC; ++i) {
            const int vi = tid + i * NT;
            const int pix = vi / ASLOTS;
            const bool ok = ace < p.Cin && vi < NPIXA * ASLOTS;
            if (TAPS == 9) {
                const int hy = pix / kHaloW, hx = pix - hy * kHaloW;
                relA[i] = ((hy - 1) * p.W + (hx - 1)) * p.xs + ace;
                hyx[i] = ok ? (hy << 8 | hx) : 0x7fff00;        // hy = 32767: never inside an image
            } else {
                relA[i] = pix * p.xs + ace;
                hyx[i] = ok ? pix : 0x7fffffff;
            }
        }
#pragma unroll
        for (int i = 0; i < DVEC; ++i) {
            const int r = (tid / DSLOTS) + i * (NT / DSLOTS);
            relD[i] = (TAPS == 9 ? ((r >> 4) * p.W + (r & 15)) : r) * p.dys + dce;
        }
    }
    const bool dvalid = co0 + dslot * VE < p.Cout;
    const int drow0 = tid / DSLOTS;
    // 1x1: the tile's pieces as buffer loads. A piece's byte offset is a per-thread constant + the tile's base, and rows past the end
    // of the tensor (the ragged last tile) or channels past the layer's (offset 2^31: the host keeps both tensors below 2^31 bytes)
    // fail the descriptor's range check and arrive as zeros: no compare / select / 64-bit address per load, no zeroing at commit.
    // (A zero row of x may become act(shift) under the affine - it meets a zero row of dY. Under DAPPLY a zero dY row would become the
    //  apply's constant term: those pieces keep their mask.)
    constexpr bool BUF = TAPS == 1;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    unsigned relAb[BUF ? AVEC : 1], relDb[BUF ? DVEC : 1], relEb[(BUF && DAPPLY) ? DVEC : 1];
    __amdgpu_buffer_rsrc_t rsx, rsd, rse;
    if constexpr (BUF) {
        constexpr unsigned kOut = 0x80000000u;
#pragma unroll
        for (int i = 0; i < AVEC; ++i) relAb[i] = hyx[i] == 0x7fffffff ? kOut : (unsigned)relA[i] * (unsigned)sizeof(T);
#pragma unroll
        for (int i = 0; i < DVEC; ++i) relDb[i] = dvalid ? (unsigned)relD[i] * (unsigned)sizeof(T) : kOut;
        rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)(((p.M - 1) * p.xs + p.Cin) * (long long)sizeof(T)), 0x00020000);
        rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dy), 0, (int)(((p.M - 1) * p.dys + p.Cout) * (long long)sizeof(T)), 0x00020000);
        if constexpr (DAPPLY) {
#pragma unroll
            for (int i = 0; i < DVEC; ++i)
                relEb[i] = dvalid ? (unsigned)((drow0 + i * (NT / DSLOTS)) * p.ap_xs + co0 + dslot * VE) * (unsigned)sizeof(T) : kOut;
            rse = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.ap_x), 0, (int)(((p.M - 1) * p.ap_xs + p.Cout) * (long long)sizeof(T)), 0x00020000);
        }
    }

    // Loads are UNCONDITIONAL (out-of-image lanes read element 0 of the tensor and are zeroed at commit time): a
    // predicated load is lowered to load + select, which waits for the data right here and defeats the prefetch.
    // Addresses and masks first, then the loads back to back.
    auto load_tile = [&](int tile) {
        if constexpr (BUF) {
            const unsigned tbA = (unsigned)tile * 128u * (unsigned)p.xs * (unsigned)sizeof(T);
            const unsigned tbD = (unsigned)tile * 128u * (unsigned)p.dys * (unsigned)sizeof(T);
            if constexpr (DAPPLY) {
                const int mleft = (int)(p.M - (long long)tile * 128);
                unsigned md = 0u;
#pragma unroll
                for (int i = 0; i < DVEC; ++i) md |= ((dvalid && drow0 + i * (NT / DSLOTS) < mleft) ? 1u : 0u) << i;
                inbd = md;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < AVEC; ++i) {
                const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)(relAb[i] + tbA), 0, 0);
                v[i].raw = make_uint4(q.x, q.y, q.z, q.w);
            }
#pragma unroll
            for (int i = 0; i < DVEC; ++i) {
                const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rsd, (int)(relDb[i] + tbD), 0, 0);
                dv[i].raw = make_uint4(q.x, q.y, q.z, q.w);
            }
            if constexpr (DAPPLY) {
                const unsigned tbE = (unsigned)tile * 128u * (unsigned)p.ap_xs * (unsigned)sizeof(T);
#pragma unroll
                for (int i = 0; i < DVEC; ++i) {
                    const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rse, (int)(relEb[i] + tbE), 0, 0);
                    ev[i].raw = make_uint4(q.x, q.y, q.z, q.w);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        int oy0 = 0, ox0 = 0, baseA, baseD, mleft = 0;
        if (TAPS == 9) {
            const int tx = tile % p.tiles_x;
            const int t2 = tile / p.tiles_x;
            const int ty = t2 % p.tiles_y;
            const int img = t2 / p.tiles_y;
            oy0 = ty * 8;
            ox0 = tx * 16;
            const int pix0 = (img * p.H + oy0) * p.W + ox0;
            baseA = pix0 * p.xs;
            baseD = pix0 * p.dys;
        } else {
            baseA = tile * 128 * p.xs;
            baseD = tile * 128 * p.dys;
            mleft = (int)(p.M - (long long)tile * 128);      // valid pixels from the tile origin on
        }
        int oa[AVEC], od[DVEC];
        unsigned ma = 0u, md = 0u;
#pragma unroll
        for (int i = 0; i < AVEC; ++i) {
            bool ok;
            if (TAPS == 9) {
                const int iy = oy0 + (hyx[i] >> 8) - 1, ix = ox0 + (hyx[i] & 255) - 1;
                ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            } else {
                ok = hyx[i] < mleft;
            }
            oa[i] = ok ? baseA + relA[i] : 0;
            ma |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < DVEC; ++i) {
            const int r = drow0 + i * (NT / DSLOTS);
            bool ok = dvalid;
            if (TAPS == 9) ok = ok && oy0 + (r >> 4) < p.H && ox0 + (r & 15) < p.W;
            else ok = ok && r < mleft;
            od[i] = ok ? baseD + relD[i] : 0;
            md |= (ok ? 1u : 0u) << i;
        }
        inb = ma;
        inbd = md;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < AVEC; ++i) v[i].load(x + oa[i]);
#pragma unroll
        for (int i = 0; i < DVEC; ++i) dv[i].load(dy + od[i]);
        if constexpr (DAPPLY) {
            // (dense layers: dy and the raw output share the pixel stride when ap_xs == dys; else the offset is rescaled per piece)
            const T* __restrict__ ax = reinterpret_cast<const T*>(p.ap_x);
#pragma unroll
            for (int i = 0; i < DVEC; ++i) {
                const int r = drow0 + i * (NT / DSLOTS);
                const bool ok = (inbd >> i) & 1u;
                ev[i].load(ax + (ok ? (long long)(tile * 128 + r) * p.ap_xs + co0 + dslot * VE : 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    auto commit_tile = [&](unsigned char* As, unsigned char* Ds) {
        float sc[VE], sh[VE];
        if (affine) {
#pragma unroll
            for (int j = 0; j < VE; j += 4) {
                const float4 a = *reinterpret_cast<const float4*>(aff + aslot * VE + j);
                const float4 c = *reinterpret_cast<const float4*>(aff + CG + aslot * VE + j);
                sc[j] = a.x; sc[j + 1] = a.y; sc[j + 2] = a.z; sc[j + 3] = a.w;
                sh[j] = c.x; sh[j + 1] = c.y; sh[j + 2] = c.z; sh[j + 3] = c.w;
            }
        }
#pragma unroll
        for (int i = 0; i < AVEC; ++i) {
            const int vi = tid + i * NT;
            if (vi < NPIXA * ASLOTS) {
                const int pix = vi / ASLOTS;
                uint4 q = *reinterpret_cast<const uint4*>(&v[i].raw);
                if (affine) {
                    float f[VE];
                    v[i].unpack(f);
#pragma unroll
                    for (int j = 0; j < VE; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * sc[j] + sh[j], lo, hi);
                    Vec16<T> o;
                    o.pack(f);
                    q = *reinterpret_cast<const uint4*>(&o.raw);
                }
                if (!BUF && !((inb >> i) & 1u)) q = make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4*>(As + pix * SA + aslot * 16) = q;
                if constexpr (DG) {   // (single-buffered: read in front of the tile's second barrier, rewritten behind it)
                    uint4 r = *reinterpret_cast<const uint4*>(&v[i].raw);
                    if (!BUF && !((inb >> i) & 1u)) r = make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(Rs + pix * SA + aslot * 16) = r;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < DVEC; ++i) {
            const int r = (tid / DSLOTS) + i * (NT / DSLOTS);
            uint4 q = *reinterpret_cast<const uint4*>(&dv[i].raw);
            if constexpr (DAPPLY) {
                float d[VE], f[VE], asc[VE], ash[VE], acb[VE], acc_[VE];
                dv[i].unpack(d);
                ev[i].unpack(f);
#pragma unroll
                for (int j4 = 0; j4 < VE; j4 += 4) {
                    const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(apc + dslot * VE + j4), q1 = *reinterpret_cast<const f32x4_t*>(apc + BNW + dslot * VE + j4);
                    const f32x4_t q2 = *reinterpret_cast<const f32x4_t*>(apc + 2 * BNW + dslot * VE + j4), q3 = *reinterpret_cast<const f32x4_t*>(apc + 3 * BNW + dslot * VE + j4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { asc[j4 + j] = q0[j]; ash[j4 + j] = q1[j]; acb[j4 + j] = q2[j]; acc_[j4 + j] = q3[j]; }
                }
#pragma unroll
                for (int j = 0; j < VE; ++j) {
                    const float pre = f[j] * asc[j] + ash[j];
                    const float gg = (pre > alo && pre < ahi) ? d[j] : 0.f;
                    d[j] = asc[j] * gg + (acb[j] * f[j] + acc_[j]);
                }
                Vec16<T> o;
                o.pack(d);
                q = *reinterpret_cast<const uint4*>(&o.raw);
            }
            if ((!BUF || DAPPLY) && !((inbd >> i) & 1u)) q = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(Ds + r * SD + dslot * 16) = q;
        }
    };

#ifdef MPN_DIAG
    unsigned long long ph[4] = {0, 0, 0, 0}, t_prev = 0;
#define MPN_WG_STAMP(k) do { if (p.dbg) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                                          ph[k] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define MPN_WG_STAMP(k) do { } while (0)
#endif
    // Schedule. The 8 waves form two groups (waves 0-3 / 4-7: one wave of each group per SIMD) that run the SAME loop
    //     load(next) ; barrier ; MFMAs(tile i) ; barrier ; commit(next)
    // half a period apart: group 1 executes one barrier more before the loop and group 0 one more after it, so while
    // one group multiplies, the other one commits its share of a later tile to the other LDS half and issues its next
    // loads - the matrix pipe of every SIMD always has one wave feeding it, and staging costs no MFMA time as long
    // as it is shorter than one tile's MFMAs of a single wave. Group 0 stages one tile ahead of its MFMAs, group 1
    // two; a tile's image is complete one barrier before the first group reads it, and nobody writes the half that
    // the other group still reads (see DESIGN.md 4 for the barrier-by-barrier table).
    // The prefetch registers are written (load) and read (commit) inside ONE iteration - carried across the back
    // edge they become PHIs whose live ranges hipcc splits with a copy right after the loads, which waits for them.
    const int grp = STAGGER ? wave >> 2 : 0;   // !STAGGER: all waves in step (one group, two barriers per tile)
    const int ntl = p.ntiles > split ? (p.ntiles - split + p.nsplit - 1) / p.nsplit : 0;   // tiles of this block
    const int ahead = 1 + grp;
    // DG: the layer's kernel as bf16 fragments W[ci = mt * 16 + l15][co = ks * 32 + lq * 8 ..] (the rounding of the packed weights the
    // separate data gradient multiplies with), zero beyond the layer's channels; running sums of the reduction per lane:
    // channels mt * 16 + lq * 4 + r over the lane's pixel column l15 of every tile
    // (from four channel tiles on, two waves share two pixel tiles and split the channel tiles - four waves and four pixel tiles from
    //  eight on: a half / a quarter of the fragment registers per wave)
    constexpr int DGS = MT_TOTAL >= 8 ? 4 : (MT_TOTAL >= 4 ? 2 : 1), MTH = MT_TOTAL / DGS;
    const int dg_mt0 = DG ? (wave % DGS) * MTH : 0, dg_pt0 = DG ? (wave / DGS) * DGS : 0;
    X8 wfr[DG ? MTH : 1][DG ? KS2 : 1];
    float bsum[DG ? MTH : 1][4], bsq[DG ? MTH : 1][4];
    if constexpr (DG) {
#pragma unroll
        for (int mt = 0; mt < MTH; ++mt) {
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                const int ci = ci0 + (dg_mt0 + mt) * 16 + l15, co = co0 + ks * 32 + lq * 8;
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (ci < p.Cin && co + j < p.Cout) ? p.wf[(long long)ci * p.Cout + co + j] : 0.f;
                Vec16<T> o;
                o.pack(f);
                wfr[mt][ks] = __builtin_bit_cast(X8, o.raw);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { bsum[mt][r] = 0.f; bsq[mt][r] = 0.f; }
        }
    }
    if (ntl > 0) load_tile(split);
    __syncthreads();   // scale / shift table visible
    if (ntl > 0) commit_tile(smem, smem + NBUF * ABYTES);
    if (grp == 1) {
        if (ntl > 1) load_tile(split + p.nsplit);
        __syncthreads();
        if (ntl > 1) commit_tile(smem + ABYTES, smem + NBUF * ABYTES + DBYTES);
    }
#ifdef MPN_DIAG
    if (p.dbg) t_prev = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int it = 0; it < ntl; ++it) {
        const int buf = NBUF == 2 ? (it & 1) : 0;
        unsigned char* As = smem + buf * ABYTES;
        unsigned char* Ds = smem + NBUF * ABYTES + buf * DBYTES;
        const bool more = it + ahead < ntl;
        if (more) load_tile(split + (it + ahead) * p.nsplit);   // flies under this tile's MFMAs
        MPN_WG_STAMP(1);
        __syncthreads();
        MPN_WG_STAMP(2);
        // k (pixel) order inside a 32-pixel k-step: the first transposed read of lane group lq covers pixels
        // 4*lq..4*lq+3 of the even tile row, the second the same columns of the odd tile row (any order works as long
        // as A and dY agree); a half-wave therefore reads 8 consecutive pixel rows of the image.
        const int qq = l15 >> 2, pp = l15 & 3;
        const unsigned char* a_lane = As + (4 * lq + qq) * SA + pp * 8 + mt0 * 32;
        const unsigned char* d_lane = Ds + (4 * lq + qq) * SD + pp * 8 + nt0 * 32;
        // One flat, fully unrolled stream of 4 k-steps x STEPS fragment steps: the fragment ring and the double-
        // buffered dY fragments run ACROSS k-step boundaries (a rolled k-step loop drains the LDS pipeline 4x per tile).
        constexpr int STEPS = TAPS * MTW;
        constexpr int TOTAL = 4 * STEPS;
        constexpr int RING0 = NTW >= 4 ? 4 : ((TAPS == 9 && RBA == 256) ? 4 : 6);   // (the 64-output 3x3 geometry: six spill three registers)
        constexpr int RING = RING0 < STEPS ? RING0 : STEPS;
        constexpr int KSA = (TAPS == 9 ? 2 * kHaloW : 32) * SA, KSD = 32 * SD;
        auto a_load = [&](int gs) -> X8 {
            const int ks = gs / STEPS, step = gs % STEPS;
            const int t = step / MTW, i = step % MTW;
            const int row = TAPS == 9 ? (t / 3) * kHaloW + (t % 3) : 0;
            const int second = TAPS == 9 ? kHaloW : 16;
            const X4 l4 = tr_read<H>(a_lane, ks * KSA + row * SA + i * 32);
            const X4 h4 = tr_read<H>(a_lane, ks * KSA + (row + second) * SA + i * 32);
            return __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        auto b_load = [&](int ks, int j) -> X8 {
            const X4 l4 = tr_read<H>(d_lane, ks * KSD + j * 32);
            const X4 h4 = tr_read<H>(d_lane, ks * KSD + 16 * SD + j * 32);
            return __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        X8 bfr[2][NTW];
        X8 ar[RING];
#pragma unroll
        for (int j = 0; j < NTW; ++j) bfr[0][j] = b_load(0, j);
#pragma unroll
        for (int r = 0; r < RING - 1; ++r) ar[r] = a_load(r);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gs = 0; gs < TOTAL; ++gs) {
            const int ks = gs / STEPS, step = gs % STEPS;
            // the next k-step's dY fragments are requested one step before its first A fragment (LDS returns in order)
            if (step == STEPS - RING && ks + 1 < 4) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) bfr[(ks + 1) & 1][j] = b_load(ks + 1, j);
            }
            if (gs + RING - 1 < TOTAL) ar[(gs + RING - 1) % RING] = a_load(gs + RING - 1);
            const int t = step / MTW, i = step % MTW;
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                acc[t][i][j] = H::mfma(ar[gs % RING], bfr[ks & 1][j], acc[t][i][j]);
            // full fence per step: without it the scheduler sinks every read down to its use (one register set,
            // lgkmcnt(0) before each MFMA group) and the ring degenerates to distance 0
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DG) {
            const bool bnr = p.bn_part != nullptr;
            // dA^T[ci][px] of this wave's pixel tiles: A operand = the kernel's fragment (rows = ci), B operand = the dY rows as they
            // lie (k = co, columns = px); a lane ends up with channels (dg_mt0 + mt) * 16 + lq * 4 .. + 3 of pixel (dg_pt0 + j) * 16 + l15
            // (a real loop over the pixel tiles: unrolled, hipcc hoists every tile's fragment reads and spills on the 128-channel tile)
#pragma unroll 1
            for (int j = 0; j < DGS; ++j) {
                const int px = (dg_pt0 + j) * 16 + l15;
                f32x4_t dacc[MTH];
#pragma unroll
                for (int mt = 0; mt < MTH; ++mt) dacc[mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
                    const X8 dfr = *reinterpret_cast<const X8*>(Ds + px * SD + ks * 64 + lq * 16);
#pragma unroll
                    for (int mt = 0; mt < MTH; ++mt) dacc[mt] = H::mfma(wfr[mt][ks], dfr, dacc[mt]);
                }
                // mask by the activation of the batch-norm that produced x (lo < x * scale + shift < hi on the RAW x), round to
                // storage, sums of g and g * x, the tile's image for whole-row stores
#pragma unroll
                for (int mt = 0; mt < MTH; ++mt) {
                    const int cl = (dg_mt0 + mt) * 16 + lq * 4;
                    Raw4h<T> r4;
                    r4.v = *reinterpret_cast<const uint2*>(Rs + px * SA + cl * 2);
                    float xf[4];
                    raw4_unpack(r4, xf);
                    const f32x4_t sc4 = *reinterpret_cast<const f32x4_t*>(aff + cl), sh4 = *reinterpret_cast<const f32x4_t*>(aff + CG + cl);
                    float g[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pre = xf[r] * sc4[r] + sh4[r];
                        const float d = round_to_storage<T>(dacc[mt][r]);
                        g[r] = (!bnr || (pre > lo && pre < hi)) ? d : 0.f;      // (no reduction asked for: the plain data gradient)
                        bsum[mt][r] += g[r];
                        bsq[mt][r] += g[r] * xf[r];
                    }
                    *reinterpret_cast<uint2*>(Os + px * OS + cl * 2) = pack4_storage<T>(g);
                }
            }
        }
        MPN_WG_STAMP(3);
        __syncthreads();
        MPN_WG_STAMP(2);
        if (more) {
            const int nb = NBUF == 2 ? ((it + ahead) & 1) : 0;
            commit_tile(smem + nb * ABYTES, smem + NBUF * ABYTES + nb * DBYTES);
        }
        if constexpr (DG) {
            // the tile's 128 x Cin gradients leave as 16-byte pieces of whole pixel rows (rows past the end / channels past Cin: skipped)
            const int tile = split + it * p.nsplit;
            const int mleft = (int)(p.M - (long long)tile * 128);
            T* __restrict__ dxp = reinterpret_cast<T*>(p.dx) + (long long)tile * 128 * p.dxs + ci0;
            for (int v0 = tid; v0 < 128 * ASLOTS; v0 += NT) {
                const int pr = v0 / ASLOTS, sl = v0 - pr * ASLOTS;
                if (pr < mleft && ci0 + sl * 8 < p.Cin)
                    *reinterpret_cast<uint4*>(dxp + (long long)pr * p.dxs + sl * 8) = *reinterpret_cast<const uint4*>(Os + pr * OS + sl * 16);
            }
        }
        MPN_WG_STAMP(0);
    }
    if (STAGGER && grp == 0) __syncthreads();
#ifdef MPN_DIAG
    if (p.dbg && lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p.dbg[((size_t)blockIdx.x * 8 + wave) * 4 + k] = ph[k];
    }
#endif
#undef MPN_WG_STAMP

    if constexpr (DG) {
        // sums over the 16 pixel lanes of a (mt, lq) group (fixed butterfly), then over the 8 waves through LDS in wave order
        float* red = reinterpret_cast<float*>(Os + OBYTES);       // [8 waves][2][CG] (a wave fills its own channel tiles, the rest stays 0)
        __syncthreads();                                           // (the last tile's copy-out has read the output image)
        for (int o = tid; o < 8 * 2 * CG; o += NT) red[o] = 0.f;
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MTH; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = bsum[mt][r], b = bsq[mt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (l15 == 0) {
                    red[(wave * 2 + 0) * CG + (dg_mt0 + mt) * 16 + lq * 4 + r] = a;
                    red[(wave * 2 + 1) * CG + (dg_mt0 + mt) * 16 + lq * 4 + r] = b;
                }
            }
        __syncthreads();
        for (int o = tid; o < 2 * CG; o += NT) {
            const int which = o / CG, c = o - which * CG;
            if (p.bn_part != nullptr && ci0 + c < p.Cin) {
                float a = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) a += red[(w8 * 2 + which) * CG + c];
                p.bn_part[((long long)split * 2 + which) * p.Cin + ci0 + c] = a;
            }
        }
    }
    float* __restrict__ dst = p.part + (long long)split * TAPS * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + (mt0 + i) * 16 + lq * 4 + r;
                    const int co = co0 + (nt0 + j) * 16 + l15;
                    if (ci < p.Cin && co < p.Cout) {
                        // the slab is read once, by the batched reduction at the end of the step: non-temporal
                        __builtin_nontemporal_store(acc[t][i][j][r], &dst[((long long)t * p.Cin + ci) * p.Cout + co]);
                    }
                }
}

template <typename T, int TAPS, int RBA, int RBD, int WM, bool STAGGER>
__global__ __launch_bounds__(512, 1) void conv_wgrad_bf16_kernel(const WgradParams p) {
    conv_wgrad_bf16_body<T, TAPS, RBA, RBD, WM, STAGGER>(p, blockIdx.x, gridDim.x);
}

template <typename T, int RBA, int RBD, int WM, bool DAPPLY>
__global__ __launch_bounds__(512, 1) void conv1x1_bwd_fused_kernel(const WgradParams p) {
    conv_wgrad_bf16_body<T, 1, RBA, RBD, WM, false, true, DAPPLY>(p, blockIdx.x, gridDim.x);
}

// up to five independent layers of one (Cin, Cout, ksize) in one grid: as launches of their own the small pyramid levels are
// latency-bound tails, and every launch splits its pixels over all 256 CUs, i.e. writes (and the slab reduction reads) 128
// partial copies of the 590 KB gradient per LEVEL; in one grid the CUs are divided among the levels by their pixel counts
constexpr int kMaxWgradGroup = 5;
struct WgradGroup {
    WgradParams p[kMaxWgradGroup];
    int begin[kMaxWgradGroup + 1];   // first block of each job; begin[njobs] = grid size
    int njobs;
};
template <typename T, int TAPS, int RBA, int RBD, int WM, bool STAGGER>
__global__ __launch_bounds__(512, 1) void conv_wgrad_bf16_grouped_kernel(const WgradGroup g) {
    int job = 0;
#pragma unroll
    for (int j = 1; j < kMaxWgradGroup; ++j)
        if (j < g.njobs && (int)blockIdx.x >= g.begin[j]) job = j;   // wave-uniform
    conv_wgrad_bf16_body<T, TAPS, RBA, RBD, WM, STAGGER>(g.p[job], (int)blockIdx.x - g.begin[job], g.begin[job + 1] - g.begin[job]);
}

struct WgradGeom {
    int n_cg, n_cb, ntiles, nsplit;
};

// the thinnest pointwise layer (Conv2d_1_pointwise 32 -> 64 at 256 x 256: 2 M pixels): a 32 x 64 block tile - on the 128 x 128 tile
// three quarters of the staged A bytes and of the MFMAs were padding, and the launch ran at 3.1 TB/s of its 403 MB
static bool wgrad_thin1x1(int Cin, int Cout, int ksize) { return ksize == 1 && Cin <= 32 && Cout <= 64; }

WgradGeom wgrad_geom(int N, int H, int W, int Cin, int Cout, int ksize, int es) {
    WgradGeom g;
    int cgsz, bnw, blocks;
    if (es == 2) {   // conv_wgrad_bf16_kernel: one 8-wave block per CU
        const bool narrow = ksize == 3 && Cout <= 64;
        cgsz = (ksize == 3 && !narrow) ? 64 : 128;
        bnw = narrow ? 64 : 128;
        if (wgrad_thin1x1(Cin, Cout, ksize)) { cgsz = 32; bnw = 64; }
        blocks = 256;
    } else {
        const bool narrow = ksize == 3 && Cout * es <= 128;   // (128,128) geometry, see launch_wgrad
        cgsz = (narrow ? 128 : (ksize == 3 ? 64 : 256)) / es;
        bnw = (narrow ? 128 : 256) / es;
        blocks = 512;
    }
    g.n_cg = (Cin + cgsz - 1) / cgsz;
    g.n_cb = (Cout + bnw - 1) / bnw;
    g.ntiles = ksize == 3 ? N * ((H + 7) / 8) * ((W + 15) / 16) : (int)(((long long)N * H * W + 127) / 128);
    int ns = blocks / (g.n_cg * g.n_cb);
    if (ns < 1) ns = 1;
    // small maps: at least 4 pixel tiles per block - a slab per tile costs more HBM traffic than the layer's inputs
    const int min_tiles = es == 2 ? 4 : 1;
    if (ns > g.ntiles / min_tiles) ns = g.ntiles / min_tiles;
    if (ns < 1) ns = 1;
    g.nsplit = ns;
    return g;
}

template <typename T, int TAPS, int RBA, int RBD, int WM, bool STAGGER>
int launch_wgrad_bf16(const WgradParams& p, hipStream_t st) {
    constexpr int NPIXA = TAPS == 9 ? kHaloW * kHaloH : 128;
    constexpr int smem = 2 * NPIXA * (RBA + 32) + 2 * 128 * (RBD + 32) + 2 * (RBA / 2) * (int)sizeof(float);
    static_assert(smem <= 160 * 1024, "LDS budget");
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)conv_wgrad_bf16_kernel<T, TAPS, RBA, RBD, WM, STAGGER>, smem, &attr_mask));
    conv_wgrad_bf16_kernel<T, TAPS, RBA, RBD, WM, STAGGER><<<dim3((unsigned)(p.n_cg * p.n_cb * p.nsplit)), dim3(512), smem, st>>>(p);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

template <typename T, int TAPS, int RBA, int RBD>
int launch_wgrad_g(const WgradParams& p, hipStream_t st) {
    constexpr int NPIXA = TAPS == 9 ? kHaloW * kHaloH : 128;
    constexpr int smem = NPIXA * RBA + 128 * RBD;
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)conv_wgrad_kernel<T, TAPS, RBA, RBD>, smem, &attr_mask));
    conv_wgrad_kernel<T, TAPS, RBA, RBD><<<dim3((unsigned)(p.n_cg * p.n_cb * p.nsplit)), dim3(kThreads), smem, st>>>(p);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

template <typename T, int TAPS>
int launch_wgrad(const WgradParams& p, hipStream_t st) {
    constexpr int ES = (int)sizeof(T);
    if (TAPS == 9 && p.Cout * ES <= 128) return launch_wgrad_g<T, TAPS, 128, 128>(p, st);
    return launch_wgrad_g<T, TAPS, (TAPS == 9 ? 64 : 256), 256>(p, st);
}

#ifdef MPN_DIAG
void* g_wgrad_dbg = nullptr;
#endif

}  // namespace

#ifdef MPN_DIAG
/* diagnostic build only (tools/stamp_wgrad.py): buf = u64 [blocks][8 waves][4 phases] of s_memtime ticks, or NULL */
extern "C" void mpn_diag_set_wgrad_stamps(void* buf) { g_wgrad_dbg = buf; }
#endif

extern "C" int mpn_conv_wgrad_num_parts(int N, int H, int W, int Cin, int Cout, int ksize, int dtype) {
    return wgrad_geom(N, H, W, Cin, Cout, ksize, dtype == MPN_F32 ? 4 : 2).nsplit;
}

namespace {
template <typename T, int TAPS, int RBA, int RBD, int WM, bool STAGGER>
int launch_wgrad_bf16_grouped(const WgradGroup& g, int blocks, hipStream_t st);
// A single 3x3 layer with more than 64 outputs runs as a group of one: as a kernel of its own (the parameters in the kernel's
// arguments instead of behind the job index) hipcc spills four registers of this 256-register geometry.
template <typename T>
int launch_wgrad_single_as_group(const WgradParams& p, hipStream_t st) {
    WgradGroup grp = {};
    grp.p[0] = p;
    const int blocks = p.n_cg * p.n_cb * p.nsplit;
    grp.begin[0] = 0;
    for (int j = 1; j <= kMaxWgradGroup; ++j) grp.begin[j] = blocks;
    grp.njobs = 1;
    return launch_wgrad_bf16_grouped<T, 9, 128, 256, 4, true>(grp, blocks, st);
}
}  // namespace

/* part: [mpn_conv_wgrad_num_parts()][ksize*ksize][Cin][Cout] f32 (HWIO per split); finish with mpn_reduce_partials */
extern "C" int mpn_conv_bwd_weight(const void* x, const void* dy, float* part, int N, int H, int W, int Cin, int Cout,
                                   int x_stride, int dy_stride, int ksize, int dtype, const float* in_scale,
                                   const float* in_shift, int in_act, mpn_stream_t stream) {
    MPN_REQUIRE(ksize == 1 || ksize == 3, MPN_ERR_BAD_SHAPE, "conv wgrad: ksize must be 1 or 3");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16 || dtype == MPN_F16, MPN_ERR_BAD_DTYPE, "conv wgrad: dtype %d", dtype);
    const int es = dtype == MPN_F32 ? 4 : 2;
    const int ve = 16 / es;
    MPN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % ve == 0 && Cout % ve == 0, MPN_ERR_BAD_SHAPE,
                "conv wgrad: bad shape (channels must be multiples of %d)", ve);
    MPN_REQUIRE(x && dy && part, MPN_ERR_BAD_ARG, "conv wgrad: null pointer");
    MPN_REQUIRE(mpn_aligned16(x) && mpn_aligned16(dy), MPN_ERR_BAD_ALIGN, "conv wgrad: pointers must be 16-byte aligned");
    MPN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), MPN_ERR_BAD_ARG, "conv wgrad: scale/shift mismatch");
    MPN_REQUIRE((long long)N * H * W * Cin < (1ll << 31) && (long long)N * H * W * Cout < (1ll << 31), MPN_ERR_BAD_SHAPE,
                "conv wgrad: tensors must have fewer than 2^31 elements");
    const WgradGeom g = wgrad_geom(N, H, W, Cin, Cout, ksize, es);
    WgradParams p;
    p.x = x; p.dy = dy; p.part = part;
    p.in_scale = in_scale; p.in_shift = in_shift; p.in_act = in_act;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.xs = x_stride > 0 ? x_stride : Cin; p.dys = dy_stride > 0 ? dy_stride : Cout;
    MPN_REQUIRE(p.xs >= Cin && p.dys >= Cout && p.xs % ve == 0 && p.dys % ve == 0, MPN_ERR_BAD_SHAPE,
                "conv wgrad: pixel strides (%d, %d) must be multiples of %d not below the channel counts", x_stride, dy_stride, ve);
    MPN_REQUIRE((long long)N * H * W * p.xs < (1ll << 31) && (long long)N * H * W * p.dys < (1ll << 31), MPN_ERR_BAD_SHAPE,
                "conv wgrad: strided tensors must span fewer than 2^31 elements");
    // (the 16-bit 1x1 kernel addresses its tiles with 32-bit byte offsets and keeps 2^31 as the out-of-range offset)
    MPN_REQUIRE(ksize != 1 || es != 2 || ((long long)N * H * W * p.xs < (1ll << 30) && (long long)N * H * W * p.dys < (1ll << 30)),
                MPN_ERR_BAD_SHAPE, "conv wgrad: 16-bit 1x1 tensors must span fewer than 2^31 bytes");
    p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 7) / 8;
    p.M = (long long)N * H * W;
    p.ntiles = g.ntiles; p.nsplit = g.nsplit; p.n_cg = g.n_cg; p.n_cb = g.n_cb; p.xcd_remap = 1;
#ifdef MPN_DIAG
    p.dbg = (unsigned long long*)g_wgrad_dbg;
#endif
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPN_F32) return ksize == 3 ? launch_wgrad<float, 9>(p, st) : launch_wgrad<float, 1>(p, st);
    // the two wave groups run staggered by half a period on the 3x3 geometries, in step on 1x1 (measured best per geometry)
    if (dtype == MPN_F16) {   // same kernel on v_mfma_f32_16x16x32_f16
        if (wgrad_thin1x1(Cin, Cout, ksize)) return launch_wgrad_bf16<half_t, 1, 64, 128, 2, false>(p, st);
        if (ksize == 1) return launch_wgrad_bf16<half_t, 1, 256, 256, 4, false>(p, st);
        if (Cout <= 64) return launch_wgrad_bf16<half_t, 9, 256, 128, 4, true>(p, st);
        return launch_wgrad_single_as_group<half_t>(p, st);
    }
    if (wgrad_thin1x1(Cin, Cout, ksize)) return launch_wgrad_bf16<bf16_t, 1, 64, 128, 2, false>(p, st);
    if (ksize == 1) return launch_wgrad_bf16<bf16_t, 1, 256, 256, 4, false>(p, st);
    if (Cout <= 64) return launch_wgrad_bf16<bf16_t, 9, 256, 128, 4, true>(p, st);
    return launch_wgrad_single_as_group<bf16_t>(p, st);
}

namespace {

// Splits of the jobs of a group: the block budget (one 8-wave block per CU) divided by pixel tiles, in multiples of `quantum`
// splits (so that every job starts on XCD 0 and its XCD-aware work map holds), at least one quantum, at least 4 tiles per split.
void wgrad_group_splits(int njobs, int N, const int* H, const int* W, int Cin, int Cout, int ksize, int* nsplit, WgradGeom* geoms) {
    long long total_tiles = 0;
    for (int j = 0; j < njobs; ++j) {
        geoms[j] = wgrad_geom(N, H[j], W[j], Cin, Cout, ksize, 2);
        total_tiles += geoms[j].ntiles;
    }
    const int units = geoms[0].n_cg * geoms[0].n_cb;          // blocks per split (shared by the jobs: same channel geometry)
    int quantum = 1;
    while ((quantum * units) % 8 != 0) ++quantum;              // blocks per job in multiples of 8
    int budget = 256 / units;                                  // splits in all
    if (budget < quantum * njobs) budget = quantum * njobs;
    // large jobs in multiples of the quantum (their XCD map), small ones rounded to the nearest split: a job rounded DOWN to a
    // quantum is the grid's long pole (level 4 of the subnet: 6 splits wanted, 4 given -> 64 instead of 43 tiles per block)
    int used = 0, big = 0;
    for (int j = 0; j < njobs; ++j) {
        const double ideal = (double)budget * geoms[j].ntiles / (double)total_tiles;
        long long ns = ideal >= 4.0 * quantum ? (long long)ideal / quantum * quantum : (long long)(ideal + 0.5);
        const int cap = geoms[j].ntiles / 4 > 0 ? geoms[j].ntiles / 4 : 1;   // at least 4 tiles per split
        if (ns > cap) ns = cap;
        if (ns < 1) ns = 1;
        nsplit[j] = (int)ns;
        used += (int)ns;
        if (geoms[j].ntiles > geoms[big].ntiles) big = j;
    }
    // the largest job absorbs what rounding left over / took too much (in quanta, so that it keeps its XCD map)
    while (used > budget && nsplit[big] > quantum) { nsplit[big] -= quantum; used -= quantum; }
    while (used + quantum <= budget && nsplit[big] + quantum <= geoms[big].ntiles / 4) { nsplit[big] += quantum; used += quantum; }
}

template <typename T, int TAPS, int RBA, int RBD, int WM, bool STAGGER>
int launch_wgrad_bf16_grouped(const WgradGroup& g, int blocks, hipStream_t st) {
    constexpr int NPIXA = TAPS == 9 ? kHaloW * kHaloH : 128;
    constexpr int smem = 2 * NPIXA * (RBA + 32) + 2 * 128 * (RBD + 32) + 2 * (RBA / 2) * (int)sizeof(float);
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)conv_wgrad_bf16_grouped_kernel<T, TAPS, RBA, RBD, WM, STAGGER>, smem, &attr_mask));
    conv_wgrad_bf16_grouped_kernel<T, TAPS, RBA, RBD, WM, STAGGER><<<dim3((unsigned)blocks), dim3(512), smem, st>>>(g);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

bool wgrad_groupable(int njobs, int dtype) { return njobs >= 2 && njobs <= kMaxWgradGroup && (dtype == MPN_BF16 || dtype == MPN_F16); }

}  // namespace

/* Partial-slab counts of mpn_conv_bwd_weight_grouped, one per job (jobs that cannot share a grid - f32, more than 5 - get the
 * counts of their separate launches, which is what the grouped call then does). */
extern "C" int mpn_conv_wgrad_grouped_num_parts(int njobs, int N, const int* H, const int* W, int Cin, int Cout, int ksize, int dtype,
                                                int* nparts) {
    MPN_REQUIRE(njobs > 0 && H && W && nparts, MPN_ERR_BAD_ARG, "conv wgrad grouped: bad arguments");
    if (!wgrad_groupable(njobs, dtype)) {
        for (int j = 0; j < njobs; ++j) nparts[j] = mpn_conv_wgrad_num_parts(N, H[j], W[j], Cin, Cout, ksize, dtype);
        return MPN_OK;
    }
    WgradGeom geoms[kMaxWgradGroup];
    wgrad_group_splits(njobs, N, H, W, Cin, Cout, ksize, nparts, geoms);
    return MPN_OK;
}

/* The weight gradients of njobs independent layers of one (Cin, Cout, ksize, dtype) - e.g. the pyramid levels of a subnet stage
 * (keypoint_subnet.py:66-79: one phi_subnet per level) - in one grid. part[j]: [nparts[j]][ksize*ksize][Cin][Cout] f32. */
extern "C" int mpn_conv_bwd_weight_grouped(int njobs, const void* const* x, const void* const* dy, float* const* part, int N,
                                           const int* H, const int* W, int Cin, int Cout, const int* x_stride, const int* dy_stride,
                                           int ksize, int dtype, const float* const* in_scale, const float* const* in_shift,
                                           int in_act, mpn_stream_t stream) {
    MPN_REQUIRE(njobs > 0 && x && dy && part && H && W && in_scale && in_shift, MPN_ERR_BAD_ARG, "conv wgrad grouped: bad arguments");
    if (!wgrad_groupable(njobs, dtype)) {
        for (int j = 0; j < njobs; ++j)
            if (int rc = mpn_conv_bwd_weight(x[j], dy[j], part[j], N, H[j], W[j], Cin, Cout, x_stride ? x_stride[j] : 0,
                                             dy_stride ? dy_stride[j] : 0, ksize, dtype, in_scale[j], in_shift[j], in_act, stream))
                return rc;
        return MPN_OK;
    }
    MPN_REQUIRE(ksize == 1 || ksize == 3, MPN_ERR_BAD_SHAPE, "conv wgrad grouped: ksize must be 1 or 3");
    MPN_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, MPN_ERR_BAD_SHAPE, "conv wgrad grouped: bad shape");
    WgradGeom geoms[kMaxWgradGroup];
    int nsplit[kMaxWgradGroup];
    wgrad_group_splits(njobs, N, H, W, Cin, Cout, ksize, nsplit, geoms);
    WgradGroup grp = {};
    int begin = 0;
    for (int j = 0; j < njobs; ++j) {
        MPN_REQUIRE(x[j] && dy[j] && part[j] && H[j] > 0 && W[j] > 0, MPN_ERR_BAD_ARG, "conv wgrad grouped: null pointer / bad size");
        MPN_REQUIRE(mpn_aligned16(x[j]) && mpn_aligned16(dy[j]), MPN_ERR_BAD_ALIGN, "conv wgrad grouped: pointers must be 16-byte aligned");
        MPN_REQUIRE((in_scale[j] == nullptr) == (in_shift[j] == nullptr), MPN_ERR_BAD_ARG, "conv wgrad grouped: scale/shift mismatch");
        WgradParams& p = grp.p[j];
        p.x = x[j]; p.dy = dy[j]; p.part = part[j];
        p.in_scale = in_scale[j]; p.in_shift = in_shift[j]; p.in_act = in_act;
        p.N = N; p.H = H[j]; p.W = W[j]; p.Cin = Cin; p.Cout = Cout;
        p.xs = (x_stride && x_stride[j] > 0) ? x_stride[j] : Cin; p.dys = (dy_stride && dy_stride[j] > 0) ? dy_stride[j] : Cout;
        MPN_REQUIRE(p.xs >= Cin && p.dys >= Cout && p.xs % 8 == 0 && p.dys % 8 == 0, MPN_ERR_BAD_SHAPE, "conv wgrad grouped: bad pixel strides");
        MPN_REQUIRE((long long)N * H[j] * W[j] * p.xs < (1ll << 31) && (long long)N * H[j] * W[j] * p.dys < (1ll << 31), MPN_ERR_BAD_SHAPE,
                    "conv wgrad grouped: tensors must span fewer than 2^31 elements");
        MPN_REQUIRE(ksize != 1 || ((long long)N * H[j] * W[j] * p.xs < (1ll << 30) && (long long)N * H[j] * W[j] * p.dys < (1ll << 30)),
                    MPN_ERR_BAD_SHAPE, "conv wgrad grouped: 1x1 tensors must span fewer than 2^31 bytes");
        p.tiles_x = (W[j] + 15) / 16; p.tiles_y = (H[j] + 7) / 8;
        p.M = (long long)N * H[j] * W[j];
        p.ntiles = geoms[j].ntiles; p.nsplit = nsplit[j]; p.n_cg = geoms[j].n_cg; p.n_cb = geoms[j].n_cb;
        const int blocks = p.n_cg * p.n_cb * p.nsplit;
        p.xcd_remap = (begin % 8 == 0 && blocks % 8 == 0) ? 1 : 0;
#ifdef MPN_DIAG
        p.dbg = nullptr;
#endif
        grp.begin[j] = begin;
        begin += blocks;
    }
    for (int j = njobs; j <= kMaxWgradGroup; ++j) grp.begin[j] = begin;
    grp.njobs = njobs;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPN_F16) {
        if (wgrad_thin1x1(Cin, Cout, ksize)) return launch_wgrad_bf16_grouped<half_t, 1, 64, 128, 2, false>(grp, begin, st);
        if (ksize == 1) return launch_wgrad_bf16_grouped<half_t, 1, 256, 256, 4, false>(grp, begin, st);
        if (Cout <= 64) return launch_wgrad_bf16_grouped<half_t, 9, 256, 128, 4, true>(grp, begin, st);
        return launch_wgrad_bf16_grouped<half_t, 9, 128, 256, 4, true>(grp, begin, st);
    }
    if (wgrad_thin1x1(Cin, Cout, ksize)) return launch_wgrad_bf16_grouped<bf16_t, 1, 64, 128, 2, false>(grp, begin, st);
    if (ksize == 1) return launch_wgrad_bf16_grouped<bf16_t, 1, 256, 256, 4, false>(grp, begin, st);
    if (Cout <= 64) return launch_wgrad_bf16_grouped<bf16_t, 9, 256, 128, 4, true>(grp, begin, st);
    return launch_wgrad_bf16_grouped<bf16_t, 9, 128, 256, 4, true>(grp, begin, st);
}

namespace {
template <typename T, int RBA, int RBD, int WM, bool DAPPLY = false>
int launch_conv1x1_bwd_fused(const WgradParams& p, hipStream_t st) {
    constexpr int CG = RBA / 2;
    constexpr int smem = 128 * (RBA + 32) + 128 * (RBD + 32) + 2 * CG * (int)sizeof(float) + 128 * (RBA + 32) + 128 * (RBA + 16) +
                         8 * 2 * CG * (int)sizeof(float) + (DAPPLY ? 4 * (RBD / 2) * (int)sizeof(float) : 0);
    static_assert(smem <= 160 * 1024, "LDS budget");
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)conv1x1_bwd_fused_kernel<T, RBA, RBD, WM, DAPPLY>, smem, &attr_mask));
    conv1x1_bwd_fused_kernel<T, RBA, RBD, WM, DAPPLY><<<dim3((unsigned)p.nsplit), dim3(512), smem, st>>>(p);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
}  // namespace

/* 1 when mpn_conv1x1_bwd_fused takes this layer: bf16 storage (the batch-norm kernels that finish the reduction take f32 / bf16), a thin pointwise layer whose channels fit ONE block tile
 * (Cin <= 128, Cout <= 128: Conv2d_1..3_pointwise of mobilenet_v1.py:66-74 at depth_multiplier 1, the FPN's lateral2) */
extern "C" int mpn_conv1x1_bwd_fused_supported(int Cin, int Cout, int dtype) {
    return (dtype == MPN_BF16 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0 && Cin <= 128 && Cout <= 128) ? 1 : 0;
}

/* A thin 1x1 convolution's backward in ONE pass over x and dy: wpart [mpn_conv_wgrad_num_parts(.., 1, ..)][Cin][Cout] = weight-gradient
 * partials over act(x * in_scale + in_shift) (finish with mpn_reduce_partials), dx [N,H,W,Cin] = dy . w^T MASKED by that activation
 * (lo < x * in_scale + in_shift < hi on the raw x) and bn_part [same rows][2][Cin] = partial sums of the masked gradient g and of
 * g * x with the RAW x (finish with mpn_bn_bwd_finalize_raw) - what mpn_conv_bwd_weight + mpn_conv_bwd_data_bn produce in two passes
 * over both tensors. w: the layer's f32 kernel [Cin][Cout] (HWIO of a 1x1), rounded to the storage type as the packed weights are. */
namespace {
struct Conv1x1Apply { const void* y_raw; int y_stride; const float *scale, *shift, *mean, *invstd, *k1, *k2; int act; };
int conv1x1_bwd_fused_impl(const void* x, const void* dy, const float* w, void* dx, float* wpart, float* bn_part, int N, int H,
                           int W, int Cin, int Cout, int x_stride, int dy_stride, int dx_stride, int dtype,
                           const float* in_scale, const float* in_shift, int in_act, const Conv1x1Apply* ap, mpn_stream_t stream) {
    MPN_REQUIRE(mpn_conv1x1_bwd_fused_supported(Cin, Cout, dtype), MPN_ERR_BAD_SHAPE, "conv1x1_bwd_fused: layer not covered (Cin %d, Cout %d)", Cin, Cout);
    MPN_REQUIRE(x && dy && w && dx && wpart && in_scale && in_shift && N > 0 && H > 0 && W > 0, MPN_ERR_BAD_ARG, "conv1x1_bwd_fused: bad arguments");
    MPN_REQUIRE(bn_part != nullptr || ap == nullptr, MPN_ERR_BAD_ARG, "conv1x1_bwd_fused_apply: needs the reduction's slab");
    MPN_REQUIRE(mpn_aligned16(x) && mpn_aligned16(dy) && mpn_aligned16(dx), MPN_ERR_BAD_ALIGN, "conv1x1_bwd_fused: pointers must be 16-byte aligned");
    MPN_REQUIRE(dx != x && dx != dy, MPN_ERR_BAD_ARG, "conv1x1_bwd_fused: dx must not alias an input");
    WgradParams p = {};
    p.x = x; p.dy = dy; p.part = wpart; p.in_scale = in_scale; p.in_shift = in_shift; p.in_act = in_act;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.xs = x_stride > 0 ? x_stride : Cin; p.dys = dy_stride > 0 ? dy_stride : Cout; p.dxs = dx_stride > 0 ? dx_stride : Cin;
    MPN_REQUIRE(p.xs >= Cin && p.dys >= Cout && p.dxs >= Cin && p.xs % 8 == 0 && p.dys % 8 == 0 && p.dxs % 8 == 0, MPN_ERR_BAD_SHAPE,
                "conv1x1_bwd_fused: pixel strides must be multiples of 8 not below the channel counts");
    MPN_REQUIRE((long long)N * H * W * p.xs < (1ll << 31) && (long long)N * H * W * p.dys < (1ll << 31), MPN_ERR_BAD_SHAPE,
                "conv1x1_bwd_fused: tensors must span fewer than 2^31 elements");
    MPN_REQUIRE((long long)N * H * W * p.xs < (1ll << 30) && (long long)N * H * W * p.dys < (1ll << 30), MPN_ERR_BAD_SHAPE,
                "conv1x1_bwd_fused: tensors must span fewer than 2^31 bytes");
    const WgradGeom g = wgrad_geom(N, H, W, Cin, Cout, 1, 2);
    MPN_REQUIRE(g.n_cg == 1 && g.n_cb == 1, MPN_ERR_BAD_SHAPE, "conv1x1_bwd_fused: one block tile must hold the layer");
    p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 7) / 8;
    p.M = (long long)N * H * W;
    p.ntiles = g.ntiles; p.nsplit = g.nsplit; p.n_cg = 1; p.n_cb = 1; p.xcd_remap = 1;
    p.wf = w; p.dx = dx; p.bn_part = bn_part;
#ifdef MPN_DIAG
    p.dbg = nullptr;
#endif
    hipStream_t st = (hipStream_t)stream;
    const bool thin32 = Cin <= 32 && Cout <= 64, wide = Cin > 64;
    if (ap != nullptr) {
        MPN_REQUIRE(!wide, MPN_ERR_BAD_SHAPE, "conv1x1_bwd_fused_apply: layer not covered (Cin %d, Cout %d)", Cin, Cout);
        MPN_REQUIRE(ap->y_raw && ap->scale && ap->shift && ap->mean && ap->invstd && ap->k1 && ap->k2 && mpn_aligned16(ap->y_raw), MPN_ERR_BAD_ARG,
                    "conv1x1_bwd_fused_apply: bad batch-norm arguments");
        p.ap_x = ap->y_raw; p.ap_xs = ap->y_stride > 0 ? ap->y_stride : Cout;
        MPN_REQUIRE(p.ap_xs >= Cout && p.ap_xs % 8 == 0 && (long long)N * H * W * p.ap_xs < (1ll << 31), MPN_ERR_BAD_SHAPE, "conv1x1_bwd_fused_apply: bad raw-output stride");
        p.ap_scale = ap->scale; p.ap_shift = ap->shift; p.ap_mean = ap->mean; p.ap_invstd = ap->invstd; p.ap_k1 = ap->k1; p.ap_k2 = ap->k2; p.ap_act = ap->act;
        return thin32 ? launch_conv1x1_bwd_fused<bf16_t, 64, 128, 2, true>(p, st) : launch_conv1x1_bwd_fused<bf16_t, 128, 256, 2, true>(p, st);
    }
    if (wide) return launch_conv1x1_bwd_fused<bf16_t, 256, 256, 4>(p, st);
    return thin32 ? launch_conv1x1_bwd_fused<bf16_t, 64, 128, 2>(p, st) : launch_conv1x1_bwd_fused<bf16_t, 128, 256, 2>(p, st);
}
}  // namespace

extern "C" int mpn_conv1x1_bwd_fused(const void* x, const void* dy, const float* w, void* dx, float* wpart, float* bn_part, int N, int H,
                                     int W, int Cin, int Cout, int x_stride, int dy_stride, int dx_stride, int dtype,
                                     const float* in_scale, const float* in_shift, int in_act, mpn_stream_t stream) {
    return conv1x1_bwd_fused_impl(x, dy, w, dx, wpart, bn_part, N, H, W, Cin, Cout, x_stride, dy_stride, dx_stride, dtype, in_scale, in_shift,
                                  in_act, nullptr, stream);
}

/* 1 when mpn_conv1x1_bwd_fused_apply takes this layer (Cin <= 64, Cout <= 128, bf16) */
extern "C" int mpn_conv1x1_bwd_fused_apply_supported(int Cin, int Cout, int dtype) {
    // (the 128-channel tile with the apply pass folded in spills 31 registers and is slower than the separate pass: 234 vs 179 us)
    return (mpn_conv1x1_bwd_fused_supported(Cin, Cout, dtype) && Cin <= 64) ? 1 : 0;
}

/* mpn_conv1x1_bwd_fused with the batch-norm backward APPLY pass of the layer's OWN batch-norm folded into the staging of dY: g = the
 * gradient w.r.t. the layer's activated output (what mpn_bn_bwd_apply would turn into dy in place), y_raw = the layer's raw output,
 * ap_* that batch-norm's affine, saved statistics and the k1 / k2 of mpn_bn_bwd_finalize. The slabs and dx of mpn_bn_bwd_apply
 * followed by mpn_conv1x1_bwd_fused (to the storage rounding of a rare staged element); g and y_raw are not written (two passes over the
 * layer's output tensor less). */
extern "C" int mpn_conv1x1_bwd_fused_apply(const void* x, const void* g, const void* y_raw, const float* w, void* dx, float* wpart,
                                           float* bn_part, int N, int H, int W, int Cin, int Cout, int x_stride, int g_stride, int y_stride,
                                           int dx_stride, int dtype, const float* in_scale, const float* in_shift, int in_act,
                                           const float* ap_scale, const float* ap_shift, const float* ap_mean, const float* ap_invstd,
                                           const float* ap_k1, const float* ap_k2, int ap_act, mpn_stream_t stream) {
    const Conv1x1Apply ap = {y_raw, y_stride, ap_scale, ap_shift, ap_mean, ap_invstd, ap_k1, ap_k2, ap_act};
    return conv1x1_bwd_fused_impl(x, g, w, dx, wpart, bn_part, N, H, W, Cin, Cout, x_stride, g_stride, dx_stride, dtype, in_scale, in_shift,
                                  in_act, &ap, stream);
}
