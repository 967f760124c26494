// K5/K6 (deep layers): pointwise 1x1 convolution as a plain GEMM on the MFMA matrix cores, bf16, NHWC.
// Replaces slim.conv2d 1x1 + the consumer side of its batch-norm (detector/backbones/mobilenet_v1.py:66-74) for the layers
// where the contraction is deep enough to be matrix work (K >= 256, N a multiple of 256: Conv2d_5..13_pointwise forward and
// the data gradients of Conv2d_6..13_pointwise) - the thin layers stay on conv_mfma.hip, where they are HBM-bound anyway.
//
//   y[m, n] = sum_k act(x[m, k] * scale[k] + shift[k]) * W[k, n]          m = pixel (N*H*W), k = Cin, n = Cout
//
// Why a second kernel: conv_mfma's 128 px x 128 ch tiles make every block re-read its A rows once per n-tile and the
// whole weight slice once per pixel tile - 300 MB of L2 -> CU traffic per launch on the 512 -> 512 layer against 67 MB of
// tensors, 18 % of the MFMA peak. Here ONE 8-wave block per CU owns BM x 256 outputs (BM = 256, or 128 when that is what
// it takes to give all 256 CUs a tile): A rows are read N/256 times, weights M/BM times (~100 MB on that layer).
//
// Structure per 64-channel k-step (128 bytes per row), two LDS buffers:
//   barrier | issue next step: weights by LDS-DMA (global_load_lds_dwordx4, no registers), A rows by global_load into
//   registers (batch-norm affine + ReLU/ReLU6 must be applied on the way in) - or by LDS-DMA too when there is no affine
//   (data gradients) | 2 x (fragment reads, MT x 4 MFMAs 16x16x32) on the current buffers | wait, transform and
//   ds_write the A registers into the other buffer.
// LDS images are LINEAR rows of 128 bytes (LDS-DMA writes wave-uniform base + lane * 16); the 16-byte slot s of row r holds
// k-slot s ^ ((r >> 1) & 7): applied on the SOURCE address when staging and on the read address of the fragments, which
// makes every ds_read_b128 lane group hit 16 distinct bank slots (MI355X_MICROARCH.md, LDS table).
// Weights come as a plain [N][K] bf16 matrix (mpn_conv_pack_weights: transpose-cast of the HWIO variable for the forward,
// a plain cast for the data gradient).
// Epilogue as in conv_mfma.hip: accumulators -> bf16 LDS image of the tile -> batch-norm partial sums on the matrix unit
// (ones x F, F^T x F) -> 16-byte row stores. One statistics row per 128 pixels (the contract of mpn_conv_num_parts).
#include "common.h"
#include "pointwise.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kThreads = 512;
constexpr int BN = 256;        // output channels per block
constexpr int KB = 128;        // bytes of K per row and k-step (64 bf16)

struct PwParams {
    const bf16_t* x;
    const bf16_t* w;           // [N][K]
    bf16_t* y;
    const float* in_scale;
    const float* in_shift;
    int in_act;
    float* stats_part;
    long long M;
    int K, N, x_stride, y_stride, n_tiles, m_tiles;
    // data gradients that also reduce for the batch-norm they feed (mpn_conv_bwd_data_bn): y is written MASKED by that layer's
    // activation (computed from bnr_x * bnr_scale + bnr_shift) and stats_part receives the sums of g and g * bnr_x (raw x)
    const bf16_t* bnr_x;
    const float* bnr_scale;
    const float* bnr_shift;
    int bnr_act, bnr_xs;
#ifdef MPN_DIAG
    unsigned long long* dbg;   // diagnostic build only: 8 u64 per block (s_memtime at the phase boundaries; [6], [7]: s_memrealtime)
#endif
};

#ifdef MPN_DIAG
#define PW_STAMP(k) do { if (p.dbg && threadIdx.x == 0) { p.dbg[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    if ((k) == 0) p.dbg[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime(); \
    if ((k) == 5) p.dbg[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define PW_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ bf16x4_t o_tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4_t __attribute__((address_space(3)))*)(p));
}

// two f32 -> one dword of two bf16 (RNE, NaN stays NaN): a vector conversion so that hipcc emits ONE v_cvt_pk_bf16_f32
// (two scalar casts + shift + or cost four instructions)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}

__device__ __forceinline__ void store4_bf16(unsigned char* p, const f32x4_t& v) {
    uint2 q;
    q.x = pack_bf16x2(v[0], v[1]);
    q.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = q;
}

__device__ __forceinline__ void glds16(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// ================= epilogue shared by the kernels below: bf16 image of the tile in LDS (over the dead staging buffers:
// the caller has passed a barrier after the last fragment read), statistics on the matrix unit, 16-byte row stores
template <int BM, bool BNR>
__device__ __forceinline__ void pw_epilogue(const PwParams& p, unsigned char* smem, f32x4_t (&acc)[BM / 32][4], const int mtile,
                                            const int n0, const long long m0) {
    constexpr int MT = BM / 32, NT = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int RSO = BN * 2 + 8;
    unsigned char* O = smem;
    float* red = reinterpret_cast<float*>(smem + BM * RSO);   // BM == 128 only: [2 wm][2][BN]

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = wm * (BM / 2) + mt * 16 + l15;
        const bool ok = (m0 + row) < p.M;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4_t v = acc[mt][nt];
            if (!ok) v = (f32x4_t){0.f, 0.f, 0.f, 0.f};   // rows past the end must not count in the statistics
            store4_bf16(O + row * RSO + (wn * 64 + nt * 16 + lq * 4) * 2, v);
        }
    }
    PW_STAMP(3);
    if (!BNR && p.stats_part != nullptr) {
        f32x4_t sa[NT], ga[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { sa[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; ga[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
        bf16x8_t ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;
        // the wave reads back ITS part of the image (written by its own lanes: in order within a wave, no barrier) with the
        // transposing read as operands F[k = pixel][channel]: lane 4q+pp of a 16-lane group supplies block row q, channels 4pp..
        const unsigned char* tb = O + (wm * (BM / 2) + (lq >> 1) * 2 + 4 * ((lq & 1) * 4 + (l15 >> 2))) * RSO +
                                  (wn * 64 + 4 * (l15 & 3)) * 2;
#pragma unroll
        for (int kb = 0; kb < MT / 2; ++kb)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x4_t lo = o_tr_read(tb + kb * 32 * RSO + nt * 32);
                const bf16x4_t hi = o_tr_read(tb + (kb * 32 + 1) * RSO + nt * 32);
                const bf16x8_t f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                sa[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f, sa[nt], 0, 0, 0);
                ga[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, ga[nt], 0, 0, 0);
            }
        const int r = l15 & 3;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float q = r == 0 ? ga[nt][0] : (r == 1 ? ga[nt][1] : (r == 2 ? ga[nt][2] : ga[nt][3]));
            if (lq == (l15 >> 2)) {
                const int cl = wn * 64 + nt * 16 + l15;
                if (BM == 256) {   // wave row wm = one 128-pixel statistics row of its own
                    const long long srow = (long long)mtile * 2 + wm;
                    if (srow * 128 < p.M) {
                        p.stats_part[(srow * 2 + 0) * p.N + n0 + cl] = sa[nt][0];
                        p.stats_part[(srow * 2 + 1) * p.N + n0 + cl] = q;
                    }
                } else {
                    red[(wm * 2 + 0) * BN + cl] = sa[nt][0];
                    red[(wm * 2 + 1) * BN + cl] = q;
                }
            }
        }
    }
    PW_STAMP(4);
    // copy-out, wave by wave: each wave stores the part of the image it wrote itself (BM/2 pixels x 64 channels = 128
    // contiguous bytes per pixel row), so no block barrier separates the waves that finish early from the stores
    {
        const unsigned char* Ow = O + (wm * (BM / 2)) * RSO + wn * 128;
        bf16_t* yw = p.y + n0 + wn * 64;
        const int prow = lane >> 3, piece = lane & 7;
        f32x2_t bsc[4], bsh[4], bs[4], bq[4];
        float blo = -INFINITY, bhi = INFINITY;
        if constexpr (BNR) {
            const float* sc = p.bnr_scale + n0 + wn * 64 + piece * 8;
            const float* sh = p.bnr_shift + n0 + wn * 64 + piece * 8;
            const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(sc), s1 = *reinterpret_cast<const f32x4_t*>(sc + 4);
            const f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(sh), h1 = *reinterpret_cast<const f32x4_t*>(sh + 4);
            bsc[0] = (f32x2_t){s0[0], s0[1]}; bsc[1] = (f32x2_t){s0[2], s0[3]}; bsc[2] = (f32x2_t){s1[0], s1[1]}; bsc[3] = (f32x2_t){s1[2], s1[3]};
            bsh[0] = (f32x2_t){h0[0], h0[1]}; bsh[1] = (f32x2_t){h0[2], h0[3]}; bsh[2] = (f32x2_t){h1[0], h1[1]}; bsh[3] = (f32x2_t){h1[2], h1[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) { bs[j] = (f32x2_t){0.f, 0.f}; bq[j] = (f32x2_t){0.f, 0.f}; }
            blo = p.bnr_act != MPN_ACT_NONE ? 0.f : -INFINITY;
            bhi = p.bnr_act == MPN_ACT_RELU6 ? 6.f : INFINITY;
        }
        // BNR: the fed layer's raw tensor at this wave's pixels x 64 channels in the copy-out layout (lane = 16-byte piece lane % 8
        // of rows lane / 8 + 8 i), eight rows (32 registers) at a time behind a fence - all BM / 16 at once, or hoisted above
        // the image stores next to the live accumulators, they spill
        uint4 bx[BNR ? 8 : 1];
        // (BNR: a REAL loop over groups of eight rows - fully unrolled, hipcc hoists every group's LDS reads to the top and
        //  spills the running sums)
#pragma unroll 1
        for (int i0 = 0; i0 < (BNR ? BM / 16 : 1); i0 += 8) {
        if constexpr (BNR) {
            const bf16_t* xw = p.bnr_x + n0 + wn * 64 + piece * 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                long long px = m0 + wm * (BM / 2) + (i0 + u) * 8 + prow;
                if (px >= p.M) px = p.M - 1;       // (rows past the end hold dy = 0)
                bx[u] = *reinterpret_cast<const uint4*>(xw + px * p.bnr_xs);
            }
        }
#pragma unroll
        for (int iu = 0; iu < (BNR ? 8 : BM / 16); ++iu) {
            const int i = i0 + iu;
            const int row = i * 8 + prow;
            const long long pixel = m0 + wm * (BM / 2) + row;
            if (BNR || pixel < p.M) {
                uint2 a = *reinterpret_cast<const uint2*>(Ow + row * RSO + piece * 16);
                uint2 b = *reinterpret_cast<const uint2*>(Ow + row * RSO + piece * 16 + 8);
                if constexpr (BNR) {
                    // g = dy where the fed batch-norm's activation passes (the test of bn_bwd_reduce / bn_bwd_apply), else 0;
                    // sums of g and g * x (rows past the end hold dy = 0)
                    const unsigned xu[4] = {bx[iu & 7].x, bx[iu & 7].y, bx[iu & 7].z, bx[iu & 7].w};
                    unsigned du[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2_t xf = {__uint_as_float(xu[j] << 16), __uint_as_float(xu[j] & 0xffff0000u)};
                        const f32x2_t pre = xf * bsc[j] + bsh[j];
                        const unsigned m = ((pre[0] > blo && pre[0] < bhi) ? 0x0000ffffu : 0u) | ((pre[1] > blo && pre[1] < bhi) ? 0xffff0000u : 0u);
                        du[j] &= m;
                        const f32x2_t gf = {__uint_as_float(du[j] << 16), __uint_as_float(du[j] & 0xffff0000u)};
                        bs[j] += gf;
                        bq[j] += gf * xf;
                    }
                    a = make_uint2(du[0], du[1]); b = make_uint2(du[2], du[3]);
                }
                if (pixel < p.M) {
// non-temporal: 33 MB of plain stores stay dirty in the L2s and are written back when the kernel ends, in front
                // of the next launch; streamed out they leave during the copy-out (2-10 % per launch, measured)
                typedef unsigned u32x4n_t __attribute__((ext_vector_type(4)));
                const u32x4n_t v4 = {a.x, a.y, b.x, b.y};
                __builtin_nontemporal_store(v4, reinterpret_cast<u32x4n_t*>(yw + pixel * p.y_stride + piece * 8));
                }
            }
        }
        }
        if constexpr (BNR) {
            // the 8 row lanes of a piece (lane bits 3..5), fixed butterfly; lanes 0..7 then hold the wave's sums of 8 channels each
#pragma unroll
            for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bs[j][0] += __shfl_xor(bs[j][0], o, 64); bs[j][1] += __shfl_xor(bs[j][1], o, 64);
                    bq[j][0] += __shfl_xor(bq[j][0], o, 64); bq[j][1] += __shfl_xor(bq[j][1], o, 64);
                }
            if (lane < 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cl = wn * 64 + lane * 8 + 2 * j;
                    if (BM == 256) {   // wave row wm = one 128-pixel statistics row of its own
                        const long long srow = (long long)mtile * 2 + wm;
                        if (srow * 128 < p.M) {
                            p.stats_part[(srow * 2 + 0) * p.N + n0 + cl] = bs[j][0]; p.stats_part[(srow * 2 + 0) * p.N + n0 + cl + 1] = bs[j][1];
                            p.stats_part[(srow * 2 + 1) * p.N + n0 + cl] = bq[j][0]; p.stats_part[(srow * 2 + 1) * p.N + n0 + cl + 1] = bq[j][1];
                        }
                    } else {
                        red[(wm * 2 + 0) * BN + cl] = bs[j][0]; red[(wm * 2 + 0) * BN + cl + 1] = bs[j][1];
                        red[(wm * 2 + 1) * BN + cl] = bq[j][0]; red[(wm * 2 + 1) * BN + cl + 1] = bq[j][1];
                    }
                }
            }
        }
    }
    if (BM == 128) __syncthreads();   // `red` complete
    if (BM == 128 && p.stats_part != nullptr && tid < 2 * BN) {
        const int which = tid / BN, c = tid % BN;
        p.stats_part[((long long)mtile * 2 + which) * p.N + n0 + c] = red[which * BN + c] + red[(2 + which) * BN + c];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PW_STAMP(5);
}

// BM: pixels per block (256 or 128); AFFINE: A rows through registers with batch-norm affine + activation, else LDS-DMA
// (a v_mfma_f32_32x32x16_bf16 variant of the same tiles was 3-8 % slower on every layer: profiles/r05_pointwise_m32.txt; removed in round 6)
template <int BM, bool AFFINE, bool BNR = false>
__global__ __launch_bounds__(kThreads, 2) void pw_gemm_kernel(const PwParams p) {
    static_assert(!BNR || !AFFINE, "the fused batch-norm backward reduction rides on a data gradient (no producer affine)");
    constexpr int MT = BM / 32;                  // 16-pixel m-tiles per wave (waves: 2 along M x 4 along N)
    constexpr int NT = 4;                        // 16-channel n-tiles per wave
    constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB;
    constexpr int AV = BM * 8 / kThreads;        // 16-byte A vectors per thread and k-step (4 or 2)
    constexpr int APIECES = BM / 8 / 8;          // LDS-DMA pieces (8 rows) per wave for A
    constexpr int BPIECES = BN / 8 / 8;          // ... for the weights: 4

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                    // [2][BM][128]
    unsigned char* Bs = smem + 2 * A_BYTES;      // [2][256][128]
    float* tab = reinterpret_cast<float*>(smem + 2 * A_BYTES + 2 * B_BYTES);   // [2][K] scale, shift (AFFINE)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;

    // XCD-aware block -> tile map (the n-tiles of a pixel tile re-read the same A rows: keep them on one L2)
    int wid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wid & 7;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wid >> 3);
    }
    const int ntile = wid % p.n_tiles;
    const int mtile = wid / p.n_tiles;
    const int n0 = ntile * BN;
    const long long m0 = (long long)mtile * BM;
    const int ksteps = p.K >> 6;

    if (AFFINE) {
        for (int i = tid; i < p.K; i += kThreads) { tab[i] = p.in_scale[i]; tab[p.K + i] = p.in_shift[i]; }
    }
    const float act_lo = (p.in_act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float act_hi = (p.in_act == MPN_ACT_RELU6) ? 6.f : INFINITY;

    // ---- staging addresses
    // LDS-DMA: piece = 8 rows x 128 bytes = one wave instruction; lane j fills (row piece*8 + j/8, slot j%8) and must fetch
    // k-slot (j%8) ^ ((row >> 1) & 7) of that row. Per-lane source bases are fixed for the whole tile (the k-step adds a
    // wave-uniform offset).
    const int prow = lane >> 3, pslot = lane & 7;
    const bf16_t* bsrc[BPIECES];
#pragma unroll
    for (int i = 0; i < BPIECES; ++i) {
        const int row = (wave * BPIECES + i) * 8 + prow;
        bsrc[i] = p.w + (long long)(n0 + row) * p.K + (pslot ^ ((row >> 1) & 7)) * 8;
    }
    auto b_issue = [&](int ks, int buf, int i) {
        glds16(bsrc[i] + ks * 64, Bs + buf * B_BYTES + (wave * BPIECES + i) * 1024);   // (+ lane * 16 by the hardware)
    };
    const bf16_t* asrc[AFFINE ? AV : APIECES];
    if (!AFFINE) {
#pragma unroll
        for (int i = 0; i < APIECES; ++i) {
            const int row = (wave * APIECES + i) * 8 + prow;
            long long m = m0 + row;
            if (m >= p.M) m = p.M - 1;   // rows past the end: any valid address (their outputs are zeroed and never stored)
            asrc[i] = p.x + m * p.x_stride + (pslot ^ ((row >> 1) & 7)) * 8;
        }
    }
    auto a_issue_dma = [&](int ks, int buf, int i) {
        glds16(asrc[i] + ks * 64, As + buf * A_BYTES + (wave * APIECES + i) * 1024);
    };
    // register path: thread -> (row = tid/8 + 64 i, k-slot tid%8); the same k-slot for all its rows, so one pair of
    // scale / shift vectors per k-step
    const int aslot = tid & 7, arow0 = tid >> 3;
    if (AFFINE) {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            long long m = m0 + arow0 + i * 64;
            if (m >= p.M) m = p.M - 1;
            asrc[i] = p.x + m * p.x_stride + aslot * 8;
        }
    }
    uint4 areg[AV];
    auto a_load = [&](int ks) {
#pragma unroll
        for (int i = 0; i < AV; ++i) areg[i] = *reinterpret_cast<const uint4*>(asrc[i] + ks * 64);
    };
    float scv[8], shv[8];
    auto a_table = [&](int ks) {
        const float* sc = tab + ks * 64 + aslot * 8;
        const float* sh = sc + p.K;
        const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(sc), s1 = *reinterpret_cast<const f32x4_t*>(sc + 4);
        const f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(sh), h1 = *reinterpret_cast<const f32x4_t*>(sh + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { scv[j] = s0[j]; scv[4 + j] = s1[j]; shv[j] = h0[j]; shv[4 + j] = h1[j]; }
    };
    auto a_commit = [&](int buf, int i) {
        const int row = arow0 + i * 64;
        const unsigned u[4] = {areg[i].x, areg[i].y, areg[i].z, areg[i].w};
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float lo = __uint_as_float(u[j] << 16), hi = __uint_as_float(u[j] & 0xffff0000u);
            lo = __builtin_amdgcn_fmed3f(lo * scv[2 * j] + shv[2 * j], act_lo, act_hi);
            hi = __builtin_amdgcn_fmed3f(hi * scv[2 * j + 1] + shv[2 * j + 1], act_lo, act_hi);
            o[j] = pack_bf16x2(lo, hi);
        }
        // (rows past the end hold a clamped row's data: their outputs are zeroed in the epilogue and never stored)
        const uint4 q = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(As + buf * A_BYTES + row * KB + ((aslot ^ ((row >> 1) & 7)) << 4)) = q;
    };

    // ---- fragment addresses: row = (wave offset + tile * 16 + l15): ((row >> 1) & 7) == l15 >> 1 for every tile, so the
    // swizzled slot is a per-lane constant; the second 32-element half of the k-step flips bit 2 of the slot (^ 64 bytes)
    const int g = l15 >> 1;
    const int foff0 = ((lq ^ (g & 3)) | (g & 4)) << 4;
    const int foff1 = foff0 ^ 64;
    const int a_lane = (wm * (BM / 2) + l15) * KB;
    const int b_lane = (wn * 64 + l15) * KB;

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stage k-step 0
    PW_STAMP(0);
#pragma unroll
    for (int i = 0; i < BPIECES; ++i) b_issue(0, 0, i);
    if (AFFINE) {
        a_load(0);
        __syncthreads();                 // scale / shift table visible
        a_table(0);
#pragma unroll
        for (int i = 0; i < AV; ++i) a_commit(0, i);
    } else {
#pragma unroll
        for (int i = 0; i < APIECES; ++i) a_issue_dma(0, 0, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PW_STAMP(1);

    // ---- main loop. A k-step is four quadrants q = (k half h, m half): [staging work of the NEXT k-step | fragment reads of
    // quadrant q+1 | MT/2 x 4 MFMAs of quadrant q]. The LDS-DMA pieces / global loads are spread over the quadrants - a wave
    // that issues its 8 pieces back to back stalls ~100 cycles on each while BOTH waves of the SIMD leave the matrix pipe idle
    // (they run in lockstep behind the barrier) - and the affine + activation + ds_write of the A registers sits in the last
    // two quadrants, where its VALU instructions fall into the issue gaps between the MFMAs.
    constexpr int MH = MT / 2;
    bf16x8_t bP[NT], bQ[NT], aP[MH], aQ[MH];
    auto read_b = [&](bf16x8_t (&b)[NT], const unsigned char* Bb, int fo) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const bf16x8_t*>(Bb + nt * 16 * KB + fo);
    };
    auto read_a = [&](bf16x8_t (&a)[MH], const unsigned char* Ab, int fo, int mh) {
#pragma unroll
        for (int mt = 0; mt < MH; ++mt) a[mt] = *reinterpret_cast<const bf16x8_t*>(Ab + (mh * MH + mt) * 16 * KB + fo);
    };
    auto mma = [&](const bf16x8_t (&a)[MH], const bf16x8_t (&b)[NT], int mh) {
#pragma unroll
        for (int mt = 0; mt < MH; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mh * MH + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt], a[mt], acc[mh * MH + mt][nt], 0, 0, 0);   // D^T = W^T x A^T
    };
    // staging work of quadrant q for the next k-step: requests in the first two quadrants (they need the rest of the k-step
    // to land), the A commit (AFFINE) in the last two
    auto stage = [&](int ks1, int nbuf, int q) {
        if (AFFINE) {
            if (q == 0) {
#pragma unroll
                for (int i = 0; i < BPIECES / 2; ++i) b_issue(ks1, nbuf, i);
                a_load(ks1);
#pragma unroll
                for (int i = BPIECES / 2; i < BPIECES; ++i) b_issue(ks1, nbuf, i);
            } else if (q == 1) {
            } else if (q == 2) {
                a_table(ks1);
#pragma unroll
                for (int i = 0; i < AV / 2; ++i) a_commit(nbuf, i);
            } else {
#pragma unroll
                for (int i = AV / 2; i < AV; ++i) a_commit(nbuf, i);
            }
        } else {
            constexpr int TOT = APIECES + BPIECES;          // 8 or 6 pieces per wave
            constexpr int HALF = (TOT + 1) / 2;
            if (q < 2) {
#pragma unroll
                for (int j = q * HALF; j < (q + 1) * HALF && j < TOT; ++j) {
                    if (j < BPIECES) b_issue(ks1, nbuf, j); else a_issue_dma(ks1, nbuf, j - BPIECES);
                }
            }
        }
    };

    auto kstep = [&](int ks, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        const int buf = ks & 1;
        __syncthreads();   // buffers `buf` complete (every wave waited for its pieces / wrote its rows); `buf ^ 1` free
        const unsigned char* Ab = As + buf * A_BYTES + a_lane;
        const unsigned char* Bb = Bs + buf * B_BYTES + b_lane;
        read_b(bP, Bb, foff0);
        read_a(aP, Ab, foff0, 0);
        // q0
        if constexpr (more) stage(ks + 1, buf ^ 1, 0);
        read_a(aQ, Ab, foff0, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(aP, bP, 0);
        // q1
        if constexpr (more) stage(ks + 1, buf ^ 1, 1);
        read_b(bQ, Bb, foff1);
        read_a(aP, Ab, foff1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(aQ, bP, 1);
        // q2
        read_a(aQ, Ab, foff1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) stage(ks + 1, buf ^ 1, 2);
        mma(aP, bQ, 0);
        if constexpr (AFFINE && more) {   // the commit's VALU instructions into the issue gaps between the MFMAs
#pragma unroll
            for (int i = 0; i < MH * NT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, AV, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // q3
        if constexpr (more) stage(ks + 1, buf ^ 1, 3);
        mma(aQ, bQ, 1);
        if constexpr (AFFINE && more) {
#pragma unroll
            for (int i = 0; i < MH * NT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, AV, 0);
            }
        }
        // (the __syncthreads() at the top of the next k-step waits for this wave's pieces: vmcnt(0) + barrier)
    };
    // the last k-step has nothing to stage: peeled, so that the staging code of the others is branch-free (one basic block
    // with the MFMAs: the scheduler can interleave them)
    for (int ks = 0; ks + 1 < ksteps; ++ks) kstep(ks, std::true_type{});
    kstep(ksteps - 1, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PW_STAMP(2);
    __syncthreads();   // every wave is done with the staging buffers: the output image may overwrite them

    pw_epilogue<BM, BNR>(p, smem, acc, mtile, n0, m0);
}

template <int BM, bool AFFINE, bool BNR = false>
int launch_pw(const PwParams& p, hipStream_t st) {
    const int smem = 2 * BM * KB + 2 * BN * KB + (AFFINE ? 2 * p.K * (int)sizeof(float) : 0);
    const int need = BM * (BN * 2 + 8) + (BM == 128 ? 4 * BN * (int)sizeof(float) : 0);   // the epilogue's image (+ red)
    const int bytes = smem > need ? smem : need;
    MPN_REQUIRE(bytes <= 160 * 1024, MPN_ERR_BAD_SHAPE, "pointwise: K = %d needs %d bytes of LDS", p.K, bytes);
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)pw_gemm_kernel<BM, AFFINE, BNR>, 160 * 1024, &attr_mask));
    pw_gemm_kernel<BM, AFFINE, BNR><<<dim3((unsigned)(p.m_tiles * p.n_tiles)), dim3(kThreads), bytes, st>>>(p);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

#ifdef MPN_DIAG
void* g_pw_dbg = nullptr;
#endif

}  // namespace

#ifdef MPN_DIAG
extern "C" void mpn_diag_set_pw_stamps(void* buf) { g_pw_dbg = buf; }
#endif

bool pw_gemm_eligible(int K, int N, int taps, int es) {
    return taps == 1 && es == 2 && K >= 256 && K % 64 == 0 && K <= 2048 && N % BN == 0;
}

int pw_gemm_launch(const void* x, const void* w_nk, void* y, long long M, int K, int N, int x_stride, int y_stride,
                   const float* in_scale, const float* in_shift, int in_act, float* stats_part, hipStream_t st,
                   const void* bnr_x, int bnr_xs, const float* bnr_scale, const float* bnr_shift, int bnr_act) {
    PwParams p;
    p.bnr_x = (const bf16_t*)bnr_x; p.bnr_scale = bnr_scale; p.bnr_shift = bnr_shift; p.bnr_act = bnr_act; p.bnr_xs = bnr_xs;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_nk; p.y = (bf16_t*)y;
    p.in_scale = in_scale; p.in_shift = in_shift; p.in_act = in_act; p.stats_part = stats_part;
    p.M = M; p.K = K; p.N = N; p.x_stride = x_stride; p.y_stride = y_stride;
    p.n_tiles = N / BN;
#ifdef MPN_DIAG
    p.dbg = (unsigned long long*)g_pw_dbg;
#endif
    // 256-pixel tiles when they still give every CU a block, else 128-pixel tiles (the 16x16 maps: 8192 pixels)
    const long long t256 = (M + 255) / 256;
    const bool big = t256 * p.n_tiles >= 256;
    p.m_tiles = (int)(big ? t256 : (M + 127) / 128);
    const bool affine = in_scale != nullptr;
    if (bnr_x != nullptr) {
        MPN_REQUIRE(!affine && stats_part && bnr_scale && bnr_shift && bnr_xs >= N && bnr_xs % 8 == 0, MPN_ERR_BAD_ARG,
                    "pointwise: the fused batch-norm reduction needs a data gradient (no producer affine), a partial slab and the layer's affine");
        return big ? launch_pw<256, false, true>(p, st) : launch_pw<128, false, true>(p, st);
    }
    if (big) return affine ? launch_pw<256, true>(p, st) : launch_pw<256, false>(p, st);
    return affine ? launch_pw<128, true>(p, st) : launch_pw<128, false>(p, st);
}
