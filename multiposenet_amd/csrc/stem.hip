// K1+K2: input standardisation fused into the first convolution.
// Replaces `x = 2*images - 1` (detector/backbones/mobilenet_v1.py:41), the NHWC->NCHW transpose
// (:53, not needed: we stay NHWC) and `Conv2d_0` = slim.conv2d 3x3 stride 2 'SAME', 3 -> C0
// channels (:56; TF SAME on an even input pads 0 before / 1 after), plus its weight gradient.
// K = 27 is too thin for the matrix cores: VALU direct convolution out of an LDS input patch.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 16;              // 16x16 output pixels per block
constexpr int kIn = 2 * kTile + 1;     // 33x33 input patch
constexpr int kMaxC0 = 64;
constexpr int kFwdTW = 2 * kTile;      // forward: 32 x 16 output pixels per block

template <typename T, bool U8>
__device__ __forceinline__ void stage_patch(const void* images, float* patch, int img, int oy0, int ox0, int H, int W,
                                            int pad_t, int pad_l) {
    const int iy0 = oy0 * 2 - pad_t, ix0 = ox0 * 2 - pad_l;
    constexpr int NEL = kIn * kIn * 3;
    constexpr int PER = (NEL + kThreads - 1) / kThreads;   // 13 loads per thread, all issued before the first use
    float v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + k * kThreads;
        const int ch = i % 3;
        const int px = (i / 3) % kIn;
        const int py = i / (3 * kIn);
        const int iy = iy0 + py, ix = ix0 + px;
        float raw = 0.5f;  // 2*0.5-1 = 0: SAME padding is zeros of the STANDARDISED tensor
        if (i < NEL && iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const long long off = (((long long)img * H + iy) * W + ix) * 3 + ch;
            raw = U8 ? (float)reinterpret_cast<const unsigned char*>(images)[off] * (1.0f / 255.0f)
                     : reinterpret_cast<const float*>(images)[off];
        }
        v[k] = raw;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + k * kThreads;
        if (i < NEL) patch[i] = 2.0f * v[k] - 1.0f;
    }
}

// copy-out of a finished output tile [kTile x kFwdTW pixels][C0 * sizeof(T) + 16 B] from LDS: tile row r = 32 pixels x C0
// channels = one contiguous run of the output tensor; optionally the batch-norm partial sums of the rounded outputs
template <typename T>
__device__ __forceinline__ void stem_copy_out(unsigned char* otile, int orow, T* __restrict__ y, int img, int ty, int tx, int OH,
                                              int OW, int C0, float* __restrict__ stats_part) {
    constexpr int VE = Vec16<T>::N;
    const int ppr = C0 * (int)sizeof(T) / 16;            // 16-byte pieces per pixel
    // batch-norm statistics of the ROUNDED outputs (the tensor the consumer normalises), fused into the copy-out: with
    // kThreads % ppr == 0 (the host checks) a thread meets the same 16-byte piece = the same VE channels in every
    // iteration, so it keeps their sum and sum of squares in registers; one fixed-order LDS reduction per block writes
    // row blockIdx.x of the partial slab (the layout of mpn_bn_stats: finish with mpn_bn_finalize)
    float ssum[VE], ssq[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
    for (int i = threadIdx.x; i < kTile * kFwdTW * ppr; i += kThreads) {
        const int pxl = i / ppr, piece = i - pxl * ppr;
        const int r = pxl / kFwdTW, cx = pxl - r * kFwdTW;
        const int yy = ty * kTile + r, xx = tx * kFwdTW + cx;
        if (yy < OH && xx < OW) {
            Vec16<T> ov;
            *reinterpret_cast<uint4*>(&ov.raw) = *reinterpret_cast<const uint4*>(otile + pxl * orow + piece * 16);
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(y + (((long long)img * OH + yy) * OW + xx) * C0) + piece * 16) =
                *reinterpret_cast<const uint4*>(&ov.raw);
            if (stats_part != nullptr) {
                float f[VE];
                ov.unpack(f);
#pragma unroll
                for (int j = 0; j < VE; ++j) { ssum[j] += f[j]; ssq[j] += f[j] * f[j]; }
            }
        }
    }
    if (stats_part != nullptr) {   // block-uniform
        __syncthreads();           // every thread is done with the output tile: reuse it
        float* red = reinterpret_cast<float*>(otile);          // [kThreads][2 * VE]
#pragma unroll
        for (int j = 0; j < VE; ++j) { red[threadIdx.x * 2 * VE + j] = ssum[j]; red[threadIdx.x * 2 * VE + VE + j] = ssq[j]; }
        __syncthreads();
        if ((int)threadIdx.x < 2 * C0) {
            const int which = (int)threadIdx.x / C0, c = (int)threadIdx.x - which * C0;
            const int piece = c / VE, j = c - piece * VE;
            float acc = 0.f;
            for (int t = piece; t < kThreads; t += ppr) acc += red[t * 2 * VE + which * VE + j];
            stats_part[((long long)blockIdx.x * 2 + which) * C0 + c] = acc;
        }
    }
}

// forward: one thread = one output pixel x all C0 channels. Its 3x3x3 input window is three runs of 9 contiguous
// floats (36 B, dword aligned): loaded straight from global with wide loads (neighbouring lanes overlap by 1/3, served
// by L1). Weights come through uniform (scalar) loads and are multiplied with packed FP32 FMAs (v_pk_fma_f32: two channels
// per instruction, the weight pair an SGPR operand). The outputs leave through an LDS tile so that every store instruction writes whole 1 KB rows of the
// NHWC tensor: per-thread stores of 16 bytes at the 64-byte pixel stride cost as much as the rest of the kernel.
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Two horizontally adjacent output pixels per thread share every weight load. This is the f32 build's kernel; the bf16
// build runs stem_fwd_mfma_kernel below (PMC on this one at bs32 @ 512x512: 1.9 waves per SIMD on average, half of every
// wave's life in s_waitcnt, vector ALU 44 % busy - 100-114 us against 43 us of HBM time).
template <typename T, bool U8>
__global__ __launch_bounds__(kThreads) void stem_fwd_kernel(const void* __restrict__ images, const float* __restrict__ w,
                                                            T* __restrict__ y, int N, int H, int W, int C0, int OH, int OW,
                                                            int pad_t, int pad_l, int tiles_x, int tiles_y,
                                                            float* __restrict__ stats_part) {
    constexpr int VE = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char stem_smem[];
    unsigned char* otile = stem_smem;                                       // [512 px][C0 * sizeof(T) + 16]
    const int orow = C0 * (int)sizeof(T) + 16;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int img = b / tiles_y;
    const int lx = threadIdx.x % kTile, ly = threadIdx.x / kTile;
    const int oy = ty * kTile + ly, ox0 = tx * kFwdTW + lx * 2;
    if (oy < OH && ox0 < OW) {
        float in[2][27];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ox = ox0 + q;
            const int ix0 = 2 * ox - pad_l;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = 2 * oy - pad_t + ky;
                const bool row_ok = iy >= 0 && iy < H && ox < OW;
                const long long base = (((long long)img * H + (row_ok ? iy : 0)) * W + (ox < OW ? ix0 : 0)) * 3;
                if (!U8 && row_ok && ix0 >= 0 && ix0 + 2 < W) {
                    const float* src = reinterpret_cast<const float*>(images) + base;
                    const float4 a = *reinterpret_cast<const float4*>(src);   // (global dwordx4 needs dword alignment only)
                    const float4 c = *reinterpret_cast<const float4*>(src + 4);
                    const float e = src[8];
                    const float r[9] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w, e};
#pragma unroll
                    for (int j = 0; j < 9; ++j) in[q][ky * 9 + j] = 2.0f * r[j] - 1.0f;
                } else {
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const int ix = ix0 + j / 3;
                        float v = 0.f;   // SAME padding: zeros of the STANDARDISED tensor
                        if (row_ok && ix >= 0 && ix < W) {
                            const float raw = U8 ? (float)reinterpret_cast<const unsigned char*>(images)[base + j] * (1.0f / 255.0f)
                                                 : reinterpret_cast<const float*>(images)[base + j];
                            v = 2.0f * raw - 1.0f;
                        }
                        in[q][ky * 9 + j] = v;
                    }
                }
            }
        }
        const int pl = ly * kFwdTW + lx * 2;                     // tile pixel of the first output
        for (int c0 = 0; c0 < C0; c0 += VE) {
            f32x2_t acc2[2][VE / 2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < VE / 2; ++j) acc2[q][j] = (f32x2_t){0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const f32x2_t iv0 = (f32x2_t){in[0][t], in[0][t]}, iv1 = (f32x2_t){in[1][t], in[1][t]};
#pragma unroll
                for (int j = 0; j < VE; j += 4) {
                    const float4 q4 = *reinterpret_cast<const float4*>(&w[t * C0 + c0 + j]);   // uniform address: scalar load
                    const f32x2_t w01 = (f32x2_t){q4.x, q4.y}, w23 = (f32x2_t){q4.z, q4.w};
                    acc2[0][j / 2] += iv0 * w01; acc2[0][j / 2 + 1] += iv0 * w23;
                    acc2[1][j / 2] += iv1 * w01; acc2[1][j / 2 + 1] += iv1 * w23;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float acc[VE];
#pragma unroll
                for (int j = 0; j < VE / 2; ++j) { acc[2 * j] = acc2[q][j].x; acc[2 * j + 1] = acc2[q][j].y; }
                Vec16<T> ov;
                ov.pack(acc);
                *reinterpret_cast<uint4*>(otile + (pl + q) * orow + c0 * (int)sizeof(T)) = *reinterpret_cast<const uint4*>(&ov.raw);
            }
        }
    }
    __syncthreads();
    stem_copy_out<T>(otile, orow, y, img, ty, tx, OH, OW, C0, stats_part);
}

// bf16 build: the same convolution on the matrix cores. K = 27 pads to ONE 16x16x32 step: k = ky * 9 + kx * 3 + c is the
// order of the HWIO kernel AND of three 9-float runs of the input patch. The block stages the standardised 33 x 65 x 3 f32
// patch of its 16 x 32 output pixels in LDS (coalesced dword loads, all in flight at once); a lane builds its operand
// fragment - 8 consecutive k of one pixel - from 8 LDS dwords. Inputs and weights stay f32-accurate: x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi), and the product is hi*hi + hi*lo + lo*hi (three MFMAs, error ~2^-17 relative; the matrix
// time is nothing next to the 235 MB this kernel moves). Weights are the A operand (rows = output channels) so that a
// lane ends up with 4 consecutive channels of one pixel; outputs leave through the same LDS tile and copy-out (and fused
// statistics) as the f32 kernel, the tile aliasing the patch.
constexpr int kPR = 2 * kTile + 1;            // 33 patch rows
constexpr int kPC = (2 * kFwdTW + 1) * 3;     // 195 floats per patch row
constexpr int kPS = kPC + 1;                  // row stride in LDS
template <bool U8, int MB>
__global__ __launch_bounds__(kThreads) void stem_fwd_mfma_kernel(const void* __restrict__ images, const float* __restrict__ w,
                                                                 bf16_t* __restrict__ y, int N, int H, int W, int OH, int OW,
                                                                 int pad_t, int pad_l, int tiles_x, int tiles_y,
                                                                 float* __restrict__ stats_part) {
    typedef H16<bf16_t> HT;
    typedef HT::x8 x8;
    typedef HT::acc_t acc_t;
    constexpr int C0 = 16 * MB;
    extern __shared__ __attribute__((aligned(16))) unsigned char stem_smem[];
    float* patch = reinterpret_cast<float*>(stem_smem);                     // [33][196]
    unsigned char* otile = stem_smem;                                       // [512 px][C0 * 2 + 16], after the last patch read
    constexpr int orow = C0 * 2 + 16;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int img = b / tiles_y;
    // ---- patch: every load issued before the first use. Thread t < 195 owns float t of every patch row (its column and channel
    //      are fixed, the row is the unrolled loop variable): no index arithmetic per element - with element i = t + 256 k
    //      spread over rows the divisions by 195 and 3 made the kernel VALU-bound (PMC: 70 % of the issue slots)
    {
        const int iy0 = ty * kTile * 2 - pad_t, ix0 = tx * kFwdTW * 2 - pad_l;
        const int pc = threadIdx.x < kPC ? threadIdx.x : 0;
        const int ix = ix0 + pc / 3;
        const bool col_ok = threadIdx.x < kPC && ix >= 0 && ix < W;
        float v[kPR];
        const long long base = ((long long)img * H * W + ix0) * 3 + pc;
#pragma unroll
        for (int py = 0; py < kPR; ++py) {
            const int iy = iy0 + py;                                             // (uniform)
            const bool ok = col_ok && iy >= 0 && iy < H;
            const long long off = ok ? base + (long long)iy * W * 3 : 0;        // (unpredicated load, valid address)
            const float raw = U8 ? (float)reinterpret_cast<const unsigned char*>(images)[off] * (1.0f / 255.0f)
                                 : reinterpret_cast<const float*>(images)[off];
            v[py] = ok ? 2.0f * raw - 1.0f : 0.0f;                              // SAME padding: zeros of the STANDARDISED tensor
        }
        if (threadIdx.x < kPC) {
#pragma unroll
            for (int py = 0; py < kPR; ++py) patch[py * kPS + pc] = v[py];
        }
    }
    // ---- weight fragments (A: row = output channel m of block mb, k = 8 * kg .. + 7), hi / lo
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, kg = lane >> 4;
    x8 a_hi[MB], a_lo[MB];
    int koff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * kg + i;
        koff[i] = k < 27 ? (k / 9) * kPS + (k % 9) : 0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const float wk = k < 27 ? w[k * C0 + mb * 16 + n] : 0.0f;
            const bf16_t h = (bf16_t)wk;
            a_hi[mb][i] = h;
            a_lo[mb][i] = (bf16_t)(wk - (float)h);
        }
    }
    __syncthreads();
    // ---- 8 groups of 16 pixels per wave: rows wv * 4 .. + 3, two half rows each
    acc_t acc[8][MB];
#pragma unroll
    for (int gi = 0; gi < 8; ++gi) {
        const int r = wv * 4 + (gi >> 1), col = (gi & 1) * 16 + n;
        const float* src = patch + (2 * r) * kPS + (2 * col) * 3;
        x8 b_hi, b_lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float f = src[koff[i]];
            const bf16_t h = (bf16_t)f;
            b_hi[i] = h;
            b_lo[i] = (bf16_t)(f - (float)h);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            acc_t c = {0.f, 0.f, 0.f, 0.f};
            c = HT::mfma(a_lo[mb], b_hi, c);
            c = HT::mfma(a_hi[mb], b_lo, c);
            c = HT::mfma(a_hi[mb], b_hi, c);
            acc[gi][mb] = c;
        }
    }
    __syncthreads();                                                        // the patch is dead: the output tile takes its place
#pragma unroll
    for (int gi = 0; gi < 8; ++gi) {
        const int r = wv * 4 + (gi >> 1), col = (gi & 1) * 16 + n;
        unsigned char* dst = otile + (r * kFwdTW + col) * orow + (4 * kg) * 2;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const acc_t c = acc[gi][mb];
            uint2 q;
            q.x = pack_bf16x2(c[0], c[1]);
            q.y = pack_bf16x2(c[2], c[3]);
            *reinterpret_cast<uint2*>(dst + mb * 32) = q;
        }
    }
    __syncthreads();
    stem_copy_out<bf16_t>(otile, orow, y, img, ty, tx, OH, OW, C0, stats_part);
}

// weight gradient: dW[27][C0] = sum_pixels patch[27] (x) dy[C0]. Register-blocked outer product out of LDS:
// a thread owns one (ky,kx) position (its 3 input channels are 12 contiguous bytes of the patch) x 8 output
// channels = 24 accumulators and 24 FMAs per 3 LDS reads; the 9 x C0/8 such threads form a "phase" and the 256
// threads run as many phases as fit, each on its own share of the tile's 256 pixels. Blocks walk tiles with the
// accumulators in registers; phases are summed through LDS once, partials [nblocks][27*C0] go to mpn_reduce_partials.
template <typename T, bool U8>
__global__ __launch_bounds__(kThreads) void stem_wgrad_kernel(const void* __restrict__ images, const T* __restrict__ dy,
                                                              float* __restrict__ part, int N, int H, int W, int C0, int OH,
                                                              int OW, int pad_t, int pad_l, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float dyn_smem[];
    float* patch = dyn_smem;                      // [33*33*3] (+ pad to 16 B)
    float* g = dyn_smem + kIn * kIn * 3 + 1;      // [256][C0]  (16-byte aligned: 3268 floats)
    const int ncq = C0 / 8;                       // 8-channel groups
    const int tpp = 9 * ncq;                      // threads per phase
    // (at most 9 phases: their sums meet in the dY tile's 256 * C0 floats of LDS, 27 * C0 per phase - with C0 = 16 the 14 phases
    //  that fit the block overran it)
    const int nphase = min(kThreads / tpp, kTile * kTile / 27);
    const int ph = threadIdx.x / tpp;
    const int r = threadIdx.x - ph * tpp;
    const int pos = r / ncq, cq = r - pos * ncq;  // pos = ky*3+kx
    const int ky = pos / 3, kx = pos - ky * 3;
    const bool active = ph < nphase;
    f32x2_t acc[3][4];                            // channel pairs: packed FP32 FMAs (v_pk_fma_f32), 12 per pixel instead of 24
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[a][j] = (f32x2_t){0.f, 0.f};
    const int nout = 27 * C0;
    const int ntiles = N * tiles_y * tiles_x;
    // Software pipeline over tiles: the NEXT tile's image patch and dy vectors are loaded into registers (unpredicated
    // loads from clamped addresses + validity masks) before this tile is multiplied, and committed to LDS afterwards -
    // one HBM round trip per tile is hidden under the outer products instead of being waited for at every barrier.
    constexpr int NEL = kIn * kIn * 3;
    constexpr int PER = (NEL + kThreads - 1) / kThreads;   // 13 patch elements per thread
    constexpr int VE = Vec16<T>::N;
    constexpr int DPER = kTile * kTile * (kMaxC0 / VE) / kThreads;   // <= 8 (bf16) / 16 (f32) dy vectors per thread
    const int vpp = C0 / VE;                       // 16-byte vectors per pixel
    const int ndv = kTile * kTile * vpp / kThreads; // dy vectors per thread actually used (C0 multiple of VE, <= DPER)
    float pv[PER];
    Vec16<T> dvv[DPER];
    unsigned pmask = 0u, dmask = 0u;

    auto load_tile = [&](int t) {
        const int tx = t % tiles_x;
        const int t2 = t / tiles_x;
        const int ty = t2 % tiles_y;
        const int img = t2 / tiles_y;
        const int oy0 = ty * kTile, ox0 = tx * kTile;
        const int iy0 = oy0 * 2 - pad_t, ix0 = ox0 * 2 - pad_l;
        pmask = 0u;
        dmask = 0u;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * kThreads;
            const int ch = i % 3;
            const int px = (i / 3) % kIn;
            const int py = i / (3 * kIn);
            const int iy = iy0 + py, ix = ix0 + px;
            const bool ok = i < NEL && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const long long off = ok ? (((long long)img * H + iy) * W + ix) * 3 + ch : 0;
            pv[k] = U8 ? (float)reinterpret_cast<const unsigned char*>(images)[off] * (1.0f / 255.0f)
                       : reinterpret_cast<const float*>(images)[off];
            pmask |= (ok ? 1u : 0u) << k;
        }
#pragma unroll
        for (int k = 0; k < DPER; ++k) {
            if (k < ndv) {
                const int i = threadIdx.x + k * kThreads;
                const int vq = i % vpp, px = i / vpp;
                const int oy = oy0 + px / kTile, ox = ox0 + px % kTile;
                const bool ok = oy < OH && ox < OW;
                dvv[k].load(dy + (ok ? (((long long)img * OH + oy) * OW + ox) * C0 + vq * VE : 0));
                dmask |= (ok ? 1u : 0u) << k;
            }
        }
    };
    auto commit_tile = [&]() {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * kThreads;
            // SAME padding is zeros of the STANDARDISED tensor (2*0.5-1)
            if (i < NEL) patch[i] = ((pmask >> k) & 1u) ? 2.0f * pv[k] - 1.0f : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < DPER; ++k) {
            if (k < ndv) {
                const int i = threadIdx.x + k * kThreads;
                const int vq = i % vpp, px = i / vpp;
                float f[VE];
                dvv[k].unpack(f);
                const bool ok = (dmask >> k) & 1u;
#pragma unroll
                for (int j = 0; j < VE; j += 4)
                    *reinterpret_cast<float4*>(g + px * C0 + vq * VE + j) =
                        ok ? make_float4(f[j], f[j + 1], f[j + 2], f[j + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    int t = blockIdx.x;
    if (t < ntiles) {
        load_tile(t);
        commit_tile();
    }
    for (; t < ntiles; t += gridDim.x) {
        const bool more = t + (int)gridDim.x < ntiles;
        if (more) load_tile(t + gridDim.x);
        __syncthreads();   // this tile's patch / dy are complete
        if (active) {
            for (int px = ph; px < kTile * kTile; px += nphase) {
                const int ly = px / kTile, lx = px - ly * kTile;
                const float* pp = patch + ((2 * ly + ky) * kIn + 2 * lx + kx) * 3;
                const float a0 = pp[0], a1 = pp[1], a2 = pp[2];
                const float4 g0 = *reinterpret_cast<const float4*>(g + px * C0 + cq * 8);
                const float4 g1 = *reinterpret_cast<const float4*>(g + px * C0 + cq * 8 + 4);
                const f32x2_t gv[4] = {(f32x2_t){g0.x, g0.y}, (f32x2_t){g0.z, g0.w}, (f32x2_t){g1.x, g1.y}, (f32x2_t){g1.z, g1.w}};
                const f32x2_t b0 = (f32x2_t){a0, a0}, b1 = (f32x2_t){a1, a1}, b2 = (f32x2_t){a2, a2};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][j] += b0 * gv[j];
                    acc[1][j] += b1 * gv[j];
                    acc[2][j] += b2 * gv[j];
                }
            }
        }
        __syncthreads();   // everybody is done reading this tile
        if (more) commit_tile();
    }
    // sum the phases through LDS (reuse g: nphase * 27*C0 floats <= 256*C0)
    __syncthreads();
    if (active) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[ph * nout + (pos * 3 + a) * C0 + cq * 8 + 2 * j] = acc[a][j].x;
                g[ph * nout + (pos * 3 + a) * C0 + cq * 8 + 2 * j + 1] = acc[a][j].y;
            }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < nout; o += kThreads) {
        float s = 0.f;
        for (int q = 0; q < nphase; ++q) s += g[q * nout + o];
        part[(long long)blockIdx.x * nout + o] = s;
    }
}

// bf16 build: the weight gradient on the matrix cores, dW^T[co][k] = sum over pixels dY[px][co] * P[px][k] with the pixel
// index as the MFMA's K (32 = one row of a 16 x 32 tile per step). The dY tile sits in LDS pixel-major as it is in HBM and
// is read channel-major by ds_read_b64_tr_b16 (A operand: row = output channel); the B operand (column = k, the order of
// stem_fwd_mfma_kernel) is built from the f32 patch - 8 LDS dwords at the 6-float pitch of neighbouring output pixels -
// and split hi + lo like the forward's, so the products keep f32 accuracy in the image (dY is bf16 already). Persistent
// blocks, next tile's patch and dY prefetched into registers under the current tile's products; a wave owns 4 of the 16
// rows; the four waves' 27 x C0 sums meet in LDS once at the end (fixed order), one slab row per block.
template <bool U8, int MB>
__global__ __launch_bounds__(kThreads) void stem_wgrad_mfma_kernel(const void* __restrict__ images, const bf16_t* __restrict__ dy,
                                                                   float* __restrict__ part, int N, int H, int W, int OH, int OW,
                                                                   int pad_t, int pad_l, int tiles_x, int tiles_y) {
    typedef H16<bf16_t> HT;
    typedef HT::x8 x8;
    typedef HT::x4 x4;
    typedef HT::acc_t acc_t;
    constexpr int C0 = 16 * MB;
    constexpr int DRS = C0 * 2 + 16;                                        // bytes per pixel row of the dY image
    extern __shared__ __attribute__((aligned(16))) unsigned char stem_smem[];
    float* patch = reinterpret_cast<float*>(stem_smem);                     // [33][196] f32
    unsigned char* dimg = stem_smem + kPR * kPS * sizeof(float);            // [512 px][DRS]
    const int ntiles = N * tiles_y * tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, kg = lane >> 4, q = n >> 2, pq = n & 3;
    constexpr int VPP = C0 / 8;                                             // 16-byte dY vectors per pixel
    constexpr int DV = kTile * kFwdTW * VPP / kThreads;                     // 4 * MB per thread
    // patch: thread t < 195 owns float t of every patch row (as in the forward kernel: no index arithmetic per element)
    float pv[kPR];
    uint4 dv[DV];
    unsigned long long pmask = 0ull;
    unsigned dmask = 0u;
    const int p_pc = threadIdx.x < kPC ? threadIdx.x : 0;
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        const int tx = t % tiles_x;
        const int t2 = t / tiles_x;
        const int ty = t2 % tiles_y;
        const int img = t2 / tiles_y;
        const int iy0 = ty * kTile * 2 - pad_t, ix0 = tx * kFwdTW * 2 - pad_l;
        pmask = 0ull;
        dmask = 0u;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));   // per-call index arithmetic: hoisted out of the tile loop it pins ~100 address registers
        {
            const int ix = ix0 + p_pc / 3;
            const bool col_ok = tid < kPC && ix >= 0 && ix < W;
            const long long base = ((long long)img * H * W + ix0) * 3 + p_pc;
#pragma unroll
            for (int py = 0; py < kPR; ++py) {
                const int iy = iy0 + py;
                const bool ok = col_ok && iy >= 0 && iy < H;
                const long long off = ok ? base + (long long)iy * W * 3 : 0;
                pv[py] = U8 ? (float)reinterpret_cast<const unsigned char*>(images)[off] * (1.0f / 255.0f)
                            : reinterpret_cast<const float*>(images)[off];
                pmask |= (unsigned long long)(ok ? 1u : 0u) << py;
            }
        }
#pragma unroll
        for (int k = 0; k < DV; ++k) {
            const int i = tid + k * kThreads;
            const int vq = i % VPP, px = i / VPP;
            const int oy = ty * kTile + px / kFwdTW, ox = tx * kFwdTW + px % kFwdTW;
            const bool ok = oy < OH && ox < OW;
            dv[k] = *reinterpret_cast<const uint4*>(dy + (ok ? (((long long)img * OH + oy) * OW + ox) * C0 + vq * 8 : 0));
            dmask |= (ok ? 1u : 0u) << k;
        }
    };
    auto commit_tile = [&]() __attribute__((always_inline)) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (tid < kPC) {
#pragma unroll
            for (int py = 0; py < kPR; ++py)   // SAME padding is zeros of the STANDARDISED tensor
                patch[py * kPS + p_pc] = ((pmask >> py) & 1ull) ? 2.0f * pv[py] - 1.0f : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < DV; ++k) {
            const int i = tid + k * kThreads;
            const int vq = i % VPP, px = i / VPP;
            *reinterpret_cast<uint4*>(dimg + px * DRS + vq * 16) = ((dmask >> k) & 1u) ? dv[k] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    // this lane's two columns k = n and 16 + n of the B operand: patch offset of (ky, kx * 3 + c), or a zero column
    int kbase[2];
    bool kval[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int k = nb * 16 + n;
        kval[nb] = k < 27;
        kbase[nb] = kval[nb] ? (k / 9) * kPS + (k % 9) : 0;
    }
    acc_t acc[MB][2];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = (acc_t){0.f, 0.f, 0.f, 0.f};

    int t = blockIdx.x;
    if (t < ntiles) {
        load_tile(t);
        commit_tile();
    }
    for (; t < ntiles; t += gridDim.x) {
        const bool more = t + (int)gridDim.x < ntiles;
        if (more) load_tile(t + gridDim.x);
        __syncthreads();   // this tile's patch / dY are complete
#pragma unroll 1
        for (int sr = 0; sr < 4; ++sr) {   // (rolled: unrolled, hipcc hoists all four rows' 64 patch reads - 294 VGPRs, one block per CU)
            const int r = wv * 4 + sr;
            x8 a[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const unsigned char* src = dimg + (r * kFwdTW + 8 * kg + q) * DRS + (mb * 16 + 4 * pq) * 2;
                const x4 lo4 = HT::tr_read(src);
                const x4 hi4 = HT::tr_read(src + 4 * DRS);
                a[mb] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const float* src = patch + (2 * r) * kPS + 6 * (8 * kg) + kbase[nb];
                x8 b_hi, b_lo;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float raw = src[6 * i];
                    const float f = kval[nb] ? raw : 0.0f;
                    const bf16_t h = (bf16_t)f;
                    b_hi[i] = h;
                    b_lo[i] = (bf16_t)(f - (float)h);
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    acc[mb][nb] = HT::mfma(a[mb], b_lo, acc[mb][nb]);
                    acc[mb][nb] = HT::mfma(a[mb], b_hi, acc[mb][nb]);
                }
            }
        }
        __syncthreads();   // everybody is done reading this tile
        if (more) commit_tile();
    }
    // the four waves' sums: lane (n, kg) of block (mb, nb) holds dW[k = nb * 16 + n][co = mb * 16 + 4 * kg + e], e = 0..3
    __syncthreads();
    float* red = reinterpret_cast<float*>(stem_smem);                        // [4][MB * 2][64][4]
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            *reinterpret_cast<float4*>(red + ((wv * (MB * 2) + mb * 2 + nb) * 64 + lane) * 4) =
                make_float4(acc[mb][nb][0], acc[mb][nb][1], acc[mb][nb][2], acc[mb][nb][3]);
    __syncthreads();
    for (int o = threadIdx.x; o < 27 * C0; o += kThreads) {
        const int k = o / C0, co = o - k * C0;
        const int nb = k >> 4, nn = k & 15, mb = co >> 4, m = co & 15;
        const int ln = (m >> 2) * 16 + nn, e = m & 3;
        float sacc = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) sacc += red[((w4 * (MB * 2) + mb * 2 + nb) * 64 + ln) * 4 + e];
        part[(long long)blockIdx.x * (27 * C0) + o] = sacc;
    }
}

template <bool U8, int MB>
int launch_stem_wgrad_mfma(int grid, hipStream_t st, const void* images, const void* dy, float* part, int N, int H, int W, int OH,
                           int OW, int pt, int pl, int tiles_x, int tiles_y) {
    constexpr int sm = kPR * kPS * (int)sizeof(float) + kTile * kFwdTW * (16 * MB * 2 + 16);
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)stem_wgrad_mfma_kernel<U8, MB>, sm, &attr_mask));
    stem_wgrad_mfma_kernel<U8, MB><<<grid, kThreads, sm, st>>>(images, (const bf16_t*)dy, part, N, H, W, OH, OW, pt, pl, tiles_x, tiles_y);
    return MPN_OK;
}

void same_pad(int size, int* out, int* pad_before) {
    *out = (size + 1) / 2;
    int total = (*out - 1) * 2 + 3 - size;
    if (total < 0) total = 0;
    *pad_before = total / 2;
}

int check(int N, int H, int W, int C0, int dtype) {
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "stem: dtype %d", dtype);
    const int ve = dtype == MPN_F32 ? 4 : 8;
    MPN_REQUIRE(N > 0 && H > 0 && W > 0, MPN_ERR_BAD_SHAPE, "stem: bad shape");
    MPN_REQUIRE(C0 > 0 && C0 <= kMaxC0 && C0 % ve == 0, MPN_ERR_BAD_SHAPE, "stem: C0 (%d) must be <= %d and a multiple of %d",
                C0, kMaxC0, ve);
    return MPN_OK;
}

}  // namespace

/* rows of the statistics slab mpn_stem_conv_fwd_stats writes (one per output tile); 0 = the fused statistics are not
 * available for this channel count (use mpn_bn_stats on the output instead) */
extern "C" int mpn_stem_conv_fwd_num_parts(int N, int H, int W, int C0, int dtype) {
    if (check(N, H, W, C0, dtype)) return 0;
    const int ppr = C0 * (dtype == MPN_F32 ? 4 : 2) / 16;
    if (ppr <= 0 || kThreads % ppr != 0 || 2 * C0 > kThreads) return 0;
    int OH, OW, pt, pl;
    same_pad(H, &OH, &pt);
    same_pad(W, &OW, &pl);
    return N * ((OH + kTile - 1) / kTile) * ((OW + kFwdTW - 1) / kFwdTW);
}

extern "C" int mpn_stem_conv_fwd(const void* images, int images_u8, const float* w, void* y, int N, int H, int W, int C0,
                                 int dtype, mpn_stream_t stream) {
    return mpn_stem_conv_fwd_stats(images, images_u8, w, y, N, H, W, C0, dtype, nullptr, stream);
}

/* mpn_stem_conv_fwd + the batch-norm statistics of its (rounded) output: stats_part [mpn_stem_conv_fwd_num_parts][2][C0]
 * receives per-tile sum and sum of squares in the layout of mpn_bn_stats (finish with mpn_bn_finalize) - the separate
 * statistics pass over the largest activation of the network (134 MB at bs32 @ 512x512) is gone. stats_part == NULL:
 * plain forward. */
extern "C" int mpn_stem_conv_fwd_stats(const void* images, int images_u8, const float* w, void* y, int N, int H, int W, int C0,
                                       int dtype, float* stats_part, mpn_stream_t stream) {
    if (int rc = check(N, H, W, C0, dtype)) return rc;
    MPN_REQUIRE(images && w && y, MPN_ERR_BAD_ARG, "stem_fwd: null pointer");
    MPN_REQUIRE(stats_part == nullptr || mpn_stem_conv_fwd_num_parts(N, H, W, C0, dtype) > 0, MPN_ERR_BAD_SHAPE,
                "stem_fwd: fused statistics not available for C0 = %d (mpn_stem_conv_fwd_num_parts == 0)", C0);
    int OH, OW, pt, pl;
    same_pad(H, &OH, &pt);
    same_pad(W, &OW, &pl);
    const int tiles_y = (OH + kTile - 1) / kTile, tiles_x = (OW + kFwdTW - 1) / kFwdTW;
    const int grid = N * tiles_y * tiles_x;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPN_BF16 && C0 % 16 == 0) {                                // matrix-core kernel
        const int otile_bytes = 2 * kThreads * (C0 * 2 + 16), patch_bytes = kPR * kPS * (int)sizeof(float);
        const int sm = otile_bytes > patch_bytes ? otile_bytes : patch_bytes;
#define MPN_STEM_MFMA(MB)                                                                                                       \
    do {                                                                                                                        \
        if (images_u8) stem_fwd_mfma_kernel<true, MB><<<grid, kThreads, sm, st>>>(images, w, (bf16_t*)y, N, H, W, OH, OW, pt, pl, tiles_x, tiles_y, stats_part); \
        else stem_fwd_mfma_kernel<false, MB><<<grid, kThreads, sm, st>>>(images, w, (bf16_t*)y, N, H, W, OH, OW, pt, pl, tiles_x, tiles_y, stats_part); \
    } while (0)
        if (sm <= 64 * 1024) {
            switch (C0 / 16) {
                case 1: MPN_STEM_MFMA(1); break;
                case 2: MPN_STEM_MFMA(2); break;
                case 3: MPN_STEM_MFMA(3); break;
                default: MPN_STEM_MFMA(4); break;
            }
#undef MPN_STEM_MFMA
            MPN_LAUNCH_CHECK();
            return MPN_OK;
        }
    }
    MPN_DISPATCH_DTYPE(dtype, {
        const int sm = 2 * kThreads * (C0 * (int)sizeof(T) + 16);
        if (images_u8) {
            if (sm > 48 * 1024)
                MPN_HIP(hipFuncSetAttribute((const void*)stem_fwd_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm));
            stem_fwd_kernel<T, true><<<grid, kThreads, sm, st>>>(images, w, (T*)y, N, H, W, C0, OH, OW, pt, pl, tiles_x, tiles_y, stats_part);
        } else {
            if (sm > 48 * 1024)
                MPN_HIP(hipFuncSetAttribute((const void*)stem_fwd_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm));
            stem_fwd_kernel<T, false><<<grid, kThreads, sm, st>>>(images, w, (T*)y, N, H, W, C0, OH, OW, pt, pl, tiles_x, tiles_y, stats_part);
        }
    });
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_stem_conv_wgrad_num_parts(int N, int H, int W) {
    const int ntiles = N * (((H + 1) / 2 + kTile - 1) / kTile) * (((W + 1) / 2 + kTile - 1) / kTile);
    return ntiles < 512 ? ntiles : 512;   // 2 resident blocks per CU (register-bound): one balanced round
}

extern "C" int mpn_stem_conv_bwd_weight(const void* images, int images_u8, const void* dy, float* part, int N, int H, int W,
                                        int C0, int dtype, mpn_stream_t stream) {
    if (int rc = check(N, H, W, C0, dtype)) return rc;
    MPN_REQUIRE(images && dy && part, MPN_ERR_BAD_ARG, "stem_wgrad: null pointer");
    int OH, OW, pt, pl;
    same_pad(H, &OH, &pt);
    same_pad(W, &OW, &pl);
    const int tiles_y = (OH + kTile - 1) / kTile, tiles_x = (OW + kTile - 1) / kTile;
    const int grid = mpn_stem_conv_wgrad_num_parts(N, H, W);
    hipStream_t st = (hipStream_t)stream;
    MPN_REQUIRE(C0 % 8 == 0, MPN_ERR_BAD_SHAPE, "stem_wgrad: C0 must be a multiple of 8");
    if (dtype == MPN_BF16 && C0 % 16 == 0) {                                // matrix-core kernel, 16 x 32-pixel tiles
        const int tx32 = (OW + kFwdTW - 1) / kFwdTW;
        int rc;
#define MPN_STEM_WG(MB)                                                                                                         \
    rc = images_u8 ? launch_stem_wgrad_mfma<true, MB>(grid, st, images, dy, part, N, H, W, OH, OW, pt, pl, tx32, tiles_y)       \
                   : launch_stem_wgrad_mfma<false, MB>(grid, st, images, dy, part, N, H, W, OH, OW, pt, pl, tx32, tiles_y)
        switch (C0 / 16) {
            case 1: MPN_STEM_WG(1); break;
            case 2: MPN_STEM_WG(2); break;
            case 3: MPN_STEM_WG(3); break;
            default: MPN_STEM_WG(4); break;
        }
#undef MPN_STEM_WG
        if (rc) return rc;
        MPN_LAUNCH_CHECK();
        return MPN_OK;
    }
    const size_t sm = (size_t)(kIn * kIn * 3 + 1 + kTile * kTile * C0) * sizeof(float);
    MPN_REQUIRE(sm <= 64 * 1024, MPN_ERR_BAD_SHAPE, "stem_wgrad: C0 too large");
    MPN_DISPATCH_DTYPE(dtype, {
        if (images_u8)
            stem_wgrad_kernel<T, true><<<grid, kThreads, sm, st>>>(images, (const T*)dy, part, N, H, W, C0, OH, OW, pt, pl, tiles_x, tiles_y);
        else
            stem_wgrad_kernel<T, false><<<grid, kThreads, sm, st>>>(images, (const T*)dy, part, N, H, W, C0, OH, OW, pt, pl, tiles_x, tiles_y);
    });
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
