// PROTOTYPE (not part of the product library): the 3x3 convolution's main loop as FOUR waves per block, one per SIMD, with
// 128-pixel x 64-channel wave tiles - the candidate DESIGN.md 6 listed as untried. Forward only, bf16, Cin % 64 == 0,
// Cout == 128, H % 16 == W % 16 == 0, no producer affine, no statistics: enough to compare the main loop's matrix-pipe
// utilisation with the shipped 8-wave kernel on the same shapes (tools/proto/run_c3w4.py).
//   tile     16 x 16 pixels x 128 output channels, persistent blocks over tiles;
//   chunk    32 input channels = one 16x16x32 K step; halo image 18 x 18 pixels x 64 B (+16 B pad), double-buffered;
//   stage    the three taps of one kernel column kx for the chunk: [ky][co 128][32 ci] bf16 = 24 576 B, ring of three slots,
//            staged through registers (global_load -> ds_write);
//   wave     rows 8 * wm .. + 7 of the tile x channels 64 * wn .. + 63: 96 MFMAs per stage from 12 weight + 10 pixel
//            fragments (output row r at tap ky and row r + 1 at ky - 1 share a halo row), fragments of stage s + 1 read
//            into a second register set while stage s multiplies; ONE barrier per stage.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 x8 __attribute__((ext_vector_type(8)));
typedef float acc_t __attribute__((ext_vector_type(4)));

namespace {
constexpr int kThreads = 256;
constexpr int kPitch = 80;                       // bytes per 32-channel row in LDS (64 + 16 pad: conflict-free fragment reads)
constexpr int kHaloPx = 18 * 18;
constexpr int kHaloBytes = kHaloPx * kPitch;     // 25 920
constexpr int kStageRows = 3 * 128;
constexpr int kStageBytes = kStageRows * kPitch; // 30 720 in LDS (24 576 in HBM)
constexpr int kHaloPer = (kHaloPx * 4 + kThreads - 1) / kThreads;   // 6 pieces of 16 B per thread and chunk
constexpr int kWPer = kStageRows * 4 / kThreads;                    // 6 pieces per thread and stage

struct PixFrags { x8 b[10]; };
struct WFrags { x8 a[4]; };

__global__ __launch_bounds__(kThreads) void c3w4_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wpk,
                                                         bf16_t* __restrict__ y, int N, int H, int W, int Cin,
                                                         long long* __restrict__ stamps, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* halo = sm;                               // [2][kHaloBytes]
    unsigned char* ring = sm + 2 * kHaloBytes;              // [3][kStageBytes]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, kg = lane >> 4;
    const int wm = wv & 1, wn = wv >> 1;
    const int nchunk = Cin / 32, nst = 3 * nchunk;          // stages per tile
    const int tiles_x = W / 16, tiles_y = H / 16;
    const int ntiles = N * tiles_y * tiles_x;
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_tiles == 0) return;
    const int total = (int)my_tiles * nst;      // flat stage count of this block

    // ---- staging registers
    uint4 wq0, wq1, wq2, wq3, wq4, wq5, hq[kHaloPer];   // (named: as an array hipcc keeps the weight pieces in scratch)
    static_assert(kWPer == 6, "six weight pieces per thread");
    unsigned hmask = 0u;
    auto tile_of = [&](int s) { return (int)blockIdx.x + (s / nst) * (int)gridDim.x; };
    auto w_load = [&](int s) __attribute__((always_inline)) {            // weights of flat stage s
        const int st = (int)(s % nst);
        const uint4* src = reinterpret_cast<const uint4*>(wpk) + (long long)st * (24576 / 16);
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        wq0 = src[tid]; wq1 = src[tid + kThreads]; wq2 = src[tid + 2 * kThreads];
        wq3 = src[tid + 3 * kThreads]; wq4 = src[tid + 4 * kThreads]; wq5 = src[tid + 5 * kThreads];
    };
    auto w_commit = [&](int s) __attribute__((always_inline)) {
        unsigned char* dst = ring + (int)(s % 3) * kStageBytes;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        unsigned char* d0 = dst + (tid >> 2) * kPitch + (tid & 3) * 16;       // piece j = tid + i * 256: row j / 4 = tid / 4 + 64 i
        constexpr int rs = 64 * kPitch;
        *reinterpret_cast<uint4*>(d0) = wq0; *reinterpret_cast<uint4*>(d0 + rs) = wq1; *reinterpret_cast<uint4*>(d0 + 2 * rs) = wq2;
        *reinterpret_cast<uint4*>(d0 + 3 * rs) = wq3; *reinterpret_cast<uint4*>(d0 + 4 * rs) = wq4; *reinterpret_cast<uint4*>(d0 + 5 * rs) = wq5;
    };
    auto h_load = [&](int q) __attribute__((always_inline)) {            // halo of flat chunk q
        const int tile = (int)blockIdx.x + (int)(q / nchunk) * (int)gridDim.x;
        const int c = (int)(q % nchunk);
        const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, img = t2 / tiles_y;
        const int iy0 = ty * 16 - 1, ix0 = tx * 16 - 1;
        const bf16_t* ximg = x + (long long)img * H * W * Cin + c * 32;
        hmask = 0u;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int i = 0; i < kHaloPer; ++i) {
            const int j = tid + i * kThreads;
            const int px = j >> 2, hy = px / 18, hx = px - hy * 18;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = j < kHaloPx * 4 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            hq[i] = *reinterpret_cast<const uint4*>(ximg + (ok ? ((long long)iy * W + ix) * Cin + (j & 3) * 8 : 0));
            hmask |= (ok ? 1u : 0u) << i;
        }
    };
    auto h_commit = [&](int q) __attribute__((always_inline)) {
        unsigned char* dst = halo + (int)(q & 1) * kHaloBytes;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int i = 0; i < kHaloPer; ++i) {
            const int j = tid + i * kThreads;
            if (j < kHaloPx * 4)
                *reinterpret_cast<uint4*>(dst + (j >> 2) * kPitch + (j & 3) * 16) = ((hmask >> i) & 1u) ? hq[i] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    // ---- fragments of flat stage s: pixels from halo buffer (s / 3) & 1 at column offset kx = s % 3 (all ten of a stage, read
    //      one stage ahead); weights of tap ky from ring slot s % 3 (four per tap, read one tap ahead)
    auto pix_read = [&](PixFrags& f, int s) __attribute__((always_inline)) {
        const int kx = (int)(s % 3);
        const unsigned char* hb = halo + (int)((s / 3) & 1) * kHaloBytes + ((8 * wm) * 18 + n + kx) * kPitch + kg * 16;
#pragma unroll
        for (int j = 0; j < 10; ++j) f.b[j] = *reinterpret_cast<const x8*>(hb + j * 18 * kPitch);
    };
    auto w_read = [&](WFrags& f, int s, int ky) __attribute__((always_inline)) {
        const unsigned char* rb = ring + (int)(s % 3) * kStageBytes + (ky * 128 + 64 * wn + n) * kPitch + kg * 16;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) f.a[nb] = *reinterpret_cast<const x8*>(rb + 16 * nb * kPitch);
    };

    acc_t acc[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[r][nb] = (acc_t){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: halo of chunk 0, weights of stages 0 and 1
    h_load(0); h_commit(0);
    w_load(0); w_commit(0);
    if (total > 1) { w_load(1); w_commit(1); }
    __syncthreads();
    PixFrags P0, P1;
    WFrags A0, A1, A2;
    pix_read(P0, 0);
    pix_read(P1, 0);
    w_read(A0, 0, 0);
    w_read(A1, 0, 1);
    w_read(A2, 0, 2);
    long long t_begin = 0;
    if (stamps && threadIdx.x == 0) t_begin = __builtin_readcyclecounter();

    auto taps = [&](const WFrags& a, const PixFrags& pf, int ky) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.a[nb], pf.b[r + ky], acc[r][nb], 0, 0, 0);
    };
    auto stage = [&](PixFrags& cur, PixFrags& nxt, int s) __attribute__((always_inline)) {
        const int kx = (int)(s % 3);
        const int q = s / 3;
        const bool ld = !(mode & 1), rd = !(mode & 2);   // knock-outs (diagnostic: wrong results)
        if (ld && s + 2 < total) w_load(s + 2);
        if (ld && kx == 0 && (q + 1) * 3 < total) h_load(q + 1);
        if (rd) w_read(A1, s, 1);
        taps(A0, cur, 0);
        if (rd) w_read(A2, s, 2);
        if (rd && s + 1 < total) pix_read(nxt, s + 1);
        taps(A1, cur, 1);
        if (rd && s + 1 < total) w_read(A0, s + 1, 0);
        taps(A2, cur, 2);
        if (ld && s + 2 < total) w_commit(s + 2);
        if (ld && kx == 1 && (q + 1) * 3 < total) h_commit(q + 1);
        if (!(mode & 4)) __syncthreads();
    };
    // tiles outside, stage pairs inside (nst is even): the accumulators are touched by nothing but MFMAs inside the pair loop -
    // with the store under a condition inside it hipcc moved all 128 of them between AGPRs and VGPRs at every stage.
    // The fragment sets alternate with compile-time names (a runtime index would put them in scratch).
    int s = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        for (int pr = 0; pr < nst / 2; ++pr, s += 2) {
            stage(P0, P1, s);
            stage(P1, P0, s + 1);
        }
        {
            const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
            const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, img = t2 / tiles_y;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                bf16_t* dst = y + (((long long)img * H + ty * 16 + 8 * wm + r) * W + tx * 16 + n) * 128 + 64 * wn + 4 * kg;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const acc_t c = acc[r][nb];
                    const bf16_t o0 = (bf16_t)c[0], o1 = (bf16_t)c[1], o2 = (bf16_t)c[2], o3 = (bf16_t)c[3];
                    uint2 qv;
                    qv.x = (unsigned)__builtin_bit_cast(unsigned short, o0) | ((unsigned)__builtin_bit_cast(unsigned short, o1) << 16);
                    qv.y = (unsigned)__builtin_bit_cast(unsigned short, o2) | ((unsigned)__builtin_bit_cast(unsigned short, o3) << 16);
                    *reinterpret_cast<uint2*>(dst + 16 * nb) = qv;
                    acc[r][nb] = (acc_t){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
    if (stamps && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = (long long)__builtin_readcyclecounter() - t_begin;
        stamps[2 * blockIdx.x + 1] = total;
    }
}
}  // namespace

extern "C" int c3w4_launch(const void* x, const void* wpk, void* y, int N, int H, int W, int Cin, int blocks, long long* stamps,
                           void* stream, int mode) {
    if (H % 16 || W % 16 || Cin % 64 || N <= 0) return 1;
    const int smem = 2 * kHaloBytes + 3 * kStageBytes;      // 144 000 B
    static bool set = false;
    if (!set) {
        if (hipFuncSetAttribute((const void*)c3w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return 2;
        set = true;
    }
    c3w4_kernel<<<blocks, kThreads, smem, (hipStream_t)stream>>>((const bf16_t*)x, (const bf16_t*)wpk, (bf16_t*)y, N, H, W, Cin, stamps, mode);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
