// Round 6: a stand-alone model of a 3x3 stage that CUTS THE MATRIX WORK with a Winograd transform, before any
// kernel is built on it.
//
// Which transform. F(2x2, 3x3) - 16 points per 4 x 4 patch, 2.25 x fewer multiplies - was priced first and is bound by the LDS on this
// chip in every decomposition that fits the register file: per transform point the product is [64 patches x 32 ci] x [32 ci x 16 co]
// per wave and MFMA, its A fragment read from LDS for ONE MFMA (a wave cannot hold the 16 points' accumulators of more than 64 patches
// x 16 channels: 16 x 4 x 4 registers = the whole file), so 8 waves x 16 points x 4 k-steps x 4 ds_read_b128 x 4 LDS cycles = 8 192 LDS
// cycles per 16 x 16-pixel tile against 8 192 matrix cycles per SIMD, before the 256 KB per tile of transformed patches are WRITTEN
// (another 3 300). The direct kernel spends 0.375 fragment reads per MFMA (a halo row feeds three taps); the 2-D transform has no tap
// reuse left - the taps are the points. The output transform (36 adds per 16 accumulator values when the points are walked outermost)
// would come on top.
// So this model takes the 1-D form, F(2, 3) along x, direct along y: 4 points per 4 x 1 patch -> 2 outputs, 1.5 x fewer multiplies.
//   * the point accumulators of a wave are 2 x the output tile, not 4 x: tile 8 rows x 32 columns x 128 output channels, the eight waves
//     split the channels as conv3x3_cs.hip's do (16 each), 8 rows x 4 points x 4 = 128 accumulator registers per lane, held through
//     the whole contraction - the OUTPUT transform (out0 = M0 + M1 + M2, out1 = M1 - M2 - M3) runs once per tile: 128 adds per lane;
//   * an A fragment is [16 x-patches of one halo row] x [32 ci] of ONE point j and still feeds the three taps ky = 0..2: 10 reads for 24
//     MFMAs per stage (j, k-step) = 0.42 per MFMA; 16 stages per tile, 384 MFMAs per wave (576 direct);
//   * the weights G g G^T (along x) are transformed when they are packed: 12 fragments per k-step instead of 9, from L2 into registers
//     a stage ahead exactly as now;
//   * the INPUT transform (V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3 per 4 x 1 patch and channel) is where the saved matrix
//     cycles are paid for: gfx950 has no packed bf16 add, so a transformed dword costs 2 unpacks per input + 2 adds + 1 pack: 100 vector
//     instructions per thread and 32-channel chunk (the direct kernel's commit: ~30), and the transformed image is 4 / 2 = 2 x the halo
//     image: 40 KB per 32 channels, so the block meets every k-step (4 stages) instead of every two.
// The model runs that stage loop with everything the kernel would carry - the loads of the next chunk's patches from a 134 MB tensor,
// the transform, the LDS commit, weights from L2, one barrier per chunk, and per tile the output transform, the tile's image in LDS and
// its copy-out in 256-byte pixel rows - on random data, and prints ALGORITHMIC TFLOP/s (the direct convolution's 2 x 256 x 128 x 1152
// per tile). Numbers are not checked (it is a model: the operands are random, the index arithmetic is the real one).
// Gate (VERDICT): >= 1 550 algorithmic TFLOP/s, or stop. MEASURED 1 423-1 496 (profiles/r06_stage3_ceiling.txt): not met, no kernel built. For scale: tools/stage2_ceiling.hip's model of the SHIPPED stage reaches 1 620
// without epilogue and tile walk, the shipped kernel 1 200.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stage3_ceiling.hip -o tools/build/stage3_ceiling && tools/build/stage3_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 x8 __attribute__((ext_vector_type(8)));
typedef float acc_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#define LDS __attribute__((address_space(3)))
typedef LDS unsigned char* lds_p;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kStage = 3 * 128 * 64;          // the weights of a stage (j, k-step): [ky 3][co 128][64 B]
constexpr int kWeights = 16 * kStage;         // [k-step 4][j 4]
constexpr int kPlane = 1024;                  // transformed patches of one (halo row, point): [16 x-patches][64 B = 32 channels]
constexpr int kVBuf = 10 * 4 * kPlane;        // 40 960: [halo row 10][point 4]
constexpr int kImg = 256 * 256;               // the tile's output image: [256 pixels][128 channels] bf16
constexpr int kLds = 2 * kVBuf + kImg;        // 147 456
constexpr int kRing = 6;
constexpr int kW = 128, kH = 128, kC = 128;   // the tensor: [32][128][128][128] bf16 (134 MB), tiles of 8 x 32 pixels
constexpr int kTilesX = kW / 32, kTilesY = kH / 8;

enum { BARE = 0, WEIGHTS = 1, CHUNK = 2, FULL = 3 };   // + weight loads, + chunk barrier, + input transform / epilogue
// parts of FULL, for the attribution: XF = loads + input transform + commit; EPI = 0 none, 1 = output transform + LDS image + row stores,
// 2 = output transform + 8-byte stores straight from the accumulators (no image, no barrier: 32-byte segments per pixel)

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_t));
}

template <int MODE, bool AFF, bool XF = true, int EPI = 1>
__global__ void __launch_bounds__(512, 1) wino_loop(const unsigned char* __restrict__ w, const unsigned char* __restrict__ in,
                                                     unsigned char* __restrict__ out, int tiles_total, long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const lds_p L = (lds_p)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    // random operands in both transformed images (the first chunk reads them before anything is committed)
    for (int i = tid; i < 2 * kVBuf / 16; i += 512) *(LDS u32x4_t*)(L + i * 16) = *reinterpret_cast<const u32x4_t*>(in + (size_t)i * 16);
    __syncthreads();
    // fragment read: lane (x-patch l15, 8-channel group lq) inside a (row, point) plane; 16-byte slots XOR-swizzled so that each
    // 16-lane group of ds_read_b128 ({0-3, 12-15, 20-27}, ...) covers the 256-byte bank row once
    const int a_lane = l15 * 64 + ((lq ^ ((l15 >> 3) << 1)) << 4);
    // weights: a wave's fragment (ky, its 16 channels) is one contiguous KB of the stage
    const unsigned lane_w = (unsigned)(wave * 1024 + lane * 16);
    auto b_load = [&](x8 (&dst)[3], int stage) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, kWeights, 0x00020000);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
            dst[ky] = __builtin_bit_cast(x8, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rs, lane_w, stage * kStage + ky * 8192, 0));
    };
    // ---- input transform. A chunk's transformed image = 10 rows x 16 patches x 4 groups of 8 channels = 640 units of (4 pixels in, 4
    // points out, 16 bytes each); a thread owns unit `tid` whole and one dword (two channels) of unit 512 + tid / 4
    const int u_h = tid >> 6, u_t = (tid >> 2) & 15, u_g = tid & 3;             // rows 0..7
    const int e_u = 512 + (tid >> 2), e_d = tid & 3;
    const int e_h = e_u >> 6, e_t = (e_u >> 2) & 15, e_g = e_u & 3;            // rows 8, 9
    const int v_dst = (u_h * 4) * kPlane + u_t * 64 + ((u_g ^ ((u_t >> 3) << 1)) << 4);        // + j * kPlane
    const int v_dst_e = (e_h * 4) * kPlane + e_t * 64 + ((e_g ^ ((e_t >> 3) << 1)) << 4) + e_d * 4;
    uint4 px[4];         // the four pixels of the own unit (8 channels each)
    unsigned pe[4];      // the extra dword's four pixels
    unsigned vo[4][4];   // transformed: [point][dword]
    unsigned ve[4];
    auto in_load = [&](int tile, int ks) __attribute__((always_inline)) {
        const int img = tile / (kTilesX * kTilesY), rem = tile - img * (kTilesX * kTilesY);
        const int ty = rem / kTilesX, tx = rem - ty * kTilesX;
        // (halo rows / columns past the image: clamped - the model does not zero them)
        const unsigned char* base = in + ((size_t)img * kH * kW) * (kC * 2) + ks * 64;
        const int y0 = min(max(ty * 8 - 1 + u_h, 0), kH - 1), y1 = min(max(ty * 8 - 1 + e_h, 0), kH - 1);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int x0 = min(max(tx * 32 - 1 + 2 * u_t + p, 0), kW - 1), x1 = min(max(tx * 32 - 1 + 2 * e_t + p, 0), kW - 1);
            px[p] = *reinterpret_cast<const uint4*>(base + ((size_t)(y0 * kW + x0) * kC + u_g * 8) * 2);
            pe[p] = *reinterpret_cast<const unsigned*>(base + ((size_t)(y1 * kW + x1) * kC + e_g * 8 + e_d * 2) * 2);
        }
    };
    // one dword (two channels) of the four points from the same dword of the four pixels; sub-steps s = 0..6 so that a stage can
    // spread the 20 vector instructions over its MFMA groups
    float f[4][2];
    auto xform = [&](float (&f)[4][2], const unsigned (&u)[4], unsigned (&o)[4], const int s) __attribute__((always_inline)) {
        if (s < 4) {
            f[s][0] = __uint_as_float(u[s] << 16); f[s][1] = __uint_as_float(u[s] & 0xffff0000u);
            if constexpr (AFF) {      // the producer's batch-norm affine + ReLU on load (own pixels only in a kernel: see the header)
                if (s < 2) {
                    f[s][0] = fmaxf(__builtin_fmaf(f[s][0], 1.01f, 0.01f), 0.f); f[s][1] = fmaxf(__builtin_fmaf(f[s][1], 0.99f, -0.01f), 0.f);
                }
            }
        } else if (s == 4) {
            o[0] = pack_bf16x2(f[0][0] - f[2][0], f[0][1] - f[2][1]);
            o[1] = pack_bf16x2(f[1][0] + f[2][0], f[1][1] + f[2][1]);
        } else if (s == 5) {
            o[2] = pack_bf16x2(f[2][0] - f[1][0], f[2][1] - f[1][1]);
            o[3] = pack_bf16x2(f[1][0] - f[3][0], f[1][1] - f[3][1]);
        }
    };

    acc_t acc[4][8];       // [point][output row]
    x8 b[2][3];
    if (MODE >= WEIGHTS) b_load(b[0], 0);
    else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) b[0][ky] = *reinterpret_cast<const x8*>(w + lane_w + ky * 8192);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) b[1][ky] = b[0][ky];
    }
    int tile = blockIdx.x;
    if (MODE >= FULL && XF) in_load(tile, 0);
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int cc = 0;            // running chunk counter: the chunk reads transformed image cc & 1
#pragma unroll 1
    for (; tile < tiles_total; tile += gridDim.x) {
        auto chunk = [&](const int ks, auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const lds_p vb = L + (cc & 1) * kVBuf + a_lane;
            const lds_p vn = L + ((cc + 1) & 1) * kVBuf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE >= WEIGHTS) b_load(b[(j + 1) & 1], (ks * 4 + j + 1) & 15);
                x8 a[10];
#pragma unroll
                for (int h = 0; h < kRing; ++h) a[h] = *(const LDS x8*)(vb + (h * 4 + j) * kPlane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 10; ++h) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int r = h - ky;
                        if (r >= 0 && r < 8)
                            acc[j][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j & 1][ky], a[h], (FIRST && ky == 0) ? (acc_t){0.f, 0.f, 0.f, 0.f} : acc[j][r], 0, 0, 0);
                    }
                    if (h + kRing < 10) a[h + kRing] = *(const LDS x8*)(vb + ((h + kRing) * 4 + j) * kPlane);
                    if (MODE >= FULL && XF) {
                        // dword j of the own unit in sub-steps at h = 1..6, the extra dword's at h = 8 of stage 0
                        if (h >= 1 && h <= 6) {
                            const unsigned u[4] = {j == 0 ? px[0].x : j == 1 ? px[0].y : j == 2 ? px[0].z : px[0].w,
                                                   j == 0 ? px[1].x : j == 1 ? px[1].y : j == 2 ? px[1].z : px[1].w,
                                                   j == 0 ? px[2].x : j == 1 ? px[2].y : j == 2 ? px[2].z : px[2].w,
                                                   j == 0 ? px[3].x : j == 1 ? px[3].y : j == 2 ? px[3].z : px[3].w};
                            unsigned o[4] = {vo[0][j], vo[1][j], vo[2][j], vo[3][j]};
                            xform(f, u, o, h - 1);
                            if (h >= 5) { vo[0][j] = o[0]; vo[1][j] = o[1]; vo[2][j] = o[2]; vo[3][j] = o[3]; }
                        }
                        if (j == 0 && h == 8) {      // (the extra dword whole, in one group: its unpacked values do not outlive the stage)
                            float fe[4][2];
#pragma unroll
                            for (int s2 = 0; s2 < 6; ++s2) xform(fe, pe, ve, s2);
                        }
                        if (j == 3 && h == 7) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) *(LDS u32x4_t*)(vn + v_dst + q * kPlane) = (u32x4_t){vo[q][0], vo[q][1], vo[q][2], vo[q][3]};
                        }
                        if (j == 3 && h == 8) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) *(LDS unsigned*)(vn + v_dst_e + q * kPlane) = ve[q];
                        }
                        // the patches of the chunk after the next, requested when this chunk's have been consumed
                        if (j == 3 && h == 9) {
                            const int nks = (ks + 2) & 3;
                            const int nt = tile + (int)gridDim.x < tiles_total ? tile + (int)gridDim.x : tile;     // (never past the tensor)
                            in_load(ks >= 2 ? nt : tile, nks);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE >= CHUNK) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            ++cc;
        };
        chunk(0, std::true_type{});
#pragma unroll 1
        for (int ks = 1; ks < 4; ++ks) chunk(ks, std::false_type{});
        if (MODE >= FULL && EPI == 2) {
            const int img_i = tile / (kTilesX * kTilesY), rem = tile - img_i * (kTilesX * kTilesY);
            const int ty = rem / kTilesX, tx = rem - ty * kTilesX;
            unsigned char* ob = out + (((size_t)img_i * kH + ty * 8) * kW + tx * 32 + 2 * l15) * (kC * 2) + (wave * 16 + lq * 4) * 2;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const acc_t o0 = acc[0][r] + acc[1][r] + acc[2][r], o1 = acc[1][r] - acc[2][r] - acc[3][r];
                *reinterpret_cast<u32x2_t*>(ob + (size_t)r * kW * (kC * 2)) = (u32x2_t){pack_bf16x2(o0[0], o0[1]), pack_bf16x2(o0[2], o0[3])};
                *reinterpret_cast<u32x2_t*>(ob + (size_t)r * kW * (kC * 2) + kC * 2) = (u32x2_t){pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3])};
            }
        }
        if (MODE >= FULL && EPI == 1) {
            // ---- the tile's epilogue (the kernel would run it under the next tile's first stages): output transform, bf16 image in LDS
            // (pixel (row r, column 2 l15 + e), this wave's channels 4 lq .. + 3: 8 bytes), copy-out in whole 256-byte pixel rows
            const lds_p img = L + 2 * kVBuf;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const acc_t o0 = acc[0][r] + acc[1][r] + acc[2][r], o1 = acc[1][r] - acc[2][r] - acc[3][r];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const acc_t o = e ? o1 : o0;
                    const int pxl = r * 32 + 2 * l15 + e;
                    *(LDS u32x2_t*)(img + pxl * 256 + (((wave * 2 + (lq >> 1)) ^ (pxl & 15)) << 4) + (lq & 1) * 8) =
                        (u32x2_t){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int img_i = tile / (kTilesX * kTilesY), rem = tile - img_i * (kTilesX * kTilesY);
            const int ty = rem / kTilesX, tx = rem - ty * kTilesX;
            unsigned char* ob = out + (((size_t)img_i * kH + ty * 8 + wave) * kW + tx * 32) * (kC * 2);     // this wave: output row `wave`
            const int piece = lane & 15, prow = lane >> 4;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int pxl = wave * 32 + prow + 4 * k;
                const u32x4_t v = *(const LDS u32x4_t*)(img + pxl * 256 + ((piece ^ (pxl & 15)) << 4));
                *reinterpret_cast<u32x4_t*>(ob + (size_t)(prow + 4 * k) * (kC * 2) + piece * 16) = v;
            }
            // (the image is free again when every wave has copied its rows: the next tile's epilogue is four chunk barriers away)
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (MODE < FULL || EPI == 0) {
        acc_t s = (acc_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) s += acc[j][r];
        reinterpret_cast<float*>(out)[(size_t)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
    }
    if (lane == 0) {
        const int wv = blockIdx.x * 8 + wave;
        stamps[2 * wv] = t1 - t0;
        stamps[2 * wv + 1] = r1 - r0;
    }
}

template <int MODE, bool AFF, bool XF = true, int EPI = 1>
static void run(const char* name, const unsigned char* w, const unsigned char* in, unsigned char* out, long long* stamps, int cus, int tiles) {
    CK(hipFuncSetAttribute((const void*)wino_loop<MODE, AFF, XF, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (float total = 0.f; total < 1500.f;) {        // warm: the clock settles under load
        CK(hipEventRecord(e0));
        for (int i = 0; i < 8; ++i) wino_loop<MODE, AFF, XF, EPI><<<cus, 512, kLds>>>(w, in, out, tiles, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    std::vector<float> t;
    for (int r = 0; r < 9; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 8; ++i) wino_loop<MODE, AFF, XF, EPI><<<cus, 512, kLds>>>(w, in, out, tiles, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / 8);
    }
    std::sort(t.begin(), t.end());
    const int nw = cus * 8;
    std::vector<long long> h(2 * nw);
    CK(hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * nw, hipMemcpyDeviceToHost));
    const double tiles_per_block = (double)tiles / cus;
    std::vector<double> cyc(nw), clk(nw);
    for (int i = 0; i < nw; ++i) {
        cyc[i] = (double)h[2 * i] / tiles_per_block;
        clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double flop = 2.0 * 256 * 128 * 1152 * (double)tiles;          // ALGORITHMIC: the direct convolution's
    const double us = t[t.size() / 2] * 1e3;
    const double tf = flop / (us * 1e-6) / 1e12;
    printf("{\"variant\": \"%s\", \"cycles_per_tile\": %.0f, \"matrix_cycles_per_tile\": 12288, \"direct_matrix_cycles_per_tile\": 18432, \"clock_GHz\": %.3f, "
           "\"us_per_launch\": %.1f, \"tiles\": %d, \"algorithmic_TFLOPs\": %.1f, \"frac_of_2500\": %.3f}\n",
           name, cyc[nw / 2], clk[nw / 2], us, tiles, tf, tf / 2500.0);
    fflush(stdout);
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const int tiles = 32 * kTilesX * kTilesY;         // 2 048: the bench layer's [32,128,128,128]
    printf("# %s, %d CUs; F(2,3) along x, tile 8 x 32 pixels x 128 channels, Cin 128: %d tiles\n", p.name, cus, tiles);
    const size_t n_in = (size_t)32 * kH * kW * kC;
    std::vector<unsigned short> h(n_in);
    unsigned s = 12345u;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        float f = (float)(s >> 8) / 16777216.f * 2.f - 1.f;
        unsigned u; memcpy(&u, &f, 4);
        v = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
    unsigned char *in, *out, *w; long long* stamps;
    CK(hipMalloc(&in, n_in * 2)); CK(hipMalloc(&out, n_in * 2)); CK(hipMalloc(&stamps, sizeof(long long) * 2 * cus * 8));
    CK(hipMalloc(&w, kWeights));
    CK(hipMemcpy(in, h.data(), n_in * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), kWeights, hipMemcpyHostToDevice));
    if (argc > 1 && !strcmp(argv[1], "loop")) {     // the full model back to back for ~8 s: board power is sampled beside it (rocm-smi)
        CK(hipFuncSetAttribute((const void*)wino_loop<FULL, false, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
        for (int i = 0; i < 70000; ++i) wino_loop<FULL, false, true, 1><<<cus, 512, kLds>>>(w, in, out, tiles, stamps);
        CK(hipDeviceSynchronize());
        printf("loop done\n");
        return 0;
    }
    run<BARE, false>("bare (MFMAs + fragment reads)", w, in, out, stamps, cus, tiles);
    run<WEIGHTS, false>("+ weights from L2", w, in, out, stamps, cus, tiles);
    run<CHUNK, false>("+ chunk barrier", w, in, out, stamps, cus, tiles);
    run<FULL, false>("full: + input transform, commit, epilogue", w, in, out, stamps, cus, tiles);
    run<FULL, true>("full + affine / ReLU on load", w, in, out, stamps, cus, tiles);
    run<FULL, false>("full (again)", w, in, out, stamps, cus, tiles);
    run<FULL, false, true, 0>("input transform only (no epilogue)", w, in, out, stamps, cus, tiles);
    run<FULL, false, false, 1>("epilogue only (no input transform)", w, in, out, stamps, cus, tiles);
    run<FULL, false, false, 2>("epilogue only, 8-byte stores from the accumulators", w, in, out, stamps, cus, tiles);
    run<FULL, false, true, 2>("full, 8-byte stores from the accumulators", w, in, out, stamps, cus, tiles);
    return 0;
}
