// Round 6: a stand-alone model (and numerics check) of a depthwise 3x3 stride-1 forward that is built like the dense 3x3 kernel instead of
// like a register sliding window, BEFORE it is built into the library. Why: the shipped walk (csrc/dwconv.hip) is bound by its ~115
// vector instructions per row step (4.96 TB/s where torch.add moves 7.24 on the same tensors, profiles/r05_power_by_kernel.txt) and
// by the size and order of its 8-byte requests (its memory shape alone tops out at 0.50-0.56 of 8 TB/s cold, a 16-byte grid-stride
// copy of the same bytes reaches 0.66: profiles/r04_walk_ceiling.txt). Here:
//   * a persistent 8-wave block stages the 18 x 18-pixel halo of a 16 x 16-pixel x 64-channel tile in LDS with 16-byte coalesced loads
//     (every input byte fetched once per tile + the halo ring), the producer's batch-norm affine + ReLU6 applied ONCE per element on the
//     way in (the walk activates every element twice);
//   * the nine taps are multiplied on the MATRIX unit, used as a wide per-channel multiplier: D^T[16 co][16 px] += Wd[16 co][32 k] x
//     X^T[32 k][16 px] per tap, k = 16 channels x {hi, lo}: Wd holds diag(bf16_hi(w[tap])) | diag(bf16_lo(w[tap])) (the f32 depthwise
//     weights as two bf16 terms: products exact to 2^-17, f32 accumulation), both k halves read the same 16 B of the halo pixel. 15 / 16 of
//     the matrix unit's multiplies are zeros - it is the vector pipe that is relieved: no unpack, no 9 x 4 FMAs per output vector, and the
//     kernel stays priced against HBM (|X| + |Y|), never against the MFMA peak;
//   * the tile leaves through a bf16 image in LDS as whole 128-byte pixel rows; batch-norm statistics from the f32 accumulators.
// Gate: >= 15 % less time than the shipped forward on the large stride-1 layers, cold - else stop. MEASURED (profiles/r06_dwm_ceiling.txt): 79-82 us against the
// shipped 71 on 128ch @128^2 (plain 64.8 against 67.3): not met, not built - the tile's own work, not memory, bounds it (the knock-outs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dwm_ceiling.hip -o tools/build/dwm_ceiling && tools/build/dwm_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#define LDS __attribute__((address_space(3)))
typedef LDS unsigned char* lds_p;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 512;
constexpr int kHW = 18, kNPix = kHW * kHW;        // halo
constexpr int kRS = 128;                           // bytes per halo / image pixel (64 channels)
constexpr int kHalo = kNPix * kRS;                 // 41 472
constexpr int kImg = 256 * kRS;                    // 32 768
constexpr int kTab = 2 * 64 * 4;                   // scale, shift of the tile's 64 channels
constexpr int kRed = 8 * 32 * 4;                   // per wave: 16 sums + 16 sums of squares
constexpr int kLds = kHalo + kImg + kTab + kRed;   // 75 776: two blocks per CU

struct DwmParams {
    const unsigned short* x; unsigned short* y; const float* w;      // x, y [N,H,W,C] bf16; w [9][C] f32
    const float* scale; const float* shift;                          // producer affine (ReLU6 on load) or null
    float* part;                                                     // [grid / ncg][2][C] or null
    int N, H, W, C, tiles_x, tiles_y, ncg, total;
};

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_t));
}
__device__ __forceinline__ unsigned short bf16_bits(float f) { return (unsigned short)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <bool AFF, bool STATS, int BPC, int KO = 0>     // KO (attribution only): 1 = no MFMAs / fragment reads, 2 = no global loads after the first tile, 3 = no stores
__global__ __launch_bounds__(kThreads, 2 * BPC) void dwm_fwd(const DwmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const lds_p L = (lds_p)smem;
    const lds_p IMG = L + kHalo;
    LDS float* tab = (LDS float*)(L + kHalo + kImg);
    LDS float* red = (LDS float*)(L + kHalo + kImg + kTab);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int g = wave & 3, ph = wave >> 2;            // 16-channel group of the tile's 64, pixel half (rows 8 ph .. + 7)
    int t = blockIdx.x;
    if (t >= p.total) return;
    const int cg = t % p.ncg;                          // (the grid is a multiple of ncg: a block keeps its 64-channel group)
    const int cbase = cg * 64;
    // ---- the nine weight fragments of this wave's 16 channels: lane (co = l15, lq): k = 8 lq + j <-> (part lq >> 1, channel 8 (lq & 1) + j);
    // the one nonzero sits at j = co & 7 of the lanes with (lq & 1) == (co >> 3)
    // (held as ONE 32-bit value per tap - the bf16 term in its half of the dword - and four dword masks: the fragment is formed per tap,
    //  four v_and for eight MFMAs; nine whole fragments are 36 registers of mostly zeros)
    unsigned wval[9], wmask[4];
    {
        const bool mine = (lq & 1) == (l15 >> 3);
        const int jd = (l15 & 7) >> 1, half = l15 & 1;
#pragma unroll
        for (int d = 0; d < 4; ++d) wmask[d] = (mine && d == jd) ? 0xffffffffu : 0u;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float wv = p.w[tap * p.C + cbase + g * 16 + l15];
            const unsigned short hi = bf16_bits(wv);
            const float rest = wv - __uint_as_float((unsigned)hi << 16);
            const unsigned bits = (lq >> 1) ? bf16_bits(rest) : hi;
            wval[tap] = bits << (16 * half);
        }
    }
    if (AFF) {
        for (int i = tid; i < 64; i += kThreads) { tab[i] = p.scale[cbase + i]; tab[64 + i] = p.shift[cbase + i]; }
    }
    // ---- halo staging: thread -> 16-byte slot tid % 8 of halo pixels q54 + 54 i, i = 0..5 (threads 432..511 repeat 352..431)
    const int slot = tid & 7, q64 = tid >> 3;
    const int q54 = q64 < 54 ? q64 : q64 - 10;
    const int qy = q54 / 18, qx = q54 - qy * 18;
    u32x4_t av[6];
    unsigned okmask = 0;
    auto a_load = [&](int tile) __attribute__((always_inline)) {
        const int pos = tile / p.ncg;
        const int tx = pos % p.tiles_x, r1 = pos / p.tiles_x, ty = r1 % p.tiles_y, img = r1 / p.tiles_y;
        const int ix = tx * 16 + qx - 1;
        const bool okx = (unsigned)ix < (unsigned)p.W;
        const unsigned short* xb = p.x + ((size_t)img * p.H * p.W) * p.C + cbase + slot * 8;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int iy = ty * 16 + qy + 3 * i - 1;
            const bool ok = okx && (unsigned)iy < (unsigned)p.H;
            okmask |= (ok ? 1u : 0u) << i;
            av[i] = *reinterpret_cast<const u32x4_t*>(xb + ((size_t)min(max(iy, 0), p.H - 1) * p.W + min(max(ix, 0), p.W - 1)) * p.C);
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
        float sc[8], sh[8];
        if (AFF) {
            const f32x4_t s0 = *(const LDS f32x4_t*)(tab + slot * 8), s1 = *(const LDS f32x4_t*)(tab + slot * 8 + 4);
            const f32x4_t h0 = *(const LDS f32x4_t*)(tab + 64 + slot * 8), h1 = *(const LDS f32x4_t*)(tab + 64 + slot * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { sc[j] = s0[j]; sc[4 + j] = s1[j]; sh[j] = h0[j]; sh[4 + j] = h1[j]; }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            u32x4_t o;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                unsigned u = av[i][d];
                if (AFF) {
                    float a = __uint_as_float(u << 16), b = __uint_as_float(u & 0xffff0000u);
                    a = __builtin_amdgcn_fmed3f(__builtin_fmaf(a, sc[2 * d], sh[2 * d]), 0.f, 6.f);
                    b = __builtin_amdgcn_fmed3f(__builtin_fmaf(b, sc[2 * d + 1], sh[2 * d + 1]), 0.f, 6.f);
                    u = pack_bf16x2(a, b);
                }
                o[d] = ((okmask >> i) & 1u) ? u : 0u;
            }
            *(LDS u32x4_t*)(L + (q54 + 54 * i) * kRS + ((slot ^ (qx & 7)) << 4)) = o;       // (pixels q54 + 54 i share the column qx)
        }
    };
    // per-lane fragment bases, one per tap: pixel (8 ph + r + dy, l15 + dx), slot 2 g + (lq & 1) (both k halves read the same 16 bytes)
    int abase[3];       // (per kernel column dx; the kernel row dy and the output row r are immediate offsets)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int hx = l15 + dx;
        abase[dx] = ((ph * 8) * kHW + hx) * kRS + (((2 * g + (lq & 1)) ^ (hx & 7)) << 4);
    }
    // image write: pixel (8 ph + r, l15), this lane's channels 16 g + 4 lq .. + 3 (8 bytes)
    const int iw = ((ph * 8) * 16 + l15) * kRS + (((2 * g + (lq >> 1)) ^ (l15 & 7)) << 4) + (lq & 1) * 8;
    // copy-out: 2 048 pieces of 16 bytes, 4 per thread: pixel tid / 8 + 64 k, slot tid % 8
    const int cpx = tid >> 3, cslot = tid & 7;
    f32x4_t ssum = {0.f, 0.f, 0.f, 0.f}, qsum = {0.f, 0.f, 0.f, 0.f};

    a_load(t);
    if (AFF) __syncthreads();
    for (; t < p.total; t += gridDim.x) {
        commit();
        __syncthreads();                                   // halo complete; the previous tile's copy-out has read the image
        const int tn = t + (int)gridDim.x;
        if (tn < p.total && KO != 2) a_load(tn);
        f32x4_t acc[8];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            // the three taps of kernel column dx; halo row h of that column feeds output rows h, h - 1, h - 2 at dy = 0, 1, 2: ten
            // fragment reads for 24 MFMAs
            x8 wf[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const unsigned wv = wval[dy * 3 + dx];
                const u32x4_t wu = {wv & wmask[0], wv & wmask[1], wv & wmask[2], wv & wmask[3]};
                wf[dy] = __builtin_bit_cast(x8, wu);
            }
#pragma unroll
            for (int h = 0; h < 10; ++h) {
                if (KO == 1) continue;
                const x8 a = *(const LDS x8*)(L + abase[dx] + h * (kHW * kRS));
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = h - dy;
                    if (r >= 0 && r < 8)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dy], a, (dx == 0 && dy == 0) ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : acc[r], 0, 0, 0);
                }
            }
        }
        if (KO == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = (f32x4_t){(float)wval[r], 0.f, 0.f, 0.f};
        }
        const int pos = t / p.ncg;
        const int tx = pos % p.tiles_x, r1 = pos / p.tiles_x, ty = r1 % p.tiles_y, img = r1 / p.tiles_y;
        const bool okx = tx * 16 + l15 < p.W;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            f32x4_t v = acc[r];
            if (STATS) {
                const bool ok = okx && ty * 16 + ph * 8 + r < p.H;         // (pixels past the image do not count)
                if (!ok) v = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                ssum += v; qsum += v * v;
            }
            *(LDS u32x2_t*)(IMG + iw + r * (16 * kRS)) = (u32x2_t){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
        __syncthreads();                                   // image complete; every wave has read the halo for the last time
        unsigned short* yb = p.y + ((size_t)img * p.H * p.W) * p.C + cbase + cslot * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int px = cpx + 64 * k, oy = ty * 16 + (px >> 4), ox = tx * 16 + (px & 15);
            const u32x4_t v = *(const LDS u32x4_t*)(IMG + px * kRS + ((cslot ^ (px & 7)) << 4));
            if (oy < p.H && ox < p.W && (KO != 3 || v[0] == 0x12345678u)) *reinterpret_cast<u32x4_t*>(yb + ((size_t)oy * p.W + ox) * p.C) = v;
        }
    }
    if (STATS && p.part != nullptr) {
        // sums over this lane's pixels -> over the 16 pixel lanes -> over the two pixel halves; one slab row per group of ncg blocks
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, 64); qsum[j] += __shfl_xor(qsum[j], o, 64); }
        }
        __syncthreads();
        if (l15 == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { red[wave * 32 + lq * 4 + j] = ssum[j]; red[wave * 32 + 16 + lq * 4 + j] = qsum[j]; }
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63, gg = c >> 4, cl = c & 15;
            const float v = red[gg * 32 + which * 16 + cl] + red[(gg + 4) * 32 + which * 16 + cl];
            p.part[((size_t)(blockIdx.x / p.ncg) * 2 + which) * p.C + cbase + c] = v;
        }
    }
}

// ---- the shipped walk's memory footprint for comparison is timed by tools/bench_dw_cold.py; here: this kernel, cold (rotating sets)
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

template <bool AFF, bool STATS, int BPC, int KO = 0>
static void launch(const DwmParams& p, int grid) {
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)dwm_fwd<AFF, STATS, BPC, KO>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds)); set = true; }
    dwm_fwd<AFF, STATS, BPC, KO><<<grid, kThreads, kLds>>>(p);
}

static int check_small() {
    const int N = 2, H = 37, W = 21, C = 128;          // ragged tiles on both axes, two channel groups
    std::vector<unsigned short> x((size_t)N * H * W * C);
    std::vector<float> w(9 * C), sc(C), sh(C);
    unsigned s = 7u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f * 2.f - 1.f; };
    for (auto& v : x) v = f2bf(rnd() * 3.f);
    for (auto& v : w) v = rnd() * 0.5f;
    for (int c = 0; c < C; ++c) { sc[c] = 0.5f + 0.5f * (rnd() + 1.f); sh[c] = rnd(); }
    unsigned short *dx, *dy; float *dw, *dsc, *dsh, *dpart;
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16, ncg = C / 64, total = N * tiles_x * tiles_y * ncg;
    const int grid = std::min(total, 8);               // few blocks: each walks several tiles
    CK(hipMalloc(&dx, x.size() * 2)); CK(hipMalloc(&dy, x.size() * 2)); CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dsc, C * 4)); CK(hipMalloc(&dsh, C * 4));
    CK(hipMalloc(&dpart, (size_t)(grid / ncg) * 2 * C * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsc, sc.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, sh.data(), C * 4, hipMemcpyHostToDevice));
    DwmParams p = {dx, dy, dw, dsc, dsh, dpart, N, H, W, C, tiles_x, tiles_y, ncg, total};
    launch<true, true, 1>(p, grid);
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> y(x.size());
    std::vector<float> part((size_t)(grid / ncg) * 2 * C);
    CK(hipMemcpy(y.data(), dy, y.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(part.data(), dpart, part.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    std::vector<double> rs(C, 0.0), rq(C, 0.0);
    int bad = 0;
    for (int n = 0; n < N; ++n) for (int oy = 0; oy < H; ++oy) for (int ox = 0; ox < W; ++ox) for (int c = 0; c < C; ++c) {
        double acc = 0;
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            float a = bf2f(x[(((size_t)n * H + iy) * W + ix) * C + c]);
            a = fminf(fmaxf(fmaf(a, sc[c], sh[c]), 0.f), 6.f);
            a = bf2f(f2bf(a));                             // the activated operand is stored as bf16 in the halo image
            acc += (double)a * (double)w[(ky * 3 + kx) * C + c];
        }
        rs[c] += acc; rq[c] += acc * acc;
        const float got = bf2f(y[(((size_t)n * H + oy) * W + ox) * C + c]);
        const double err = fabs(got - acc);
        maxerr = std::max(maxerr, err); maxref = std::max(maxref, fabs(acc));
        if (err > 0.004 * fabs(acc) + 1e-3) ++bad;        // one bf16 ulp of the output
    }
    double serr = 0;
    for (int c = 0; c < C; ++c) {
        double s0 = 0, q0 = 0;
        for (int r = 0; r < grid / ncg; ++r) { s0 += part[((size_t)r * 2) * C + c]; q0 += part[((size_t)r * 2 + 1) * C + c]; }
        serr = std::max(serr, fabs(s0 - rs[c]) / (fabs(rs[c]) + 1.0));
        serr = std::max(serr, fabs(q0 - rq[c]) / (fabs(rq[c]) + 1.0));
    }
    printf("# numerics, [%d,%d,%d,%d] with affine + ReLU6 on load and statistics against a double-precision host loop: max |err| %.4g (max |ref| %.3g), "
           "%d of %zu outputs beyond one bf16 ulp; statistics rel err %.2e\n", N, H, W, C, maxerr, maxref, bad, y.size(), serr);
    CK(hipFree(dx)); CK(hipFree(dy)); CK(hipFree(dw)); CK(hipFree(dsc)); CK(hipFree(dsh)); CK(hipFree(dpart));
    return bad == 0 && serr < 1e-4 ? 0 : 1;
}

template <bool AFF, bool STATS, int BPC, int KO = 0>
static void bench_shape(const char* name, int N, int H, int W, int C, int cus) {
    const size_t elems = (size_t)N * H * W * C;
    const int sets = std::max(2, (int)((size_t)1400 * 1000 * 1000 / (elems * 4)) + 1);      // x + y per set; > 1.4 GB in rotation: cold
    std::vector<unsigned short*> xs(sets), ys(sets);
    std::vector<unsigned short> h(elems);
    unsigned s = 99u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = f2bf((float)(s >> 8) / 16777216.f * 4.f - 2.f); }
    for (int i = 0; i < sets; ++i) { CK(hipMalloc(&xs[i], elems * 2)); CK(hipMalloc(&ys[i], elems * 2)); CK(hipMemcpy(xs[i], h.data(), elems * 2, hipMemcpyHostToDevice)); }
    std::vector<float> w(9 * C, 0.1f), sc(C, 1.01f), sh(C, 0.02f);
    float *dw, *dsc, *dsh, *dpart;
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16, ncg = C / 64, total = N * tiles_x * tiles_y * ncg;
    int grid = cus * BPC; grid -= grid % ncg; if (grid > total) grid = total - total % ncg;
    CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dsc, C * 4)); CK(hipMalloc(&dsh, C * 4)); CK(hipMalloc(&dpart, (size_t)(grid / ncg + 1) * 2 * C * 4));
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, sh.data(), C * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](bool cold, int iters) {
        std::vector<float> ts;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) {
                const int k = cold ? (rep * iters + i) % sets : 0;
                DwmParams p = {xs[k], ys[k], dw, AFF ? dsc : nullptr, AFF ? dsh : nullptr, STATS ? dpart : nullptr, N, H, W, C, tiles_x, tiles_y, ncg, total};
                launch<AFF, STATS, BPC, KO>(p, grid);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms * 1e3f / iters);
        }
        std::sort(ts.begin(), ts.end());
        return ts[ts.size() / 2];
    };
    run(true, sets);
    const float cold = run(true, 2 * sets), warm = run(false, 20);
    const double bytes = 2.0 * elems * 2;
    if (KO) printf("[knock-out %d: %s] ", KO, KO == 1 ? "no MFMAs / fragment reads" : KO == 2 ? "no global loads" : "no global stores");
    printf("%-28s blocks/CU %d %s%s: cold %6.1f us (%.3f of 8 TB/s) / same buffers %6.1f us (%.3f)   [%d sets of %.0f MB, %d tiles on %d blocks]\n", name, BPC,
           AFF ? "affine+ReLU6 " : "plain ", STATS ? "+stats" : "", cold, bytes / cold / 8e6, warm, bytes / warm / 8e6, sets, bytes / 1e6, total, grid);
    fflush(stdout);
    for (int i = 0; i < sets; ++i) { CK(hipFree(xs[i])); CK(hipFree(ys[i])); }
    CK(hipFree(dw)); CK(hipFree(dsc)); CK(hipFree(dsh)); CK(hipFree(dpart));
}

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("# %d CUs; depthwise 3x3 stride 1 forward as LDS-staged 16 x 16 x 64 tiles, the taps on the matrix unit (diagonal weights, hi + lo)\n", cus);
    if (check_small()) { printf("NUMERICS FAILED\n"); return 1; }
    bench_shape<true, true, 1>("128ch @128x128 x32 (dw3)", 32, 128, 128, 128, cus);
    bench_shape<true, true, 2>("128ch @128x128 x32 (dw3)", 32, 128, 128, 128, cus);
    bench_shape<false, false, 1>("128ch @128x128 x32 (dw3)", 32, 128, 128, 128, cus);
    bench_shape<false, false, 2>("128ch @128x128 x32 (dw3)", 32, 128, 128, 128, cus);
    bench_shape<true, true, 2, 1>("128ch @128x128 x32 (dw3)", 32, 128, 128, 128, cus);
    bench_shape<true, true, 2, 2>("128ch @128x128 x32 (dw3)", 32, 128, 128, 128, cus);
    bench_shape<true, true, 2, 3>("128ch @128x128 x32 (dw3)", 32, 128, 128, 128, cus);
    bench_shape<true, true, 2>("256ch @64x64 x32 (dw5)", 32, 64, 64, 256, cus);
    bench_shape<true, true, 2>("512ch @32x32 x32 (dw7-11)", 32, 32, 32, 512, cus);
    bench_shape<true, true, 2>("1024ch @16x16 x32 (dw13)", 32, 16, 16, 1024, cus);
    return 0;
}
