// 3x3 convolution, 16-bit storage, 256-pixel x 128-channel block tiles (conv3x3.hip). Shared with conv_mfma.hip, which
// owns the C ABI entry points (mpn_conv_fwd, mpn_conv_fwd_grouped, the weight packers) and routes eligible layers here.
#pragma once
#include "common.h"

namespace mpn_c3 {

// eligible: 3x3, bf16 / fp16 storage, GEMM K (input channels) a multiple of 64 up to 512, GEMM N (output channels) of 128
__host__ __device__ inline bool eligible(int Kin, int Nout, int taps, int es) {
    return taps == 9 && es == 2 && Kin % 64 == 0 && Kin <= 512 && Nout % 128 == 0;
}
// the 64-channel-tile variant (the two waves of a row group split K): GEMM N an odd multiple of 64
__host__ __device__ inline bool eligible64(int Kin, int Nout, int taps, int es) {
    return taps == 9 && es == 2 && Kin % 64 == 0 && Kin <= 512 && Nout % 128 == 64;
}

// Packed weights, per n-tile of 128 output channels:  [chunk of 64 input channels][kx 3][k-step 2][ky 3][co 128][64 bytes],
// i.e. stages of 24 576 bytes = the three taps of one kernel COLUMN for one 32-channel k-step; the four 16-byte slots of a
// 64-byte row are XOR-swizzled with swz(co) like the tiled kernel's image (the global image is the LDS image).
constexpr int kStageBytes = 3 * 128 * 64;
__host__ __device__ constexpr int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }
// 64-channel-tile variant, per n-tile of 64:  [chunk of 64][kx 3][ky 3][k-step 2][co 64][64 bytes] - the same stage bytes, LDS
// row = k-step * 64 + co, slots swizzled with swz(co).
inline long long packed_bytes(int Kin, int Nout) { return 9ll * Kin * Nout * 2; }
inline long long tile_bytes(int Kin) { return 9ll * Kin * 128 * 2; }
inline long long tile_bytes64(int Kin) { return 9ll * Kin * 64 * 2; }

constexpr int kMaxJobs = 5;
struct Job {
    const void* x;        // [N,H,W,*] pixel stride xs
    const void* wp;       // packed weights (above)
    void* y;              // [N,H,W,*] pixel stride ys
    const float* in_scale;   // producer's batch-norm affine applied on load (NULL: none)
    const float* in_shift;
    float* stats_part;    // [stats_rows(N, H, W, Cout)][2][Cout] partial sums of the rounded outputs (one row per block that has a
                          // tile of the job: mpn_conv_stats_rows), or NULL
    int in_act;
    int N, H, W, Cin, Cout, xs, ys;
    // Data-gradient launches that also reduce for the batch-norm they feed (mpn_conv_bwd_data_bn): y is the gradient w.r.t. the
    // ACTIVATED output of a batch-norm layer whose raw input is bnr_x [N,H,W,*] (pixel stride bnr_xs, Cout channels). The tile
    // is written MASKED (g = y where lo < bnr_x * bnr_scale + bnr_shift < hi, else 0: the activation's derivative) and
    // stats_part receives the partial sums of g and of g * bnr_x (raw: the finalize turns them into sum g * xhat).
    const void* bnr_x;    // NULL: a plain convolution
    const float* bnr_scale;
    const float* bnr_shift;
    int bnr_act, bnr_xs;
#ifdef MPN_DIAG
    unsigned long long* dbg;
#endif
};
inline int blocks_of(const Job& j) {
    return j.N * ((j.H + 15) / 16) * ((j.W + 15) / 16) * ((j.Cout & 127) ? j.Cout / 64 : j.Cout / 128);
}

// one grid over up to kMaxJobs independent layers of the same (Cin, Cout, dtype) - e.g. the pyramid levels of a subnet stage
int launch(const Job* jobs, int njobs, int dtype, hipStream_t st);
// rows of the statistics slab a job of this shape writes (the same alone and in a group); < 0: no device
int stats_rows(int N, int H, int W, int Cout);

}  // namespace mpn_c3

// ---------------------------------------------------------------- shared by the two kernels (conv3x3.hip, conv3x3_cs.hip)
#include <type_traits>
namespace mpn_c3 {
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kThreads = 512;
// The producer's batch-norm affine + activation on eight 16-bit values: two elements per v_pk_fma_f32, rounded back to storage, and
// the activation on the ROUNDED pairs as integers - sign-magnitude formats order like int16 on the non-negative side, so ReLU is
// v_pk_max_i16(x, 0) and the upper clamp v_pk_min_i16(x, 6.0): one instruction per two elements instead of a v_med3_f32 per
// element. Rounding is monotone and keeps 0 and 6 fixed, so act(round(v)) == round(act(v)) bit for bit (a -0.0 becomes +0.0).
// lo2 = 0 (ReLU / ReLU6) or 0x80008000 (no activation: max with -32768 is the identity); RELU6 adds the upper clamp.
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, a), __builtin_bit_cast(s16x2_t, b)));
}
__device__ __forceinline__ unsigned pk_min_i16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(s16x2_t, a), __builtin_bit_cast(s16x2_t, b)));
}
template <typename T> __device__ __forceinline__ constexpr unsigned six_pair() { return sizeof(T) == 2 && std::is_same<T, bf16_t>::value ? 0x40C040C0u : 0x46004600u; }
template <typename T, bool RELU6>
__device__ __forceinline__ void affine_act(Vec16<T>& v, const f32x2_t (&sc)[4], const f32x2_t (&sh)[4], unsigned lo2) {
    float f[8];
    v.unpack(f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2_t r = __builtin_elementwise_fma((f32x2_t){f[2 * j], f[2 * j + 1]}, sc[j], sh[j]);
        f[2 * j] = r[0]; f[2 * j + 1] = r[1];
    }
    v.pack(f);
    unsigned u[4] = {v.raw.x, v.raw.y, v.raw.z, v.raw.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u[j] = pk_max_i16(u[j], lo2);
        if (RELU6) u[j] = pk_min_i16(u[j], six_pair<T>());
    }
    v.raw = make_uint4(u[0], u[1], u[2], u[3]);
}

__device__ __forceinline__ void store4(bf16_t* p, const f32x4_t& v) {
    typedef __bf16 b4_t __attribute__((ext_vector_type(4)));
    const b4_t h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ void store4(half_t* p, const f32x4_t& v) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    const h4_t h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, h);
}

struct Group {
    Job job[kMaxJobs];
    int begin[kMaxJobs + 1];   // first tile of each job; begin[njobs] = number of tiles
    int njobs;
};

// where a tile lives (wave-uniform)
struct Tile {
    int job, ntile, img, oy0, ox0, ty, tx;
};
template <bool N64>
__device__ __forceinline__ Tile tile_of(const Group& g, int w) {
    Tile t;
    t.job = 0;
#pragma unroll
    for (int k = 1; k < kMaxJobs; ++k)
        if (k < g.njobs && w >= g.begin[k]) t.job = k;
    const Job& p = g.job[t.job];
    int b = w - g.begin[t.job];
    const int n_tiles = N64 ? (p.Cout >> 6) : (p.Cout >> 7), tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + 15) >> 4;
    t.ntile = b % n_tiles; b /= n_tiles;
    t.tx = b % tiles_x; b /= tiles_x;
    t.ty = b % tiles_y;
    t.img = b / tiles_y;
    t.oy0 = t.ty * 16; t.ox0 = t.tx * 16;
    return t;
}


// the channel-split kernel (conv3x3_cs.hip): 128-channel tiles, or (n64) 64-channel tiles
int launch_cs(const Group& g, int blocks, int dtype, bool affine, bool bnr, bool n64, hipStream_t st);

}  // namespace mpn_c3
