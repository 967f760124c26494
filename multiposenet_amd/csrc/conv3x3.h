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
