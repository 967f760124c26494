// Target-heatmap rendering: the label producer of the keypoint path
// (reference detector/input_pipeline/heatmap_creation.py:6-118, called per image through tf.py_func at
// keypoints_detector_pipeline.py:86-90).  One batched launch renders [B, h, w, 17] float32 maps:
//
//   out[b, y, x, j] = max(0, max over visible persons p of image b of  float32(g_p[|y-cy|] * g_p[|x-cx|]))
//
// with the reference's arithmetic reproduced step by step (float32 sigma / centre math, float64 window, float32
// rounding of the separable product), so the result is bit-identical to the numpy code; peaks are exactly 1.0.
// HBM-bound: the kernel writes every output byte once (h*w*17*4 B per image) and reads a few hundred bytes.
#include "common.h"

namespace {

constexpr int kParts = 17;
constexpr int kMaxHalf = 13;                 // sigma <= 4  ->  k = ceil(sqrt(2*16*ln 100)) = 13
constexpr int kG = 16;                       // doubles per person in the window table (g[0..13], padded)
constexpr int kTileH = 8, kTileW = 32, kThreads = kTileH * kTileW;
constexpr int kChunk = 60;                   // persons per culling pass (60*17 = 1020 candidate blobs)
constexpr int kInvisible = 0x7fffffff;

struct RenderTables {
    double* g;      // [P][kG]
    int2* centre;   // [P][17]  (cy, cx); cy == kInvisible for an invisible keypoint
    int* half;      // [P]
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

inline RenderTables carve(void* ws, int P) {
    RenderTables t;
    unsigned char* p = reinterpret_cast<unsigned char*>(ws);
    t.g = reinterpret_cast<double*>(p);
    p += align16((size_t)P * kG * sizeof(double));
    t.centre = reinterpret_cast<int2*>(p);
    p += align16((size_t)P * kParts * sizeof(int2));
    t.half = reinterpret_cast<int*>(p);
    return t;
}

// heatmap_creation.py:30-37,78-84: per-person sigma, half window k and the 1-D window; :23-24,57,104-107: centres.
__global__ void __launch_bounds__(256) render_prepare_kernel(const int32_t* __restrict__ keypoints,
                                                              const float* __restrict__ boxes, int P, float hm1,
                                                              float wm1, float oh1, float ow1, RenderTables t) {
    // Every operation below is one IEEE-754 round-to-nearest step of the numpy code: no contraction into FMAs, and
    // sqrtf / operator/ are the correctly rounded forms (the __f*_rn intrinsics map to the approximate native ops).
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * kParts) return;
    const int p = i / kParts, j = i - p * kParts;
    const float ymin = boxes[p * 4 + 0], xmin = boxes[p * 4 + 1], ymax = boxes[p * 4 + 2], xmax = boxes[p * 4 + 3];
    const float area = (ymax - ymin) * (xmax - xmin);
    float s = sqrtf(area) * 0.007f;
    s = fminf(fmaxf(s, 1.0f), 4.0f);
    const float s2 = s * s;
    // k = ceil(sqrt(float32(-2 s^2) * ln(0.01)))  in float64
    const double arg = (double)(-2.0f * s2) * -0x1.26bb1bbb55515p+2;
    int k = (int)ceil(sqrt(arg));
    k = k > kMaxHalf ? kMaxHalf : k;
    if (j == 0) t.half[p] = k;
    if (j <= kMaxHalf) {
        const double sig2 = (double)((2.0f * s) * s);
        t.g[p * kG + j] = j <= k ? exp(-(double)(j * j) / sig2) : 0.0;
    }
    const int32_t* kp = keypoints + (size_t)i * 3;   // (y, x, visibility)
    int2 c;
    if (kp[2] > 0) {
        const float ny = (float)kp[0] / hm1, nx = (float)kp[1] / wm1;
        c.x = (int)rintf(ny * oh1);
        c.y = (int)rintf(nx * ow1);
    } else {
        c.x = kInvisible;
        c.y = 0;
    }
    t.centre[i] = c;
}

__global__ void __launch_bounds__(kThreads) render_kernel(const int32_t* __restrict__ first_person, int h, int w,
                                                         int tiles_x, RenderTables t, float* __restrict__ out) {
    __shared__ float tile[kThreads * kParts];           // [row][col][17] == the global layout of a 32-px row piece
    __shared__ double gl[kChunk * (kMaxHalf + 1)];
    __shared__ int4 hits[kChunk * kParts];              // (cy, cx, k, part | local person << 8)
    __shared__ int nhits;

    const int b = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * kTileH, tx0 = (blockIdx.x % tiles_x) * kTileW;
    const int tid = threadIdx.x;
    const int y = ty0 + (tid >> 5), x = tx0 + (tid & 31);
    const int p_begin = first_person[b], p_end = first_person[b + 1];

#pragma unroll
    for (int j = 0; j < kParts; ++j) tile[j * kThreads + tid] = 0.f;

    for (int p0 = p_begin; p0 < p_end; p0 += kChunk) {
        const int np = min(kChunk, p_end - p0);
        if (tid == 0) nhits = 0;
        __syncthreads();
        for (int i = tid; i < np * kParts; i += kThreads) {
            const int lp = i / kParts, j = i - lp * kParts;
            const int2 c = t.centre[(size_t)p0 * kParts + i];
            const int k = t.half[p0 + lp];
            if (c.x != kInvisible && c.x + k >= ty0 && c.x - k < ty0 + kTileH && c.y + k >= tx0 &&
                c.y - k < tx0 + kTileW) {
                const int slot = atomicAdd(&nhits, 1);
                hits[slot] = make_int4(c.x, c.y, k, j | (lp << 8));
            }
        }
        __syncthreads();
        const int n = nhits;
        if (n > 0) {
            for (int i = tid; i < np * (kMaxHalf + 1); i += kThreads) {
                const int lp = i / (kMaxHalf + 1), d = i - lp * (kMaxHalf + 1);
                gl[i] = t.g[(size_t)(p0 + lp) * kG + d];
            }
            __syncthreads();
            for (int i = 0; i < n; ++i) {
                const int4 hit = hits[i];
                const int dy = abs(y - hit.x), dx = abs(x - hit.y);
                if (dy <= hit.z && dx <= hit.z) {
                    const double* g = gl + (hit.w >> 8) * (kMaxHalf + 1);
                    const float v = (float)(g[dy] * g[dx]);
                    float* cell = tile + tid * kParts + (hit.w & 255);
                    *cell = fmaxf(*cell, v);
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();

    const int rows = min(kTileH, h - ty0);
    if (tx0 + kTileW <= w && (w & 3) == 0) {
        constexpr int kRowVec = kTileW * kParts / 4;     // 136 float4 per tile row
        const float4* src = reinterpret_cast<const float4*>(tile);
        for (int i = tid; i < rows * kRowVec; i += kThreads) {
            const int r = i / kRowVec, c = i - r * kRowVec;
            float4* dst = reinterpret_cast<float4*>(out + (((size_t)b * h + ty0 + r) * w + tx0) * kParts);
            dst[c] = src[i];
        }
    } else {
        const int cols = min(kTileW, w - tx0);
        for (int i = tid; i < rows * cols * kParts; i += kThreads) {
            const int r = i / (cols * kParts), c = i - r * (cols * kParts);
            out[(((size_t)b * h + ty0 + r) * w + tx0) * kParts + c] = tile[r * kTileW * kParts + c];
        }
    }
}

}  // namespace

extern "C" size_t mpn_heatmap_render_workspace_bytes(int total_persons) {
    if (total_persons <= 0) return 16;
    const size_t P = (size_t)total_persons;
    return align16(P * kG * sizeof(double)) + align16(P * kParts * sizeof(int2)) + align16(P * sizeof(int));
}

extern "C" int mpn_heatmap_render(const int32_t* keypoints, const float* boxes, const int32_t* first_person, int B,
                                  int total_persons, int width, int height, int downsample, float* out,
                                  void* workspace, size_t workspace_bytes, mpn_stream_t stream) {
    MPN_REQUIRE(B >= 0 && total_persons >= 0, MPN_ERR_BAD_SHAPE, "render: bad B=%d persons=%d", B, total_persons);
    MPN_REQUIRE(width >= 2 && height >= 2 && downsample >= 1, MPN_ERR_BAD_SHAPE,
                "render: width, height must be >= 2 and downsample >= 1 (got %d x %d / %d)", width, height,
                downsample);
    MPN_REQUIRE(width < (1 << 24) && height < (1 << 24), MPN_ERR_BAD_SHAPE, "render: image too large");
    if (B == 0) return MPN_OK;
    const int h = mpn_div_up(height, downsample), w = mpn_div_up(width, downsample);
    MPN_REQUIRE(B <= 65535, MPN_ERR_BAD_SHAPE, "render: B must be <= 65535");
    MPN_REQUIRE(first_person && out && workspace, MPN_ERR_BAD_ARG, "render: null pointer");
    MPN_REQUIRE(total_persons == 0 || (keypoints && boxes), MPN_ERR_BAD_ARG, "render: null pointer");
    MPN_REQUIRE(mpn_aligned16(out) && mpn_aligned16(workspace), MPN_ERR_BAD_ALIGN,
                "render: out/workspace must be 16-byte aligned");
    MPN_REQUIRE(workspace_bytes >= mpn_heatmap_render_workspace_bytes(total_persons), MPN_ERR_WORKSPACE,
                "render: workspace too small (%zu < %zu)", workspace_bytes,
                mpn_heatmap_render_workspace_bytes(total_persons));
    hipStream_t st = (hipStream_t)stream;
    const RenderTables t = carve(workspace, total_persons);
    if (total_persons > 0) {
        const int n = total_persons * kParts;
        render_prepare_kernel<<<mpn_div_up(n, 256), 256, 0, st>>>(keypoints, boxes, total_persons,
                                                                  (float)(height - 1.0), (float)(width - 1.0),
                                                                  (float)(h - 1), (float)(w - 1), t);
        MPN_LAUNCH_CHECK();
    }
    const int tiles_x = mpn_div_up(w, kTileW), tiles_y = mpn_div_up(h, kTileH);
    render_kernel<<<dim3((unsigned)(tiles_x * tiles_y), (unsigned)B), kThreads, 0, st>>>(first_person, h, w,
                                                                                         tiles_x, t, out);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
