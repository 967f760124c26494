// Inference glue between the keypoint heatmaps and the pose residual network (create_pb.py:86-142):
//   mpn_heatmap_minmax   per (image, channel) min / max of the sigmoid heatmaps            (create_pb.py:90-92)
//   mpn_prn_crop         (h - m) / (M - m) * [M > 0.2], then tf.image.crop_and_resize
//                        (bilinear, extrapolation 0) of every person box to 56 x 36       (create_pb.py:93-109)
//   mpn_prn_decode       softmax over the 2016 positions of each keypoint channel, its maximum (score) and the
//                        normalised (y, x) of the first maximum                            (create_pb.py:114-138)
// All three are HBM / latency-bound byte work on small tensors (35 MB of heatmaps at batch 32, 17.5 MB of crops per
// 128 persons); arithmetic is plain IEEE f32 in the reference's operation order (no FMA contraction).
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;
constexpr int kPL = 15;            // pixel lanes per channel: 17 channels x 15 lanes = 255 threads
constexpr int kMaxC = 17;

// order-preserving float <-> unsigned key (atomicMin / atomicMax on keys = min / max on floats; exact and order-free)
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void minmax_init_kernel(unsigned* __restrict__ keys, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = (i & 1) ? f2key(-INFINITY) : f2key(INFINITY);   // [..][0] = running min, [..][1] = running max
}

// grid (splits, B). A block walks its share of the image in chunks of 240 pixels: coalesced 16-byte loads into LDS, then
// thread (channel, lane) scans 16 pixels of its channel (LDS stride 17 dwords: conflict-free).
__global__ __launch_bounds__(kThreads) void minmax_kernel(const float* __restrict__ hm, int HW, int C, int splits,
                                                         unsigned* __restrict__ keys) {
    constexpr int CHUNK = 16 * kPL;   // pixels per chunk
    __shared__ __attribute__((aligned(16))) float tile[CHUNK * kMaxC];
    __shared__ float red[2][kThreads];
    const int b = blockIdx.y;
    const int c = threadIdx.x / kPL, pl = threadIdx.x % kPL;
    const bool on = c < C;
    const int per = (HW + splits - 1) / splits;
    const int p_begin = blockIdx.x * per, p_end = min(p_begin + per, HW);
    const float* img = hm + (long long)b * HW * C;
    float lo = INFINITY, hi = -INFINITY;
    for (int p0 = p_begin; p0 < p_end; p0 += CHUNK) {
        const int np = min(CHUNK, p_end - p0);
        const int nfl = np * C;
        const long long off = (long long)p0 * C;
        __syncthreads();
        if ((((uintptr_t)(img + off)) & 15) == 0) {
            for (int i = threadIdx.x * 4; i < nfl; i += kThreads * 4) {
                if (i + 4 <= nfl) *reinterpret_cast<float4*>(&tile[i]) = *reinterpret_cast<const float4*>(img + off + i);
                else for (int k = i; k < nfl; ++k) tile[k] = img[off + k];
            }
        } else {
            for (int i = threadIdx.x; i < nfl; i += kThreads) tile[i] = img[off + i];
        }
        __syncthreads();
        if (on)
            for (int p = pl; p < np; p += kPL) {
                const float v = tile[p * C + c];
                lo = fminf(lo, v);
                hi = fmaxf(hi, v);
            }
    }
    red[0][threadIdx.x] = lo;
    red[1][threadIdx.x] = hi;
    __syncthreads();
    if (on && pl == 0) {
        for (int k = 1; k < kPL; ++k) { lo = fminf(lo, red[0][c * kPL + k]); hi = fmaxf(hi, red[1][c * kPL + k]); }
        if (p_begin < p_end) {
            atomicMin(&keys[((long long)b * C + c) * 2 + 0], f2key(lo));
            atomicMax(&keys[((long long)b * C + c) * 2 + 1], f2key(hi));
        }
    }
}

// one thread per crop element, channels fastest (the 17 channels of a tap are 68 contiguous bytes, stores are dense)
__global__ __launch_bounds__(kThreads) void crop_kernel(const float* __restrict__ hm, const unsigned* __restrict__ keys,
                                                       const float* __restrict__ boxes, const int* __restrict__ box_ind,
                                                       const int* __restrict__ num_boxes, int max_boxes, int slot0,
                                                       long long total, int B, int H, int W, int C, int CH, int CW,
                                                       float thresh, float* __restrict__ crops) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int x = (int)(r % CW); r /= CW;
        const int y = (int)(r % CH);
        const int n = (int)(r / CH);
        // box of crop n: boxes[n] of image box_ind[n], or (box_ind == NULL) slot slot0 + n of a [B, max_boxes, 4] detection
        // array - image s / max_boxes, live when s % max_boxes < num_boxes[image] (create_pb.py:96-104)
        int b, bn = n;
        if (box_ind != nullptr) {
            b = box_ind[n];
        } else {
            bn = slot0 + n;
            const int img = bn / max_boxes;
            b = (img < B && bn - img * max_boxes < num_boxes[img]) ? img : -1;
        }
        float out = 0.f;   // extrapolation value
        if (b >= 0 && b < B) {
            const float y1 = boxes[bn * 4 + 0], x1 = boxes[bn * 4 + 1], y2 = boxes[bn * 4 + 2], x2 = boxes[bn * 4 + 3];
            // tensorflow/core/kernels/crop_and_resize_op.cc (1.15), bilinear
            const float hs = (CH > 1) ? (y2 - y1) * (float)(H - 1) / (float)(CH - 1) : 0.f;
            const float ws = (CW > 1) ? (x2 - x1) * (float)(W - 1) / (float)(CW - 1) : 0.f;
            const float in_y = (CH > 1) ? y1 * (float)(H - 1) + (float)y * hs : 0.5f * (y1 + y2) * (float)(H - 1);
            const float in_x = (CW > 1) ? x1 * (float)(W - 1) + (float)x * ws : 0.5f * (x1 + x2) * (float)(W - 1);
            if (!(in_y < 0.f || in_y > (float)(H - 1) || in_x < 0.f || in_x > (float)(W - 1))) {
                const int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
                const int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
                const float yl = in_y - (float)ty, xl = in_x - (float)lx;
                const float m = key2f(keys[((long long)b * C + c) * 2 + 0]);
                const float M = key2f(keys[((long long)b * C + c) * 2 + 1]);
                const float mask = M > thresh ? 1.f : 0.f;
                const float den = M - m;
                const float* img = hm + (long long)b * H * W * C + c;
                auto tap = [&](int yy, int xx) { return (img[((long long)yy * W + xx) * C] - m) / den * mask; };
                const float tl = tap(ty, lx), tr = tap(ty, rx), bl = tap(by, lx), br = tap(by, rx);
                const float top = tl + (tr - tl) * xl;
                const float bot = bl + (br - bl) * xl;
                out = top + (bot - top) * yl;
            }
        }
        crops[i] = out;
    }
}

// one block of 1024 threads per crop, LT = (1024 / C) * C of them active: thread t owns elements t, t + LT, ... of the crop's
// flattened [P][C] logits (consecutive threads read consecutive addresses; a thread's channel t % C never changes). Channel
// reductions through LDS in a fixed order; the softmax is never formed: score = 1 / sum exp(z - max), position = the first
// arg-max (tf.argmax: smallest index among equal maxima).
constexpr int kDecThreads = 1024;
__global__ __launch_bounds__(kDecThreads) void decode_kernel(const float* __restrict__ logits, int P, int CW, int CH, int C,
                                                            float* __restrict__ scores, float* __restrict__ positions) {
    __shared__ float redv[kDecThreads];
    __shared__ int redi[kDecThreads];
    __shared__ float chan[32];
    const int n = blockIdx.x;
    const int t = threadIdx.x;
    const int per = kDecThreads / C, LT = per * C;
    const bool on = t < LT;
    const int c = t % C;
    const float* z = logits + (long long)n * P * C;
    const int total = P * C;
    float m = -INFINITY;
    if (on)
        for (int i = t; i < total; i += LT) m = fmaxf(m, z[i]);
    redv[t] = m;
    __syncthreads();
    if (t < C) {
        float r = -INFINITY;
        for (int k = 0; k < per; ++k) r = fmaxf(r, redv[k * C + t]);
        chan[t] = r;
    }
    __syncthreads();
    m = chan[c];
    __syncthreads();
    // softmax numerators exp(z - m): their sum, and the FIRST position whose numerator is the maximum (= 1.0f)
    float se = 0.f;
    int first = 0x7fffffff;
    if (on)
        for (int i = t; i < total; i += LT) {
            const float e = expf(z[i] - m);
            se += e;
            const int p = i / C;
            if (e == 1.0f && p < first) first = p;
        }
    redv[t] = se;
    redi[t] = first;
    __syncthreads();
    if (t < C) {
        float sum = 0.f;
        int f = 0x7fffffff;
        for (int k = 0; k < per; ++k) { sum += redv[k * C + t]; f = min(f, redi[k * C + t]); }
        if (f == 0x7fffffff) f = 0;   // all-NaN column: tf.argmax returns 0
        scores[(long long)n * C + t] = 1.0f / sum;
        positions[((long long)n * C + t) * 2 + 0] = (float)(f / CW) / (float)CH;
        positions[((long long)n * C + t) * 2 + 1] = (float)(f % CW) / (float)CW;
    }
}

}  // namespace

/* minmax_keys: B*C*2 u32 (opaque order-preserving keys; mpn_prn_crop decodes them). heatmaps f32 [B,h,w,C], C <= 17 */
extern "C" int mpn_heatmap_minmax(const float* heatmaps, int B, int h, int w, int C, void* minmax_keys, mpn_stream_t stream) {
    MPN_REQUIRE(heatmaps && minmax_keys, MPN_ERR_BAD_ARG, "heatmap_minmax: null pointer");
    MPN_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0 && C <= kMaxC, MPN_ERR_BAD_SHAPE, "heatmap_minmax: bad shape (C <= 17)");
    hipStream_t st = (hipStream_t)stream;
    const int n = B * C * 2;
    minmax_init_kernel<<<(n + 255) / 256, 256, 0, st>>>((unsigned*)minmax_keys, n);
    int splits = (h * w + 3839) / 3840;   // 16 chunks of 240 pixels per block ...
    const int chunks = (h * w + 239) / 240;
    if (B * splits < 256) splits = (256 + B - 1) / B;    // ... fewer where that leaves CUs idle (one image: 7 blocks took 35 us)
    if (splits > chunks) splits = chunks;
    if (splits < 1) splits = 1;
    if (splits > 64 && B * 64 >= 256) splits = 64;
    minmax_kernel<<<dim3((unsigned)splits, (unsigned)B), kThreads, 0, st>>>(heatmaps, h * w, C, splits, (unsigned*)minmax_keys);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* boxes f32 [nb,4] normalised (y1,x1,y2,x2), box_ind i32 [nb] -> crops f32 [nb,crop_h,crop_w,C] */
extern "C" int mpn_prn_crop(const float* heatmaps, const void* minmax_keys, const float* boxes, const int* box_ind, int nb, int B,
                            int h, int w, int C, int crop_h, int crop_w, float threshold, float* crops, mpn_stream_t stream) {
    MPN_REQUIRE(heatmaps && minmax_keys && boxes && box_ind && crops, MPN_ERR_BAD_ARG, "prn_crop: null pointer");
    MPN_REQUIRE(nb > 0 && B > 0 && h > 0 && w > 0 && C > 0 && C <= kMaxC && crop_h > 0 && crop_w > 0, MPN_ERR_BAD_SHAPE,
                "prn_crop: bad shape");
    const long long total = (long long)nb * crop_h * crop_w * C;
    long long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 16384) blocks = 16384;
    crop_kernel<<<(unsigned)blocks, kThreads, 0, (hipStream_t)stream>>>(heatmaps, (const unsigned*)minmax_keys, boxes, box_ind, nullptr, 1,
                                                                       0, total, B, h, w, C, crop_h, crop_w, threshold, crops);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* The same for `nb` consecutive SLOTS slot0 .. slot0 + nb - 1 of a detector's padded output (retinanet.py:60-84): boxes f32
 * [B,max_boxes,4], num_boxes i32 [B]; slot s is box s % max_boxes of image s / max_boxes and gives a zero crop when it is
 * padding (s % max_boxes >= num_boxes[image]) or lies past the array: the per-image [:n] slices, the box_ind vectors and the
 * concat of create_pb.py:96-104 without materialising them. */
extern "C" int mpn_prn_crop_slots(const float* heatmaps, const void* minmax_keys, const float* boxes, const int* num_boxes,
                                  int slot0, int nb, int max_boxes, int B, int h, int w, int C, int crop_h, int crop_w,
                                  float threshold, float* crops, mpn_stream_t stream) {
    MPN_REQUIRE(heatmaps && minmax_keys && boxes && num_boxes && crops, MPN_ERR_BAD_ARG, "prn_crop_slots: null pointer");
    MPN_REQUIRE(nb > 0 && slot0 >= 0 && max_boxes > 0 && B > 0 && h > 0 && w > 0 && C > 0 && C <= kMaxC && crop_h > 0 && crop_w > 0,
                MPN_ERR_BAD_SHAPE, "prn_crop_slots: bad shape");
    const long long total = (long long)nb * crop_h * crop_w * C;
    long long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 16384) blocks = 16384;
    crop_kernel<<<(unsigned)blocks, kThreads, 0, (hipStream_t)stream>>>(heatmaps, (const unsigned*)minmax_keys, boxes, nullptr, num_boxes,
                                                                       max_boxes, slot0, total, B, h, w, C, crop_h, crop_w, threshold, crops);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

/* logits f32 [nb,crop_h,crop_w,C] -> scores f32 [nb,C], positions f32 [nb,C,2] = (y / crop_h, x / crop_w) of the argmax */
extern "C" int mpn_prn_decode(const float* logits, int nb, int crop_h, int crop_w, int C, float* scores, float* positions,
                              mpn_stream_t stream) {
    MPN_REQUIRE(logits && scores && positions, MPN_ERR_BAD_ARG, "prn_decode: null pointer");
    MPN_REQUIRE(nb > 0 && crop_h > 0 && crop_w > 0 && C > 0 && C <= kMaxC, MPN_ERR_BAD_SHAPE, "prn_decode: bad shape (C <= 17)");
    decode_kernel<<<(unsigned)nb, kDecThreads, 0, (hipStream_t)stream>>>(logits, crop_h * crop_w, crop_w, crop_h, C, scores, positions);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
