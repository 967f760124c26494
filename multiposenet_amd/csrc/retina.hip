// L3: RetinaNet person-detector head - everything around its convolutions (BASELINE config 4).
// The convolutions, batch-norms and the optimizer are the kernels of the keypoint path (conv_mfma, bn, optim); this file
// holds what is specific to the detector:
//   mpn_patchify3x3s2 / mpn_unpatchify3x3s2  the stride-2 3x3 convolutions of the coarse FPN branch (detector/fpn.py:42-46:
//       p6, p7; conv2d_same, layer_utils.py:19-39: pad 1, then VALID) as a gather into [pixels, 9*C] rows + a 1x1 GEMM
//       (the gather applies the producer's batch-norm affine + activation), and the transposed gather for the data gradient;
//   mpn_retina_match_*   anchor <-> groundtruth matching and regression targets (detector/training_target_creation.py:5-159,
//       detector/utils/box_utils.py:14-110), float32 step by step like the reference's TF ops (IEEE division, no FMA
//       contraction); arg-maxes take the FIRST maximum (tf.argmax);
//   mpn_retina_loss      focal + smooth-L1 losses and their gradients w.r.t. the raw outputs of the two towers
//       (detector/retinanet.py:86-217), bias gradients included;
//   mpn_retina_nms       sigmoid, box decoding, clipping and greedy non-maximum suppression (detector/retinanet.py:60-84,
//       detector/utils/nms.py:6-61; tf.image.non_max_suppression restated from non_max_suppression_op.cc).
// All HBM-bound elementwise / reduction work; reductions are deterministic (ordered keys, integer counters, partial slabs).
#include "common.h"
#include <math.h>

namespace {

constexpr int kThreads = 256;
constexpr int kLevels = 5;
constexpr int kAPL = 6;            // anchors per location (anchor_generator.py: 2 scale multipliers x 3 aspect ratios)
constexpr float kEps = 1e-8f;      // constants.py:16

// ------------------------------------------------------------------------------------------------ stride-2 3x3 gathers
template <typename T>
__global__ __launch_bounds__(kThreads) void patchify_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int H, int W,
                                                            int C, int OH, int OW, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int act, long long nvec) {
    constexpr int VE = Vec16<T>::N;
    const int cvec = C / VE;
    const float lo = (act != MPN_ACT_NONE) ? 0.f : -INFINITY;
    const float hi = (act == MPN_ACT_RELU6) ? 6.f : INFINITY;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long long)gridDim.x * kThreads) {
        long long r = i;
        const int cv = (int)(r % cvec); r /= cvec;
        const int tap = (int)(r % 9); r /= 9;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        const int iy = 2 * oy - 1 + tap / 3, ix = 2 * ox - 1 + tap % 3;
        Vec16<T> v;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            v.load(x + (((long long)n * H + iy) * W + ix) * C + cv * VE);
            if (scale != nullptr) {
                float f[VE];
                v.unpack(f);
#pragma unroll
                for (int j = 0; j < VE; ++j) f[j] = __builtin_amdgcn_fmed3f(f[j] * scale[cv * VE + j] + shift[cv * VE + j], lo, hi);
                v.pack(f);
            }
        } else {
            v.zero();     // the zero padding is applied AFTER the activation (tf.pad of the activated tensor)
        }
        v.store(out + i * VE);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void unpatchify_kernel(const T* __restrict__ dp, T* __restrict__ dx, int N, int H, int W,
                                                              int C, int OH, int OW, long long nvec) {
    constexpr int VE = Vec16<T>::N;
    const int cvec = C / VE;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long long)gridDim.x * kThreads) {
        long long r = i;
        const int cv = (int)(r % cvec); r /= cvec;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int n = (int)(r / H);
        float acc[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) acc[j] = 0.f;
        // iy = 2 oy - 1 + ky  ->  oy = (iy + 1 - ky) / 2 for the ky of matching parity
        for (int ky = 0; ky < 3; ++ky) {
            const int t = iy + 1 - ky;
            if (t < 0 || (t & 1)) continue;
            const int oy = t >> 1;
            if (oy >= OH) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int s = ix + 1 - kx;
                if (s < 0 || (s & 1)) continue;
                const int ox = s >> 1;
                if (ox >= OW) continue;
                Vec16<T> v;
                v.load(dp + ((((long long)n * OH + oy) * OW + ox) * 9 + ky * 3 + kx) * C + cv * VE);
                float f[VE];
                v.unpack(f);
#pragma unroll
                for (int j = 0; j < VE; ++j) acc[j] += f[j];
            }
        }
        Vec16<T> o;
        o.pack(acc);
        o.store(dx + i * VE);
    }
}

// ------------------------------------------------------------------------------------------------ matching
struct Box { float ymin, xmin, ymax, xmax; };

// box_utils.py:14-47 in float32, one rounding per TF op
__device__ __forceinline__ float iou_f32(const Box& a, const Box& b) {
#pragma clang fp contract(off)
    const float ih = fmaxf(0.0f, fminf(a.ymax, b.ymax) - fmaxf(a.ymin, b.ymin));
    const float iw = fmaxf(0.0f, fminf(a.xmax, b.xmax) - fmaxf(a.xmin, b.xmin));
    const float inter = ih * iw;
    const float a1 = (a.ymax - a.ymin) * (a.xmax - a.xmin);
    const float a2 = (b.ymax - b.ymin) * (b.xmax - b.xmin);
    const float uni = (a1 + a2) - inter;
    const float q = __fdiv_rn(inter, uni + kEps);
    return fminf(fmaxf(q, 0.0f), 1.0f);
}

// (a launch rather than hipMemsetAsync: the step is captured into a hipGraph)
__global__ void match_reset_kernel(unsigned long long* __restrict__ keys, int n, int* __restrict__ num_matched) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = 0ull;
    if (i == 0) *num_matched = 0;
}

// pass 1: per (image, anchor) the best groundtruth box (first maximum) -> preliminary matches (training_target_creation.py:
// 86-99); per (image, groundtruth box) the best anchor as an ordered key, merged with atomicMax: key = (iou bits << 32) |
// ~anchor index, so the largest iou wins and among equal ious the smallest anchor index (tf.argmax) - order-free, exact.
__global__ __launch_bounds__(kThreads) void match_anchor_kernel(const float* __restrict__ anchors, const float* __restrict__ gt,
                                                                const int* __restrict__ num_boxes, int B, int A, int maxN,
                                                                float pos_thr, float neg_thr, int* __restrict__ matches,
                                                                unsigned long long* __restrict__ keys) {
    const int b = blockIdx.y;
    const int a = blockIdx.x * kThreads + threadIdx.x;
    const int N = min(num_boxes[b], maxN);
    extern __shared__ float sgt[];   // [maxN][4]
    for (int i = threadIdx.x; i < N * 4; i += kThreads) sgt[i] = gt[(long long)b * maxN * 4 + i];
    __syncthreads();
    const bool live = a < A;                                       // (no early return: the wave reductions below need every lane)
    const float4 av = *reinterpret_cast<const float4*>(anchors + (long long)(live ? a : 0) * 4);
    const Box an = {av.x, av.y, av.z, av.w};
    int best = 0;
    float best_v = -1.f;
    for (int n = 0; n < N; ++n) {
        const Box g = {sgt[n * 4], sgt[n * 4 + 1], sgt[n * 4 + 2], sgt[n * 4 + 3]};
        const float v = live ? iou_f32(g, an) : 0.f;
        if (v > best_v) { best_v = v; best = n; }
        // anchors that do not overlap the box cannot become its forced match (a forced match needs iou >= 0.05): only the
        // few thousand overlapping ones issue an atomic. A box nothing overlaps keeps key 0 = "no anchor".
        // One atomic per wave and box: the lanes' keys are reduced first (a box overlapped by a few thousand anchors drew as
        // many contended atomics: 112 us of the detector step).
        unsigned long long key = v > 0.f ? ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)a) : 0ull;
        if (__any(key != 0ull)) {   // (wave-uniform)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned lo = __shfl_xor((unsigned)key, o, 64);
                const unsigned hi = __shfl_xor((unsigned)(key >> 32), o, 64);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                key = other > key ? other : key;
            }
            if ((threadIdx.x & 63) == 0) atomicMax(&keys[(long long)b * maxN + n], key);
        }
    }
    int m = -1;
    if (N > 0) {
        const bool pos = best_v >= pos_thr;
        if (pos_thr == neg_thr) m = pos ? best : -1;
        else m = pos ? best : (neg_thr > best_v ? -1 : -2);
    }
    if (live) matches[(long long)b * A + a] = m;
}

// pass 2: forced matches (training_target_creation.py:101-121): groundtruth box n forces its best anchor f(n); the anchor
// takes the SMALLEST n that forces it (argmax over the 0/1 indicator rows, computed before the is_okay mask) whenever ANY
// box forcing it has iou >= 0.05. One block per image, one thread per box, O(N^2).
__global__ __launch_bounds__(kThreads) void match_forced_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ num_boxes,
                                                                int A, int maxN, int* __restrict__ matches) {
    const int b = blockIdx.x;
    const int N = min(num_boxes[b], maxN);
    for (int n = threadIdx.x; n < N; n += kThreads) {
        const unsigned long long kn = keys[(long long)b * maxN + n];
        const unsigned f = 0xFFFFFFFFu - (unsigned)(kn & 0xFFFFFFFFull);
        int row = n;
        bool ok = false;
        for (int j = 0; j < N; ++j) {
            const unsigned long long kj = keys[(long long)b * maxN + j];
            if (0xFFFFFFFFu - (unsigned)(kj & 0xFFFFFFFFull) != f) continue;
            if (j < row) row = j;
            ok = ok || (__uint_as_float((unsigned)(kj >> 32)) >= 0.05f);
        }
        if (row == n && ok && f < (unsigned)A) matches[(long long)b * A + f] = n;   // one writer per anchor: the smallest n
    }
}

// pass 3: regression targets of the matched anchors (create_targets + encode), zeros elsewhere; matched-anchor count
__global__ __launch_bounds__(kThreads) void match_targets_kernel(const float* __restrict__ anchors, const float* __restrict__ gt,
                                                                 const int* __restrict__ matches, int A, int maxN,
                                                                 float* __restrict__ targets, int* __restrict__ num_matched) {
#pragma clang fp contract(off)
    const int b = blockIdx.y;
    const int a = blockIdx.x * kThreads + threadIdx.x;
    int cnt = 0;
    if (a < A) {
        const int m = matches[(long long)b * A + a];
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m >= 0) {
            cnt = 1;
            const float4 an = *reinterpret_cast<const float4*>(anchors + (long long)a * 4);
            const float4 g = *reinterpret_cast<const float4*>(gt + ((long long)b * maxN + m) * 4);
            float ha = an.z - an.x, wa = an.w - an.y;
            const float ya = an.x + 0.5f * ha, xa = an.y + 0.5f * wa;
            float h = g.z - g.x, w = g.w - g.y;
            const float y = g.x + 0.5f * h, x = g.y + 0.5f * w;
            ha += kEps; wa += kEps; h += kEps; w += kEps;
            t.x = __fdiv_rn(y - ya, ha) * 10.0f;
            t.y = __fdiv_rn(x - xa, wa) * 10.0f;
            t.z = logf(__fdiv_rn(h, ha)) * 5.0f;
            t.w = logf(__fdiv_rn(w, wa)) * 5.0f;
        }
        *reinterpret_cast<float4*>(targets + ((long long)b * A + a) * 4) = t;
    }
    // integer count: order-free, exact
    const unsigned long long bal = __ballot(cnt != 0);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(num_matched, (int)__popcll(bal));
}

// ------------------------------------------------------------------------------------------------ losses
struct RetinaLevels {
    const void* logits[kLevels];   // [B,h,w,8]  raw class-tower output (channels 6, 7 are padding)
    const void* boxes[kLevels];    // [B,h,w,24] raw box-tower output
    void* dlogits[kLevels];        // gradients in the same layouts (NULL: evaluation)
    void* dboxes[kLevels];
    int hw[kLevels];               // h*w per level
    int first[kLevels + 1];        // first anchor of each level; first[kLevels] = A
};

template <typename T>
__global__ __launch_bounds__(kThreads) void retina_loss_kernel(const RetinaLevels lv, const float* __restrict__ cls_bias,
                                                               const float* __restrict__ box_bias, const int* __restrict__ matches,
                                                               const float* __restrict__ targets, const int* __restrict__ num_matched,
                                                               int B, int A, float gamma, float alpha, float loc_w, float cls_w,
                                                               float* __restrict__ part /* [blocks][32] */) {
    __shared__ float red[kThreads / 64][32];
    const int b = blockIdx.y;
    const int loc = blockIdx.x * kThreads + threadIdx.x;       // (pixel, anchor) location index inside the image: a = loc
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    const float inv_norm = 1.0f / fmaxf((float)num_matched[0], 1.0f);
    if (loc < A) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < kLevels; ++k)
            if (loc >= lv.first[k]) l = k;
        const int rel = loc - lv.first[l];
        const int pix = rel / kAPL, k = rel - pix * kAPL;
        const long long pbase = (long long)b * lv.hw[l] + pix;
        const T* lg = reinterpret_cast<const T*>(lv.logits[l]) + pbase * 8 + k;
        const T* bx = reinterpret_cast<const T*>(lv.boxes[l]) + pbase * 24 + k * 4;
        const int m = matches[(long long)b * A + loc];
        const float is_matched = m >= 0 ? 1.f : 0.f, not_ignore = m >= -1 ? 1.f : 0.f;
        // ---- classification: focal loss on the logit (retinanet.py:188-217)
        const float x = to_f32(*lg) + cls_bias[k];
        const float t = is_matched;
        const float nlp = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));       // sigmoid_cross_entropy_with_logits
        const float p = 1.0f / (1.0f + expf(-x));
        const float p_t = t == 1.0f ? p : 1.0f - p;
        const float q = 1.0f - p_t;
        const float a_t = t == 1.0f ? alpha : 1.0f - alpha;
        const float qg = powf(q, gamma);
        acc[0] = not_ignore * qg * a_t * nlp;
        const float sgn = t == 1.0f ? 1.f : -1.f;
        // d/dx [a_t q^g nlp] = a_t * sgn * (-g q^g p_t nlp - q^(g+1))
        const float dcls = not_ignore * a_t * sgn * (-gamma * qg * p_t * nlp - qg * q) * (cls_w * inv_norm);
        // ---- localisation: smooth L1 on the matched anchors (retinanet.py:169-185)
        float dloc[4] = {0.f, 0.f, 0.f, 0.f};
        float lsum = 0.f;
        const float4 tg = *reinterpret_cast<const float4*>(targets + ((long long)b * A + loc) * 4);
        const float tv[4] = {tg.x, tg.y, tg.z, tg.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = (to_f32(bx[j]) + box_bias[k * 4 + j]) - tv[j];
            const float ad = fabsf(d);
            lsum += ad < 1.0f ? 0.5f * ad * ad : ad - 0.5f;
            dloc[j] = is_matched * fminf(fmaxf(d, -1.f), 1.f) * (loc_w * inv_norm);
        }
        acc[1] = is_matched * lsum;
        if (lv.dlogits[l] != nullptr) {
            T* dl = reinterpret_cast<T*>(lv.dlogits[l]) + pbase * 8 + k;
            *dl = from_f32<T>(dcls);
            if (k == 0) { dl[6] = from_f32<T>(0.f); dl[7] = from_f32<T>(0.f); }   // the padding channels carry no gradient
            T* db = reinterpret_cast<T*>(lv.dboxes[l]) + pbase * 24 + k * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) db[j] = from_f32<T>(dloc[j]);
            // bias gradients = sums of the ROUNDED gradients the convolutions' weight gradients see
            acc[2 + k] = to_f32(from_f32<T>(dcls));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[8 + k * 4 + j] = to_f32(from_f32<T>(dloc[j]));
        }
    }
    // block reduction in a fixed order: wave shuffles, then the four waves through LDS
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = wave_sum(acc[j]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 32; ++j) red[wave][j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float s = 0.f;
        for (int w = 0; w < kThreads / 64; ++w) s += red[w][threadIdx.x];
        part[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 32 + threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------ post-processing
// IOU() of tensorflow/core/kernels/non_max_suppression_op.cc
__device__ __forceinline__ float nms_iou(const float4 a, const float4 b) {
#pragma clang fp contract(off)
    const float ymin_i = fminf(a.x, a.z), xmin_i = fminf(a.y, a.w), ymax_i = fmaxf(a.x, a.z), xmax_i = fmaxf(a.y, a.w);
    const float ymin_j = fminf(b.x, b.z), xmin_j = fminf(b.y, b.w), ymax_j = fmaxf(b.x, b.z), xmax_j = fmaxf(b.y, b.w);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i), area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.f || area_j <= 0.f) return 0.f;
    const float iy = fmaxf(fminf(ymax_i, ymax_j) - fmaxf(ymin_i, ymin_j), 0.f);
    const float ix = fmaxf(fminf(xmax_i, xmax_j) - fmaxf(xmin_i, xmin_j), 0.f);
    const float inter = iy * ix;
    return __fdiv_rn(inter, (area_i + area_j) - inter);
}

constexpr int kNmsThreads = 1024;

// Two launches. (1) retina_candidates_kernel, a grid over (image, anchor chunk): sigmoid, the score test, box decoding - and the
// live candidates APPENDED to the image's list (one wave-aggregated atomicAdd per wave; the order of the list is arbitrary, the
// selection below orders by (score, anchor index), so results do not depend on it). A trained detector leaves hundreds of
// candidates out of 157 542 anchors. (2) retina_nms_kernel, one block per image: up to max_det rounds of {block arg-max over
// the live candidates (largest score, smallest anchor index), suppress what overlaps the winner} - greedy NMS without a sort -
// over the compacted list, held in LDS when it fits (kNmsLds candidates), else in the workspace.
// (Round 2 ran both phases in the one block per image over all A anchors: 838 us of a 2.1 ms inference call at 640 x 640.)
struct NmsCand { float4 box; float score; int anchor; int pad0, pad1; };   // 32 bytes
constexpr int kNmsLds = 4096;      // candidates kept in LDS (128 KB)

// (counts[B] is the call's OVERFLOW word: set when a list would grow past its A slots - possible only if a counter was not 0 when
//  the appends began; the host reads it with the outputs: mpn_retina_nms_overflow_offset)
__global__ void nms_reset_kernel(int* __restrict__ counts, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= B) counts[i] = 0;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void retina_candidates_kernel(const RetinaLevels lv, const float* __restrict__ cls_bias,
                                                                    const float* __restrict__ box_bias, const float* __restrict__ anchors,
                                                                    int A, float score_thr, NmsCand* __restrict__ cands, int* __restrict__ counts) {
#pragma clang fp contract(off)
    const int b = blockIdx.y;
    const int a = blockIdx.x * kThreads + threadIdx.x;
    bool live = false;
    NmsCand c;
    c.pad0 = c.pad1 = 0;
    if (a < A) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < kLevels; ++k)
            if (a >= lv.first[k]) l = k;
        const int rel = a - lv.first[l];
        const int pix = rel / kAPL, k = rel - pix * kAPL;
        const long long pbase = (long long)b * lv.hw[l] + pix;
        const float x = to_f32(reinterpret_cast<const T*>(lv.logits[l])[pbase * 8 + k]) + cls_bias[k];
        const float s = __fdiv_rn(1.0f, 1.0f + expf(-x));
        // nms.py:29-32 keeps score >= threshold; the NMS op itself admits score > threshold
        live = s >= score_thr && s > score_thr;
        if (live) {
            const T* cp = reinterpret_cast<const T*>(lv.boxes[l]) + pbase * 24 + k * 4;
            const float4 an = *reinterpret_cast<const float4*>(anchors + (long long)a * 4);
            const float ha = an.z - an.x, wa = an.w - an.y;
            const float ya = an.x + 0.5f * ha, xa = an.y + 0.5f * wa;
            const float ty = __fdiv_rn(to_f32(cp[0]) + box_bias[k * 4 + 0], 10.0f), tx = __fdiv_rn(to_f32(cp[1]) + box_bias[k * 4 + 1], 10.0f);
            const float th = __fdiv_rn(to_f32(cp[2]) + box_bias[k * 4 + 2], 5.0f), tw = __fdiv_rn(to_f32(cp[3]) + box_bias[k * 4 + 3], 5.0f);
            const float h = expf(th) * ha, w = expf(tw) * wa;
            const float yc = ty * ha + ya, xc = tx * wa + xa;
            c.box.x = fminf(fmaxf(yc - 0.5f * h, 0.f), 1.f); c.box.y = fminf(fmaxf(xc - 0.5f * w, 0.f), 1.f);
            c.box.z = fminf(fmaxf(yc + 0.5f * h, 0.f), 1.f); c.box.w = fminf(fmaxf(xc + 0.5f * w, 0.f), 1.f);
            c.score = s;
            c.anchor = a;
        }
    }
    // wave-aggregated append
    const unsigned long long m = __ballot(live);
    if (m != 0ull) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(counts + b, __popcll(m));
        base = __shfl(base, leader, 64);
        if (live) {
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            // (every anchor appends at most once: slot < A unless the counter did not start at 0 - refused AND reported, never dropped silently)
            if (slot < A) cands[(long long)b * A + slot] = c;
            else counts[gridDim.y] = 1;
        }
    }
}

__global__ __launch_bounds__(kNmsThreads) void retina_nms_kernel(const NmsCand* __restrict__ cands, const int* __restrict__ counts, int A,
                                                                 float iou_thr, int max_det, float* __restrict__ out_boxes,
                                                                 float* __restrict__ out_scores, int* __restrict__ out_num) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
    __shared__ unsigned long long wkey[kNmsThreads / 64];
    __shared__ int wslot[kNmsThreads / 64];
    __shared__ unsigned long long best_key;
    __shared__ float4 best_box;
    const int b = blockIdx.x;
    const int C_raw = counts[b];
    const int C = C_raw < A ? C_raw : A;          // (never past the image's list, whatever the counter holds)
    if (C_raw > A && threadIdx.x == 0) const_cast<int*>(counts)[gridDim.x] = 1;
    const bool in_lds = C <= kNmsLds;
    const NmsCand* gl = cands + (long long)b * A;
    float4* lbox = reinterpret_cast<float4*>(nms_smem);                          // [kNmsLds]
    float* lsc = reinterpret_cast<float*>(nms_smem + kNmsLds * 16);              // [kNmsLds] score, -1 = dead
    int* lan = reinterpret_cast<int*>(nms_smem + kNmsLds * 20);                  // [kNmsLds] anchor index
    // dead flags of the global path live in the list itself (score = -1): the list is this launch's scratch
    NmsCand* gw = const_cast<NmsCand*>(gl);
    if (in_lds) {
        for (int i = threadIdx.x; i < C; i += kNmsThreads) {
            const NmsCand c = gl[i];
            lbox[i] = c.box; lsc[i] = c.score; lan[i] = c.anchor;
        }
    }
    __syncthreads();
    int n_out = 0;
    // this thread's best surviving candidate: found by a scan before the first selection, afterwards by the suppression pass
    // of the previous selection itself (one pass over the candidates per detection instead of two)
    unsigned long long key = 0;
    int slot = -1;
    for (int i = threadIdx.x; i < C; i += kNmsThreads) {
        const float s = in_lds ? lsc[i] : gw[i].score;
        if (s > 0.f) {
            const int an = in_lds ? lan[i] : gw[i].anchor;
            const unsigned long long k2 = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)an);
            if (k2 > key) { key = k2; slot = i; }
        }
    }
    for (int it = 0; it < max_det; ++it) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o, 64);
            const int os = __shfl_xor(slot, o, 64);
            if (other > key) { key = other; slot = os; }
        }
        if ((threadIdx.x & 63) == 0) { wkey[threadIdx.x >> 6] = key; wslot[threadIdx.x >> 6] = slot; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long k3 = 0;
            int s3 = -1;
            for (int w = 0; w < kNmsThreads / 64; ++w)
                if (wkey[w] > k3) { k3 = wkey[w]; s3 = wslot[w]; }
            best_key = k3;
            if (k3 != 0) {
                const float4 bb = in_lds ? lbox[s3] : gw[s3].box;
                best_box = bb;
                float* ob = out_boxes + ((long long)b * max_det + it) * 4;
                ob[0] = bb.x; ob[1] = bb.y; ob[2] = bb.z; ob[3] = bb.w;
                out_scores[(long long)b * max_det + it] = __uint_as_float((unsigned)(k3 >> 32));
                if (in_lds) lsc[s3] = -1.f; else gw[s3].score = -1.f;
            }
        }
        __syncthreads();
        if (best_key == 0) break;      // (block-uniform)
        ++n_out;
        const float4 bb = best_box;
        key = 0;
        slot = -1;
        for (int i = threadIdx.x; i < C; i += kNmsThreads) {
            float sc = in_lds ? lsc[i] : gw[i].score;      // (the selected one was marked dead before the barrier above)
            if (sc > 0.f && nms_iou(in_lds ? lbox[i] : gw[i].box, bb) > iou_thr) {
                sc = -1.f;
                if (in_lds) lsc[i] = -1.f; else gw[i].score = -1.f;
            }
            if (sc > 0.f) {
                const int an = in_lds ? lan[i] : gw[i].anchor;
                const unsigned long long k2 = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)an);
                if (k2 > key) { key = k2; slot = i; }
            }
        }
        // (no barrier here: a thread reads and writes only its own candidates in this pass, and the next selection's barriers
        //  order wkey / best_key)
    }
    for (int i = n_out * 4 + threadIdx.x; i < max_det * 4; i += kNmsThreads) out_boxes[(long long)b * max_det * 4 + i] = 0.f;   // zero padding (nms.py:49-51)
    for (int i = n_out + threadIdx.x; i < max_det; i += kNmsThreads) out_scores[(long long)b * max_det + i] = 0.f;
    if (threadIdx.x == 0) out_num[b] = n_out;
}

int fill_levels(RetinaLevels& lv, const void* const* logits, const void* const* boxes, void* const* dlogits, void* const* dboxes,
                const int* h, const int* w) {
    int first = 0;
    for (int l = 0; l < kLevels; ++l) {
        MPN_REQUIRE(logits[l] && boxes[l] && h[l] > 0 && w[l] > 0, MPN_ERR_BAD_ARG, "retina: level %d: null pointer / bad size", l);
        lv.logits[l] = logits[l]; lv.boxes[l] = boxes[l];
        lv.dlogits[l] = dlogits ? dlogits[l] : nullptr;
        lv.dboxes[l] = dboxes ? dboxes[l] : nullptr;
        MPN_REQUIRE((lv.dlogits[l] == nullptr) == (lv.dboxes[l] == nullptr), MPN_ERR_BAD_ARG, "retina: gradient buffers of level %d", l);
        lv.hw[l] = h[l] * w[l];
        lv.first[l] = first;
        first += h[l] * w[l] * kAPL;
    }
    lv.first[kLevels] = first;
    return MPN_OK;
}

int stream_blocks(long long nvec) {
    long long b = (nvec + kThreads - 1) / kThreads;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int mpn_patchify3x3s2(const void* x, void* patches, int N, int H, int W, int C, int dtype, const float* in_scale,
                                 const float* in_shift, int in_act, mpn_stream_t stream) {
    MPN_REQUIRE(x && patches && N > 0 && H > 0 && W > 0, MPN_ERR_BAD_ARG, "patchify: bad arguments");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "patchify: dtype %d", dtype);
    const int ve = dtype == MPN_F32 ? 4 : 8;
    MPN_REQUIRE(C > 0 && C % ve == 0, MPN_ERR_BAD_SHAPE, "patchify: C (%d) must be a multiple of %d", C, ve);
    MPN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), MPN_ERR_BAD_ARG, "patchify: scale/shift mismatch");
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const long long nvec = (long long)N * OH * OW * 9 * (C / ve);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (patchify_kernel<T><<<stream_blocks(nvec), kThreads, 0, st>>>((const T*)x, (T*)patches, N, H, W, C, OH, OW,
                                                                                            in_scale, in_shift, in_act, nvec)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_unpatchify3x3s2(const void* dpatches, void* dx, int N, int H, int W, int C, int dtype, mpn_stream_t stream) {
    MPN_REQUIRE(dpatches && dx && N > 0 && H > 0 && W > 0, MPN_ERR_BAD_ARG, "unpatchify: bad arguments");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "unpatchify: dtype %d", dtype);
    const int ve = dtype == MPN_F32 ? 4 : 8;
    MPN_REQUIRE(C > 0 && C % ve == 0, MPN_ERR_BAD_SHAPE, "unpatchify: C (%d) must be a multiple of %d", C, ve);
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const long long nvec = (long long)N * H * W * (C / ve);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (unpatchify_kernel<T><<<stream_blocks(nvec), kThreads, 0, st>>>((const T*)dpatches, (T*)dx, N, H, W, C, OH, OW, nvec)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" size_t mpn_retina_match_workspace_bytes(int B, int max_boxes) { return (size_t)B * (max_boxes > 0 ? max_boxes : 1) * 8; }

extern "C" int mpn_retina_match(const float* anchors, const float* gt_boxes, const int* num_boxes, int B, int A, int max_boxes,
                                float positives_threshold, float negatives_threshold, int* matches, float* targets,
                                int* num_matched, void* workspace, size_t workspace_bytes, mpn_stream_t stream) {
    MPN_REQUIRE(anchors && gt_boxes && num_boxes && matches && targets && num_matched && workspace, MPN_ERR_BAD_ARG, "retina_match: null pointer");
    MPN_REQUIRE(B > 0 && A > 0 && max_boxes > 0 && max_boxes <= 2048, MPN_ERR_BAD_SHAPE, "retina_match: bad sizes");
    MPN_REQUIRE(positives_threshold >= negatives_threshold, MPN_ERR_BAD_ARG, "retina_match: thresholds");   // training_target_creation.py:83
    MPN_REQUIRE(workspace_bytes >= mpn_retina_match_workspace_bytes(B, max_boxes), MPN_ERR_WORKSPACE, "retina_match: workspace too small");
    MPN_REQUIRE(mpn_aligned16(anchors) && mpn_aligned16(gt_boxes) && mpn_aligned16(targets), MPN_ERR_BAD_ALIGN, "retina_match: alignment");
    hipStream_t st = (hipStream_t)stream;
    match_reset_kernel<<<(B * max_boxes + kThreads - 1) / kThreads, kThreads, 0, st>>>((unsigned long long*)workspace, B * max_boxes, num_matched);
    const dim3 grid((A + kThreads - 1) / kThreads, B);
    match_anchor_kernel<<<grid, kThreads, (size_t)max_boxes * 4 * sizeof(float), st>>>(
        anchors, gt_boxes, num_boxes, B, A, max_boxes, positives_threshold, negatives_threshold, matches, (unsigned long long*)workspace);
    match_forced_kernel<<<B, kThreads, 0, st>>>((const unsigned long long*)workspace, num_boxes, A, max_boxes, matches);
    match_targets_kernel<<<grid, kThreads, 0, st>>>(anchors, gt_boxes, matches, A, max_boxes, targets, num_matched);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

extern "C" int mpn_retina_loss_num_parts(int B, int A) { return B * ((A + kThreads - 1) / kThreads); }

extern "C" int mpn_retina_loss(const void* const* logits, const void* const* boxes, void* const* dlogits, void* const* dboxes,
                               const int* h, const int* w, int dtype, const float* cls_bias, const float* box_bias,
                               const int* matches, const float* targets, const int* num_matched, int B, float gamma, float alpha,
                               float localization_loss_weight, float classification_loss_weight, float* part,
                               mpn_stream_t stream) {
    MPN_REQUIRE(logits && boxes && h && w && cls_bias && box_bias && matches && targets && num_matched && part && B > 0, MPN_ERR_BAD_ARG,
                "retina_loss: bad arguments");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "retina_loss: dtype %d", dtype);
    RetinaLevels lv;
    if (int rc = fill_levels(lv, logits, boxes, dlogits, dboxes, h, w)) return rc;
    const int A = lv.first[kLevels];
    const dim3 grid((A + kThreads - 1) / kThreads, B);
    hipStream_t st = (hipStream_t)stream;
    MPN_DISPATCH_DTYPE(dtype, (retina_loss_kernel<T><<<grid, kThreads, 0, st>>>(lv, cls_bias, box_bias, matches, targets, num_matched, B, A, gamma,
                                                                              alpha, localization_loss_weight, classification_loss_weight, part)));
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

__global__ void retina_loss_finalize_kernel(const float* __restrict__ sums, const int* __restrict__ num_matched, float lw, float cw,
                                            float* __restrict__ losses, float* __restrict__ dbias_cls, float* __restrict__ dbias_box) {
    const int t = threadIdx.x;
    if (t == 0) {
        const int nm = num_matched[0];
        const float inv = 1.0f / (float)(nm > 1 ? nm : 1);      // tf.maximum(num_matches, 1) (retinanet.py:128-131)
        const float loc = sums[1] * inv, cls = sums[0] * inv;
        losses[0] = loc;
        losses[1] = cls;
        losses[3] = lw * loc + cw * cls + losses[2];
    }
    if (dbias_cls != nullptr && t < 6) dbias_cls[t] = sums[2 + t];
    if (dbias_box != nullptr && t < 24) dbias_box[t] = sums[8 + t];
}

extern "C" int mpn_retina_loss_finalize(const float* sums, const int* num_matched, float localization_loss_weight,
                                        float classification_loss_weight, float* losses, float* dbias_cls, float* dbias_box,
                                        mpn_stream_t stream) {
    MPN_REQUIRE(sums && num_matched && losses, MPN_ERR_BAD_ARG, "retina_loss_finalize: null pointer");
    MPN_REQUIRE((dbias_cls == nullptr) == (dbias_box == nullptr), MPN_ERR_BAD_ARG, "retina_loss_finalize: bias gradients");
    retina_loss_finalize_kernel<<<1, 64, 0, (hipStream_t)stream>>>(sums, num_matched, localization_loss_weight, classification_loss_weight,
                                                                   losses, dbias_cls, dbias_box);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}

// the candidate lists (32 bytes per anchor: every anchor may pass the score test) + one counter per image + the overflow word
extern "C" size_t mpn_retina_nms_workspace_bytes(int B, int A) { return (size_t)B * A * sizeof(NmsCand) + ((((size_t)B + 1) * sizeof(int) + 15) & ~(size_t)15); }
extern "C" size_t mpn_retina_nms_overflow_offset(int B, int A) { return (size_t)B * A * sizeof(NmsCand) + (size_t)B * sizeof(int); }

extern "C" int mpn_retina_nms(const void* const* logits, const void* const* boxes, const int* h, const int* w, int dtype,
                              const float* cls_bias, const float* box_bias, const float* anchors, int B, float score_threshold,
                              float iou_threshold, int max_detections, float* out_boxes, float* out_scores, int* out_num,
                              void* workspace, size_t workspace_bytes, mpn_stream_t stream) {
    MPN_REQUIRE(logits && boxes && h && w && cls_bias && box_bias && anchors && out_boxes && out_scores && out_num && workspace && B > 0 &&
                    max_detections > 0, MPN_ERR_BAD_ARG, "retina_nms: bad arguments");
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16, MPN_ERR_BAD_DTYPE, "retina_nms: dtype %d", dtype);
    RetinaLevels lv;
    if (int rc = fill_levels(lv, logits, boxes, nullptr, nullptr, h, w)) return rc;
    const int A = lv.first[kLevels];
    MPN_REQUIRE(workspace_bytes >= mpn_retina_nms_workspace_bytes(B, A), MPN_ERR_WORKSPACE, "retina_nms: workspace too small");
    MPN_REQUIRE(mpn_aligned16(workspace) && mpn_aligned16(anchors), MPN_ERR_BAD_ALIGN, "retina_nms: alignment");
    NmsCand* cands = reinterpret_cast<NmsCand*>(workspace);
    int* counts = reinterpret_cast<int*>(cands + (size_t)B * A);
    hipStream_t st = (hipStream_t)stream;
    // (a launch rather than hipMemsetAsync - round 5's device fault: the second store of an append in retina_candidates_kernel, replayed
    //  from a hipGraph; what was observed and what is inferred about the memset node: DESIGN section 2)
    nms_reset_kernel<<<(B + 1 + 63) / 64, 64, 0, st>>>(counts, B);
    MPN_LAUNCH_CHECK();
    const dim3 grid((unsigned)((A + kThreads - 1) / kThreads), (unsigned)B);
    MPN_DISPATCH_DTYPE(dtype, (retina_candidates_kernel<T><<<grid, kThreads, 0, st>>>(lv, cls_bias, box_bias, anchors, A, score_threshold, cands, counts)));
    MPN_LAUNCH_CHECK();
    constexpr int kNmsSmem = kNmsLds * 24;
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)retina_nms_kernel, kNmsSmem, &attr_mask));
    retina_nms_kernel<<<B, kNmsThreads, kNmsSmem, st>>>(cands, counts, A, iou_threshold, max_detections, out_boxes, out_scores, out_num);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
