// K14: heatmap peak decode (replaces inference/utils.py:29-52 get_keypoints, batched).
//
// HBM-bound: each image is h*w*17 interleaved values read exactly once.
//   * a block owns a run of pixels of one image; every iteration stages 256 pixels
//     (256*17 values) into LDS with fully coalesced 16-byte loads, then lane p reads the 17
//     channels of "its" pixel (LDS stride 17 dwords: conflict-free) and updates 17 running
//     (ordered-value, first-index) pairs held in registers;
//   * ordering key = (monotone u32 image of the f32 value, ~flat_index) so that a u64 max
//     picks the largest value and, on ties, the SMALLEST flat index (numpy argmax /
//     tf.argmax rule). -0.0 is canonicalised to +0.0 (numpy compares by value); a NaN maps
//     to the top key so that, like numpy's max(), it poisons its channel;
//   * block reduction through LDS + wave shuffles, one atomicMax(u64) per (block, channel),
//     then the last block of each image (ticket counter) reads the 17 keys back with
//     returning atomics, resets the workspace and writes the outputs.
#include "common.h"

namespace {

constexpr int kC = 17;
constexpr int kThreads = 256;
constexpr int kPix = 256;  // pixels staged per iteration

__device__ __forceinline__ unsigned ordered_key(float v) {
    v = v + 0.0f;  // -0.0 -> +0.0
    const unsigned u = __float_as_uint(v);
    const unsigned m = (unsigned)((int)u >> 31) | 0x80000000u;
    return (v != v) ? 0xffffffffu : (u ^ m);
}
__device__ __forceinline__ float key_to_float(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(u);
}

template <typename T> __device__ __forceinline__ float lds_elem(const unsigned char* base, int i);
template <> __device__ __forceinline__ float lds_elem<float>(const unsigned char* base, int i) {
    return reinterpret_cast<const float*>(base)[i];
}
template <> __device__ __forceinline__ float lds_elem<bf16_t>(const unsigned char* base, int i) {
    return (float)reinterpret_cast<const bf16_t*>(base)[i];
}
template <> __device__ __forceinline__ float lds_elem<_Float16>(const unsigned char* base, int i) {
    return (float)reinterpret_cast<const _Float16*>(base)[i];
}

struct DecodeWs {
    unsigned long long* keys;  // [B][32]
    unsigned* tickets;         // [B]
};

template <typename T>
__global__ __launch_bounds__(kThreads) void decode_kernel(
    const unsigned char* __restrict__ hm, long long total_bytes, int h, int w, int splits,
    int chunks_per_block, const double* __restrict__ box_hw, float threshold,
    int* __restrict__ out_xyv, float* __restrict__ out_score, int* __restrict__ out_index,
    unsigned long long* __restrict__ gkeys, unsigned* __restrict__ tickets) {
    constexpr int ES = (int)sizeof(T);
    // staging buffer: 256 px * 17 * ES bytes + 16 bytes of alignment slack on both sides
    __shared__ __attribute__((aligned(16))) unsigned char stage[kPix * kC * ES + 32];
    __shared__ unsigned long long red[kC][kThreads];
    __shared__ int is_last;

    const int b = blockIdx.x / splits;
    const int s = blockIdx.x - b * splits;
    const int tid = threadIdx.x;
    const int npix = h * w;
    const long long img_byte0 = (long long)b * npix * kC * ES;

    unsigned best_hi[kC];
    unsigned best_idx[kC];
#pragma unroll
    for (int c = 0; c < kC; ++c) { best_hi[c] = 0u; best_idx[c] = 0xffffffffu; }

    const int pix_begin = s * chunks_per_block * kPix;
    const int pix_end = min(npix, pix_begin + chunks_per_block * kPix);

    if constexpr (ES == 4) {
        // f32: every lane streams "its" pixels straight from HBM - 68 contiguous bytes (4 x dwordx4 + 1 dword, dword
        // aligned) per pixel, a wave covers 4352 contiguous bytes per pixel row so every fetched byte is used - two
        // pixels in flight per lane, no LDS staging and no barrier in the streaming loop.
        const float* __restrict__ img = reinterpret_cast<const float*>(hm + img_byte0);
        // (two pixels in flight per lane; four measured slower: 11.8 vs 11.1 us at B = 32, 48.7 vs 44.7 at B = 256)
        constexpr int U = 2;
        for (int p0 = pix_begin + tid; p0 < pix_end; p0 += U * kThreads) {
            float v[U][kC + 3];
            int px[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                px[u] = p0 + u * kThreads;
                const float* sp = img + (long long)(px[u] < pix_end ? px[u] : p0) * kC;      // (unconditional loads from a valid pixel)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = *reinterpret_cast<const float4*>(sp + 4 * q);
                    v[u][4 * q] = a.x; v[u][4 * q + 1] = a.y; v[u][4 * q + 2] = a.z; v[u][4 * q + 3] = a.w;
                }
                v[u][16] = sp[16];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (px[u] < pix_end) {
#pragma unroll
                    for (int c = 0; c < kC; ++c) {
                        const unsigned k = ordered_key(v[u][c]);
                        const bool gt = k > best_hi[c];  // strict: keeps the first occurrence (pixels ascend per lane)
                        best_hi[c] = gt ? k : best_hi[c];
                        best_idx[c] = gt ? (unsigned)px[u] : best_idx[c];
                    }
                }
            }
        }
    } else {
    for (int p0 = pix_begin; p0 < pix_end; p0 += kPix) {
        const int np = min(kPix, pix_end - p0);
        const long long byte0 = img_byte0 + (long long)p0 * kC * ES;
        const long long byte1 = byte0 + (long long)np * kC * ES;
        const long long a0 = byte0 & ~15ll;  // hipMalloc base is >=256-B aligned
        const int head = (int)(byte0 - a0);
        const int nvec = (int)((byte1 - a0 + 15) >> 4);
        __syncthreads();  // previous iteration's LDS reads are done
        for (int v = tid; v < nvec; v += kThreads) {
            const long long off = a0 + ((long long)v << 4);
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (off + 16 <= total_bytes) {
                val = *reinterpret_cast<const uint4*>(hm + off);
            } else {  // ragged tail of the whole tensor: byte-wise, never reads past the end
                unsigned char tmp[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) tmp[i] = (off + i < total_bytes) ? hm[off + i] : (unsigned char)0;
                val = *reinterpret_cast<const uint4*>(tmp);
            }
            *reinterpret_cast<uint4*>(stage + ((size_t)v << 4)) = val;
        }
        __syncthreads();
        if (tid < np) {
            const unsigned pidx = (unsigned)(p0 + tid);
            const unsigned char* base = stage + head;  // head is a multiple of ES
#pragma unroll
            for (int c = 0; c < kC; ++c) {
                const float v = lds_elem<T>(base, tid * kC + c);
                const unsigned k = ordered_key(v);
                const bool gt = k > best_hi[c];  // strict: keeps the first occurrence
                best_hi[c] = gt ? k : best_hi[c];
                best_idx[c] = gt ? pidx : best_idx[c];
            }
        }
    }
    }

#pragma unroll
    for (int c = 0; c < kC; ++c)
        red[c][tid] = ((unsigned long long)best_hi[c] << 32) | (unsigned long long)(~best_idx[c]);
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    for (int c = wave; c < kC; c += kThreads / 64) {
        unsigned long long k = red[c][lane];
#pragma unroll
        for (int j = 1; j < kThreads / 64; ++j) {
            const unsigned long long o = red[c][lane + 64 * j];
            k = o > k ? o : k;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)k, o, 64);
            const unsigned hi = __shfl_xor((unsigned)(k >> 32), o, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            k = other > k ? other : k;
        }
        if (lane == 0 && k != 0ull) atomicMax(&gkeys[b * 32 + c], k);
    }
    // every atomic of this block has been performed before the ticket is drawn
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = atomicAdd(&tickets[b], 1u);
        is_last = (t == (unsigned)(splits - 1));
    }
    __syncthreads();
    if (!is_last) return;

    if (tid < kC) {
        const int c = tid;
        // returning atomics: coherent with the other blocks' atomicMax wherever it executed;
        // they also leave the workspace zeroed for the next call.
        const unsigned long long k = atomicExch(&gkeys[b * 32 + c], 0ull);
        const unsigned hi = (unsigned)(k >> 32);
        const unsigned idx = ~(unsigned)k;
        const bool has_nan = (hi == 0xffffffffu);
        const float score = has_nan ? __uint_as_float(0x7fc00000u) : key_to_float(hi);
        int x = 0, y = 0, vis = 0;
        if (!has_nan && score > threshold) {
            const double height = box_hw[2 * b], width = box_hw[2 * b + 1];
            const int yi = (int)(idx / (unsigned)w), xi = (int)(idx % (unsigned)w);
            // utils.py:48-49: np.clip(int(y * height / h), 0, height), then stored as int32
            const double qy = trunc((double)yi * height / (double)h);
            const double qx = trunc((double)xi * width / (double)w);
            y = (int)fmin(fmax(qy, 0.0), height);
            x = (int)fmin(fmax(qx, 0.0), width);
            vis = 1;
        }
        const int o = b * kC + c;
        out_xyv[3 * o + 0] = x;
        out_xyv[3 * o + 1] = y;
        out_xyv[3 * o + 2] = vis;
        if (out_score) out_score[o] = score;
        if (out_index) out_index[o] = (int)idx;
    }
    if (tid == 0) atomicExch(&tickets[b], 0u);
}

}  // namespace

extern "C" size_t mpn_heatmap_decode_workspace_bytes(int B) {
    if (B <= 0) return 0;
    return (size_t)B * 32 * sizeof(unsigned long long) + (((size_t)B * sizeof(unsigned) + 15) & ~(size_t)15);
}

extern "C" int mpn_heatmap_decode(const void* heatmaps, int dtype, int B, int h, int w, int C,
                                  const double* box_hw, float threshold, int32_t* out_xyv,
                                  float* out_score, int32_t* out_index, void* workspace,
                                  size_t workspace_bytes, mpn_stream_t stream) {
    MPN_REQUIRE(C == kC, MPN_ERR_BAD_SHAPE, "decode: C must be 17 (got %d)", C);
    MPN_REQUIRE(B >= 0 && h > 0 && w > 0, MPN_ERR_BAD_SHAPE, "decode: bad shape B=%d h=%d w=%d", B, h, w);
    MPN_REQUIRE((long long)h * w < (1ll << 31) - 1, MPN_ERR_BAD_SHAPE, "decode: h*w too large");
    if (B == 0) return MPN_OK;
    MPN_REQUIRE(heatmaps && box_hw && out_xyv && workspace, MPN_ERR_BAD_ARG, "decode: null pointer");
    MPN_REQUIRE(mpn_aligned16(heatmaps) && mpn_aligned16(workspace), MPN_ERR_BAD_ALIGN,
                "decode: heatmaps/workspace must be 16-byte aligned");
    MPN_REQUIRE(workspace_bytes >= mpn_heatmap_decode_workspace_bytes(B), MPN_ERR_WORKSPACE,
                "decode: workspace too small (%zu < %zu)", workspace_bytes,
                mpn_heatmap_decode_workspace_bytes(B));
    const int es = dtype == MPN_F32 ? 4 : 2;
    MPN_REQUIRE(dtype == MPN_F32 || dtype == MPN_BF16 || dtype == MPN_F16, MPN_ERR_BAD_DTYPE,
                "decode: unsupported dtype %d", dtype);
    const int npix = h * w;
    const int nchunks = mpn_div_up(npix, kPix);
    // blocks per image. The tail of a launch is a chain of dependent atomics (17 atomicMax per block, a ticket, the last block's
    // exchanges) whose cost grows with the blocks that contend: measured at 128 x 128 (us per launch by total blocks) B = 1: 10.0
    // (32) / 12.8 (64); B = 32: 11.1 (256) / 11.9 (512) / 17.5 (1024) / 34.9 (2048); B = 256: 44.7 (512) / 46.7 (1024) / 62 (256)
    int target = 8 * B;
    if (target < 32) target = 32;
    if (target > 512) target = 512;
    int splits = (target + B - 1) / B;
    if (splits < 1) splits = 1;
    if (splits > nchunks) splits = nchunks;
    const int cpb = mpn_div_up(nchunks, splits);
    splits = mpn_div_up(nchunks, cpb);
    const long long total_bytes = (long long)B * npix * kC * es;

    unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);
    unsigned* tickets = reinterpret_cast<unsigned*>(keys + (size_t)B * 32);
    const dim3 grid((unsigned)(B * splits)), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
    const unsigned char* hm = reinterpret_cast<const unsigned char*>(heatmaps);
    if (dtype == MPN_F32)
        decode_kernel<float><<<grid, block, 0, st>>>(hm, total_bytes, h, w, splits, cpb, box_hw, threshold,
                                                     out_xyv, out_score, out_index, keys, tickets);
    else if (dtype == MPN_BF16)
        decode_kernel<bf16_t><<<grid, block, 0, st>>>(hm, total_bytes, h, w, splits, cpb, box_hw, threshold,
                                                      out_xyv, out_score, out_index, keys, tickets);
    else
        decode_kernel<_Float16><<<grid, block, 0, st>>>(hm, total_bytes, h, w, splits, cpb, box_hw, threshold,
                                                        out_xyv, out_score, out_index, keys, tickets);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
