/* libmpn_hip.so - C ABI of the MI355X-native keypoint hot path of MultiPoseNet.
 *
 * The reference (TropComplique/MultiPoseNet) has no FFI of its own: the hot path sits
 * behind plain Python callables that delegate all arithmetic to TensorFlow 1.15 ops.
 * Each entry point below replaces the TF op call sites listed in its comment
 * (paths relative to the reference root), so a maintainer can bind them with ctypes
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host";
 *   - activations are NHWC (channels innermost). The reference computes in NCHW
 *     (detector/constants.py:7) but its API edge is NHWC (images in, heatmaps out);
 *   - `dtype` selects activation storage: MPN_F32 or MPN_BF16 (accumulation is f32);
 *   - parameters, batch-norm statistics, gradients and optimizer state are f32;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never
 *     synchronises, never allocates: scratch memory comes in through `workspace`;
 *   - return value: 0 = ok, negative = MPN_ERR_*; mpn_last_error() has the text.
 */
#ifndef MPN_H_
#define MPN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPN_VERSION 100

enum { MPN_F32 = 0, MPN_BF16 = 1, MPN_F16 = 2 /* decode input only */ };

enum {
    MPN_OK = 0,
    MPN_ERR_BAD_SHAPE = -1,
    MPN_ERR_BAD_DTYPE = -2,
    MPN_ERR_BAD_ALIGN = -3,
    MPN_ERR_HIP = -4,
    MPN_ERR_BAD_ARG = -5,
    MPN_ERR_WORKSPACE = -6
};

/* activation applied after the batch-norm affine */
enum { MPN_ACT_NONE = 0, MPN_ACT_RELU = 1, MPN_ACT_RELU6 = 2 };

typedef void* mpn_stream_t; /* hipStream_t */

int mpn_version(void);
/* copies the calling thread's last error message into buf (host), returns its length */
int mpn_last_error(char* buf, size_t n);

/* ------------------------------------------------------------------------------------
 * K14  heatmap peak decode.
 * Replaces inference/utils.py:29-52 `get_keypoints` (numpy: per-channel max > threshold,
 * first-occurrence argmax, scale into the box, int truncation) for a whole batch, and
 * the tie rule of create_pb.py:120-142 `argmax_2d` (smallest flat index).
 *
 *   heatmaps  [B,h,w,C] NHWC, C == 17, dtype f32 / bf16 / f16
 *   box_hw    [B,2] f64: (height, width) = (ymax-ymin, xmax-xmin) of each image's box
 *   threshold compared as `max > threshold` in f32 (the shim rounds it the way numpy does)
 *   out_xyv   [B,C,3] int32 (x, y, visible) - zeros where the channel is skipped
 *   out_score [B,C] f32 per-channel max (NaN if the channel holds a NaN)   (may be NULL)
 *   out_index [B,C] int32 flat argmax y*w+x of every channel               (may be NULL)
 *   workspace mpn_heatmap_decode_workspace_bytes(B) bytes, ZERO-FILLED once by the
 *             caller when it is allocated; the kernel leaves it zeroed again.
 */
size_t mpn_heatmap_decode_workspace_bytes(int B);
int mpn_heatmap_decode(const void* heatmaps, int dtype, int B, int h, int w, int C,
                       const double* box_hw, float threshold,
                       int32_t* out_xyv, float* out_score, int32_t* out_index,
                       void* workspace, size_t workspace_bytes, mpn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MPN_H_ */
