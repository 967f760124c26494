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

#define MPN_VERSION 600   /* r6: see INTEGRATION.md "ABI revisions" */

enum { MPN_F32 = 0, MPN_BF16 = 1, MPN_F16 = 2 /* dense 1x1 / 3x3 convolutions (forward, weight gradient, pack), the PRN entry points and the decode input; the BN / depthwise / loss kernels of the keypoint step take F32 and BF16 only */ };

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

/* ------------------------------------------------------------------------------------
 * K5/K6/K8  dense convolution on the MFMA matrix cores (NHWC, stride 1; 3x3 uses pad 1).
 * Replaces slim.conv2d 1x1 (detector/backbones/mobilenet_v1.py:73) and conv2d_same
 * (detector/utils/layer_utils.py:19-39; call sites detector/fpn.py:38,39,50,52 and
 * detector/keypoint_subnet.py:38,75,77). The same kernel computes data-gradients when it is
 * given weights packed with transpose=1 (flipped taps, Cin/Cout swapped).
 *
 *   x        [N,H,W,Cin]  raw output of the producing conv (or any tensor)
 *   in_scale, in_shift [Cin] f32 or NULL: the producer's batch-norm affine, applied on load
 *            together with in_act (MPN_ACT_*), so normalised activations never touch HBM
 *   w_packed weights from mpn_conv_pack_weights (same dtype as x)
 *   y        [N,H,W,Cout]
 *   stats_part NULL or [mpn_conv_num_parts()][2][Cout] f32: partial sums and sums of squares of y for the
 *            following batch-norm; mpn_conv_stats_rows() rows are written (mpn_bn_finalize reduces them)
 *   up_res   NULL or [N,H/2,W/2,Cout]: y += nearest-2x-upsample(up_res)  (detector/fpn.py:51,58-76)
 */
size_t mpn_conv_packed_bytes(int Cin, int Cout, int ksize, int transpose, int dtype);
/* w_hwio: f32 [ksize,ksize,Cin,Cout] in the reference's variable layout (HWIO) */
int mpn_conv_pack_weights(const float* w_hwio, int Cin, int Cout, int ksize, int transpose,
                          int dtype, void* out, mpn_stream_t stream);
/* Batched packing (one launch for every conv of the network after an optimizer step): fill one host descriptor
 * per (conv, direction) with mpn_conv_pack_desc_fill (returns the job's block count; block_begin = running sum),
 * copy the array to the device once, then launch mpn_conv_pack_weights_batched each step. */
size_t mpn_conv_pack_desc_bytes(void);
int mpn_conv_pack_desc_fill(void* desc_host, const float* w_hwio, int Cin, int Cout, int ksize, int transpose,
                            int dtype, void* out, int block_begin);
int mpn_conv_pack_weights_batched(const void* descs_device, int ndesc, int total_blocks, int dtype,
                                  mpn_stream_t stream);
/* rows to SIZE a statistics slab with: one per 8 x 16-pixel tile (3x3) / per 128 pixels (1x1) */
int mpn_conv_num_parts(int N, int H, int W, int ksize);
/* rows a convolution of this shape WRITES (what mpn_bn_finalize / the finalize descriptors must be given as nparts): the persistent
 * 3x3 kernel (16-bit storage, Cin % 64 == 0, Cout % 64 == 0) sums the rows of a block's tiles and writes one row per block - the same
 * count alone and inside a group -, every other kernel mpn_conv_num_parts rows. Never more than mpn_conv_num_parts; < 0 without a device. */
int mpn_conv_stats_rows(int N, int H, int W, int Cin, int Cout, int ksize, int dtype);
/* x_stride / y_stride: elements between consecutive pixels of x / y; 0 = dense (Cin / Cout). A larger stride reads /
 * writes a channel slice of a wider NHWC tensor in place - phi_subnet_2's second conv writes straight into the first 128
 * channels of the 512-channel concat tensor (keypoint_subnet.py:37: tf.concat with upsample factor 1 is a copy). */
int mpn_conv_fwd(const void* x, const void* w_packed, void* y, int N, int H, int W, int Cin,
                 int Cout, int x_stride, int y_stride, int ksize, int dtype, const float* in_scale,
                 const float* in_shift, int in_act, float* stats_part, const void* up_res,
                 mpn_stream_t stream);
/* Up to five independent 3x3 convolutions of the same channel geometry in ONE grid, largest first (the four pyramid levels
 * of a keypoint-subnet stage, keypoint_subnet.py:64-91: as launches of their own the small levels are latency-bound tails
 * of 15-45 us). Per job: x, w_packed, y, H, W, x_stride / y_stride (arrays or NULL = dense), in_scale / in_shift (or NULL), stats_part
 * (or NULL); shared: N, Cin, Cout, ksize, dtype, in_act. Results are those of mpn_conv_fwd per job, bit for bit;
 * configurations the grouped grid does not cover (f32, 1x1, more than five jobs) run as the separate launches they replace. */
int mpn_conv_fwd_grouped(int njobs, const void* const* x, const void* const* w_packed, void* const* y, int N, const int* H,
                         const int* W, int Cin, int Cout, const int* x_stride, const int* y_stride, int ksize, int dtype,
                         const float* const* in_scale, const float* const* in_shift, int in_act,
                         float* const* stats_part, mpn_stream_t stream);
/* Data gradients of up to five independent 3x3 convolutions in ONE grid (as mpn_conv_fwd_grouped on the transposed packed
 * weights) that ALSO do the first pass of the batch-norm backward of the layer they feed (keypoint_subnet.py:75-78: conv ->
 * bn -> relu -> conv; layer_utils.py:9-16): dx[j] is written MASKED - g = dx where lo < bn_x[j] * bn_scale[j] + bn_shift[j] < hi
 * (the activation bn_act passed), else 0, so mpn_bn_bwd_apply's own mask is a no-op on it - and part[j]
 * ([mpn_conv_num_parts(N,H[j],W[j],3)][2][C] floats, mpn_conv_stats_rows of them written) receives the partial sums of g and of g * bn_x with the RAW x: finish
 * with a finalize built by mpn_bn_bwd_fin_desc_fill_raw. One tensor read and one launch less than mpn_bn_bwd_reduce behind
 * the data gradient. dy [N,H,W,K], dx / bn_x [N,H,W,C] (pixel strides as arrays or NULL = dense); 16-bit storage,
 * K % 64 == 0, K <= 512, C % 64 == 0, C <= 512: mpn_conv_bwd_data_bn_supported(K, C, ksize, dtype) != 0. */
int mpn_conv_bwd_data_bn_supported(int K, int C, int ksize, int dtype);
/* One layer: 3x3 as above, or a 1x1 layer (bf16, K and C multiples of 8: the data gradients of Conv2d_1..13_pointwise, which
 * feed the depthwise layers' batch-norms, mobilenet_v1.py:66-74, and of the FPN's lateral of c5, fpn.py:38) - through the GEMM
 * kernel where mpn_conv_fwd routes the geometry there, else the tiled kernel. part: [mpn_conv_num_parts(N,H,W,ksize)][2][C] (mpn_conv_stats_rows written);
 * finish with mpn_bn_bwd_finalize_raw. */
int mpn_conv_bwd_data_bn(const void* dy, const void* w_packed_t, void* dx, int N, int H, int W, int K, int C, int dy_stride,
                         int dx_stride, int ksize, int dtype, const void* bn_x, int bn_x_stride, const float* bn_scale,
                         const float* bn_shift, int bn_act, float* part, mpn_stream_t stream);
int mpn_conv_bwd_data_bn_grouped(int njobs, const void* const* dy, const void* const* w_packed_t, void* const* dx, int N,
                                 const int* H, const int* W, int K, int C, const int* dy_stride, const int* dx_stride,
                                 int dtype, const void* const* bn_x, const int* bn_x_stride, const float* const* bn_scale,
                                 const float* const* bn_shift, int bn_act, float* const* part, mpn_stream_t stream);

/* Weight gradient of mpn_conv_fwd: dW[tap][ci][co] = sum_pixels act(bn(x))[pixel+tap][ci]*dy[pixel][co].
 * Split-K over pixel tiles: part [mpn_conv_wgrad_num_parts()][ksize*ksize][Cin][Cout] f32 (one HWIO slab
 * per split), summed in a fixed order by mpn_reduce_partials. */
int mpn_conv_wgrad_num_parts(int N, int H, int W, int Cin, int Cout, int ksize, int dtype);
/* x_stride / dy_stride: pixel strides in elements, 0 = dense (channel slices of wider tensors, as for mpn_conv_fwd) */
int mpn_conv_bwd_weight(const void* x, const void* dy, float* part, int N, int H, int W, int Cin,
                        int Cout, int x_stride, int dy_stride, int ksize, int dtype, const float* in_scale,
                        const float* in_shift, int in_act, mpn_stream_t stream);

/* A thin 1x1 convolution's backward in ONE pass over x and dy (Cin <= 128, Cout <= 128, bf16: Conv2d_1..3_pointwise,
 * /root/reference/detector/backbones/mobilenet_v1.py:66-74: tf.gradients of slim.conv2d w.r.t. its kernel and its input, and the
 * reduction of the batch-norm below, mobilenet_v1.py:29-38): wpart [mpn_conv_wgrad_num_parts(N,H,W,Cin,Cout,1,dtype)][Cin][Cout] =
 * weight-gradient partials over act(x * in_scale + in_shift) (finish with mpn_reduce_partials); dx [N,H,W,Cin] = dy . w^T MASKED by that
 * activation (lo < x * in_scale + in_shift < hi on the raw x); bn_part [same rows][2][Cin] = partial sums of the masked gradient g and
 * of g * x with the RAW x (finish with mpn_bn_bwd_finalize_raw). What mpn_conv_bwd_weight + mpn_conv_bwd_data_bn give in two passes over
 * both tensors. bn_part == NULL: no reduction and dx is the plain, unmasked data gradient (an FPN lateral, /root/reference/detector/fpn.py:36-47).
 * w: the layer's f32 kernel [Cin][Cout] (HWIO of a 1x1). Strides in elements, 0 = dense. dx must not alias x or dy. */
int mpn_conv1x1_bwd_fused_supported(int Cin, int Cout, int dtype);
int mpn_conv1x1_bwd_fused(const void* x, const void* dy, const float* w, void* dx, float* wpart, float* bn_part, int N, int H, int W,
                          int Cin, int Cout, int x_stride, int dy_stride, int dx_stride, int dtype, const float* in_scale,
                          const float* in_shift, int in_act, mpn_stream_t stream);
/* ... with the batch-norm backward APPLY pass of the layer's OWN batch-norm folded into the staging of dY (Cin <= 64, Cout <= 128):
 * g = the gradient w.r.t. the layer's activated output (what mpn_bn_bwd_apply would turn into dy in place),
 * y_raw = the layer's raw output, ap_* that batch-norm's affine, saved statistics and the k1 / k2 of mpn_bn_bwd_finalize. The slabs and dx of
 * mpn_bn_bwd_apply followed by mpn_conv1x1_bwd_fused (to the storage rounding of a rare staged element); g and y_raw are not written. */
int mpn_conv1x1_bwd_fused_apply_supported(int Cin, int Cout, int dtype);
int mpn_conv1x1_bwd_fused_apply(const void* x, const void* g, const void* y_raw, const float* w, void* dx, float* wpart, float* bn_part,
                                int N, int H, int W, int Cin, int Cout, int x_stride, int g_stride, int y_stride, int dx_stride,
                                int dtype, const float* in_scale, const float* in_shift, int in_act, const float* ap_scale,
                                const float* ap_shift, const float* ap_mean, const float* ap_invstd, const float* ap_k1,
                                const float* ap_k2, int ap_act, mpn_stream_t stream);
/* The weight gradients of njobs independent layers of one (Cin, Cout, ksize, dtype) in ONE grid - the pyramid levels of a
 * subnet stage (keypoint_subnet.py:66-79: one phi_subnet per level; fpn.py:38-52: one 3x3 per level). The 256 blocks are
 * divided among the jobs by their pixel counts, so a stage leaves 128 partial slabs in all instead of 128 per level, and the
 * small levels stop being latency-bound launches of their own. mpn_conv_wgrad_grouped_num_parts fills nparts[j] (the slab
 * count of job j in THAT grid; f32 / more than 5 jobs: the counts of the separate launches, which the grouped call then
 * performs); part[j]: [nparts[j]][ksize*ksize][Cin][Cout] f32. Arrays of njobs entries; x_stride / dy_stride may be NULL. */
int mpn_conv_wgrad_grouped_num_parts(int njobs, int N, const int* H, const int* W, int Cin, int Cout, int ksize, int dtype,
                                     int* nparts);
int mpn_conv_bwd_weight_grouped(int njobs, const void* const* x, const void* const* dy, float* const* part, int N,
                                const int* H, const int* W, int Cin, int Cout, const int* x_stride, const int* dy_stride,
                                int ksize, int dtype, const float* const* in_scale, const float* const* in_shift,
                                int in_act, mpn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K4  batch normalisation (tf.layers.batch_normalization(fused=True, momentum=.95, eps=1e-3):
 * detector/backbones/mobilenet_v1.py:29-38, detector/utils/layer_utils.py:9-16).
 * Producers emit partial sums [nparts][2][C] (sum, sum of squares); mpn_bn_finalize turns them
 * into the per-channel affine (scale = gamma*invstd, shift = beta - mean*scale) that CONSUMERS
 * apply on load, and updates the moving statistics (unbiased variance, TF-1.15 semantic) when
 * moving_mean/moving_var are non-NULL. x is viewed as [M rows][C channels] (NHWC).
 */
/* (The finalizes take `part` as SCRATCH: from 4096 partial rows on they first add groups of 32 consecutive rows in place -
 * a 1x1 layer at 256 x 256 leaves 16 384 rows, a 45-55 us finalize in a handful of blocks otherwise. The slab's contents are
 * consumed; results do not depend on the launch (fixed orders, no atomics).) */
int mpn_bn_stats_num_parts(long long M);
int mpn_bn_stats(const void* x, long long M, int C, int dtype, float* part, mpn_stream_t stream);
int mpn_bn_finalize(float* part, int nparts, int C, long long count, const float* gamma,
                    const float* beta, float* moving_mean, float* moving_var, float momentum,
                    float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                    mpn_stream_t stream);
/* is_training=False path: affine from the moving statistics */
/* The two backward passes of up to five independent layers of one channel count in ONE grid each, largest first (the
 * pyramid levels of a subnet stage). Arrays per job; results are those of mpn_bn_bwd_reduce / mpn_bn_bwd_apply per layer,
 * bit for bit; part[j] holds mpn_bn_stats_num_parts(M[j]) rows. dA_stride / x_stride (arrays or NULL = dense): elements
 * between consecutive rows of dA[j] / x[j] - the level-2 tensors of phi_subnet's second batch-norm live in channel slices
 * of the 512-channel concat tensor and of its gradient (keypoint_subnet.py:37). */
int mpn_bn_bwd_reduce_grouped(int njobs, void* const* dA, const void* const* x, const long long* M, int C, int dtype,
                              const float* const* scale, const float* const* shift, const float* const* mean,
                              const float* const* invstd, int act, float* const* part, const int* dA_stride,
                              const int* x_stride, mpn_stream_t stream);
int mpn_bn_bwd_apply_grouped(int njobs, void* const* dA, const void* const* x, const long long* M, int C, int dtype,
                             const float* const* scale, const float* const* shift, const float* const* mean,
                             const float* const* invstd, const float* const* k1, const float* const* k2, int act,
                             const float* const* add_ch0, const int* dA_stride, const int* x_stride,
                             mpn_stream_t stream);
/* Several independent layers' finalizes in ONE launch (the four pyramid levels of the keypoint subnet,
 * keypoint_subnet.py:64-91, produce their statistics side by side). Descriptor tables as for the batched slab reduction:
 * mpn_bn_fin_desc_fill / mpn_bn_bwd_fin_desc_fill write one host-side descriptor each (mpn_*_desc_bytes() bytes; return the
 * number of blocks of the job, -1 on bad arguments; block_begin = running sum); the caller copies the array to the device
 * once. Same arithmetic, bit for bit, as mpn_bn_finalize / mpn_bn_bwd_finalize per layer BELOW 4096 partial rows; from 4096 rows
 * on the per-layer entry points first add groups of 32 rows in place (f64 inside a group, the group's sum rounded to f32: `part`
 * is scratch and is DESTROYED - finalizing the same slab twice double-counts), which the batched kernels do not: the two then
 * agree to f32 rounding of the group sums (tests/test_ops_bwd_gpu.py::test_batched_and_per_layer_finalize_from_4096_rows). */
size_t mpn_bn_fin_desc_bytes(void);
size_t mpn_bn_bwd_fin_desc_bytes(void);
int mpn_bn_fin_desc_fill(void* desc_host, const float* part, int nparts, int C, long long count, const float* gamma,
                         const float* beta, float* moving_mean, float* moving_var, float* scale, float* shift,
                         float* save_mean, float* save_invstd, int block_begin);
int mpn_bn_bwd_fin_desc_fill(void* desc_host, const float* part, int nparts, int C, long long count, float* dgamma,
                             float* dbeta, float* k1, float* k2, int block_begin);
/* The same for a slab whose second row holds sum g * x with the RAW x (mpn_conv_bwd_data_bn_grouped): mean / invstd = the
 * layer's saved batch statistics; the finalize forms sum g * xhat = invstd * (sum g x - mean * sum g) in f64. */
int mpn_bn_bwd_fin_desc_fill_raw(void* desc_host, const float* part, int nparts, int C, long long count, float* dgamma,
                                 float* dbeta, float* k1, float* k2, const float* mean, const float* invstd,
                                 int block_begin);
int mpn_bn_finalize_batched(const void* descs_device, int ndesc, int total_blocks, float momentum, float eps,
                            mpn_stream_t stream);
int mpn_bn_bwd_finalize_batched(const void* descs_device, int ndesc, int total_blocks, mpn_stream_t stream);
int mpn_bn_inference_affine(int C, const float* gamma, const float* beta, const float* moving_mean,
                            const float* moving_var, float eps, float* scale, float* shift,
                            mpn_stream_t stream);
/* y = act(x*scale + shift), materialised (only needed at the API edge) */
int mpn_bn_act_apply(const void* x, void* y, long long M, int C, int dtype, const float* scale,
                     const float* shift, int act, mpn_stream_t stream);
/* backward: g = dA*act'(.), partials of sum(g), sum(g*xhat) -> [mpn_bn_stats_num_parts(M)][2][C] */
int mpn_bn_bwd_reduce(const void* dA, const void* x, long long M, int C, int dtype,
                      const float* scale, const float* shift, const float* mean,
                      const float* invstd, int act, float* part, mpn_stream_t stream);
int mpn_bn_bwd_finalize(float* part, int nparts, int C, long long count, float* dgamma,
                        float* dbeta, float* k1, float* k2, mpn_stream_t stream);
/* mpn_bn_bwd_finalize for a slab whose second row holds sum g * x with the RAW x (mpn_conv_bwd_data_bn): mean / invstd = the
 * layer's saved batch statistics. */
int mpn_bn_bwd_finalize_raw(float* part, int nparts, int C, long long count, float* dgamma, float* dbeta, float* k1,
                            float* k2, const float* mean, const float* invstd, mpn_stream_t stream);
/* dA <- scale*(g - k1 - xhat*k2) in place; add_ch0 (NULL or [M] f32) is added to channel 0 */
int mpn_bn_bwd_apply(void* dA, const void* x, long long M, int C, int dtype, const float* scale,
                     const float* shift, const float* mean, const float* invstd, const float* k1,
                     const float* k2, int act, const float* add_ch0, mpn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K3  depthwise 3x3, TF 'SAME' padding, stride 1|2 (tf.nn.depthwise_conv2d,
 * detector/backbones/mobilenet_v1.py:82-104). w = `depthwise_weights` [3,3,C,1] f32 as is.
 * x [N,H,W,C] -> y [N,OH,OW,C], OH = ceil(H/stride). in_scale/in_shift/in_act as for mpn_conv_fwd.
 * stats_part: NULL or [mpn_dwconv_num_parts()][2][C].
 */
int mpn_dwconv_out_size(int size, int stride);
int mpn_dwconv_num_parts(int N, int H, int W, int C, int stride, int dtype);
int mpn_dwconv_fwd(const void* x, const float* w, void* y, int N, int H, int W, int C, int stride,
                   int dtype, const float* in_scale, const float* in_shift, int in_act, int flip,
                   float* stats_part, mpn_stream_t stream);
/* dy [N,OH,OW,C] -> dx [N,H,W,C]  (H, W: forward input size) */
int mpn_dwconv_bwd_data(const void* dy, const float* w, void* dx, int N, int H, int W, int C,
                        int stride, int dtype, mpn_stream_t stream);
/* The same with the batch-norm backward REDUCTION of the layer that dx feeds fused in (the pointwise conv + batch-norm whose
 * activated output the depthwise conv read, mobilenet_v1.py:88-110): dx is that layer's dA, x_bn its raw conv output
 * [N,H,W,C]; part [mpn_dwconv_bwd_data_bn_num_parts][2][C] receives sum(g) and sum(g*xhat) per block in the layout of
 * mpn_bn_bwd_reduce (finish with mpn_bn_bwd_finalize(part, rows, C, N*H*W, ...), then mpn_bn_bwd_apply) - one tensor read
 * and one launch less than reducing afterwards. num_parts == 0: not available for this shape (odd H / W at stride 2). */
int mpn_dwconv_bwd_data_bn_num_parts(int N, int H, int W, int C, int stride, int dtype);
int mpn_dwconv_bwd_data_bn(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int stride, int dtype,
                           const void* x_bn, const float* scale, const float* shift, const float* mean,
                           const float* invstd, int act, float* part, mpn_stream_t stream);
/* dx = data gradient + addend, addend [N,H,W,C] of dx's type and not aliasing it: the gradient an FPN lateral sends into
 * the same backbone feature map (the sum of the two consumers of c2..c4, mobilenet_v1.py:76-79 / fpn.py:48) - one read
 * instead of a separate read-modify-write pass and a single rounding of the sum. x_bn != NULL: also the batch-norm
 * backward reduction of mpn_dwconv_bwd_data_bn, over the SUM (part: mpn_dwconv_bwd_data_bn_num_parts rows); x_bn == NULL:
 * scale .. part are ignored. Stride-2 layers with even H, W only (mpn_dwconv_bwd_data_add_supported == 1). */
int mpn_dwconv_bwd_data_add_supported(int N, int H, int W, int C, int stride, int dtype);
int mpn_dwconv_bwd_data_add(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int stride, int dtype,
                            const void* addend, const void* x_bn, const float* scale, const float* shift,
                            const float* mean, const float* invstd, int act, float* part, mpn_stream_t stream);
int mpn_dwconv_wgrad_num_parts(int N, int H, int W, int C, int stride, int dtype);
/* part [mpn_dwconv_wgrad_num_parts()][9][C]; finish with mpn_reduce_partials */
int mpn_dwconv_bwd_weight(const void* x, const void* dy, float* part, int N, int H, int W, int C,
                          int stride, int dtype, const float* in_scale, const float* in_shift,
                          int in_act, mpn_stream_t stream);

/* Stride-1 depthwise backward in ONE pass over dy, x and dx - both gradients of tf.nn.depthwise_conv2d
 * (/root/reference/detector/backbones/mobilenet_v1.py:101) and the batch-norm backward reduction of the layer that produced x
 * (mobilenet_v1.py:29-38): dx [N,H,W,C] = data gradient; wpart [mpn_dwconv_wgrad_num_parts][9][C] = weight-gradient partials over
 * act(x * in_scale + in_shift) (finish with mpn_reduce_partials); bn_part (NULL: no reduction) [mpn_dwconv_wgrad_num_parts][2][C] =
 * sum(g), sum(g * xhat) with g = dx where in_act passes, xhat = (x - mean) * invstd (finish with mpn_bn_bwd_finalize). Replaces
 * mpn_dwconv_bwd_weight + mpn_dwconv_bwd_data_bn (five tensor passes) by three. dx must not alias x or dy. */
int mpn_dwconv_bwd_fused_supported(int N, int H, int W, int C, int stride, int dtype);
int mpn_dwconv_bwd_fused(const void* x, const void* dy, const float* w, void* dx, float* wpart, int N, int H, int W, int C,
                         int dtype, const float* in_scale, const float* in_shift, int in_act, const float* mean,
                         const float* invstd, float* bn_part, mpn_stream_t stream);
/* The stride-2 counterpart (even H and W; mpn_dwconv_bwd_fused_supported(.., 2, ..) == 1): dy [N,H/2,W/2,C]; dx [N,H,W,C] = data gradient
 * (+ addend when not NULL: a tensor of dx's shape - the FPN lateral's gradient into the same backbone feature map - added before
 * the store and before the reduction, as mpn_dwconv_bwd_data_add does); wpart [mpn_dwconv_wgrad_num_parts(.., 2, ..)][9][C]; bn_part
 * (NULL: no reduction) [same rows][2][C]. The input and dy are read once instead of twice. dx must not alias x, dy or addend. */
int mpn_dwconv_bwd_fused_s2(const void* x, const void* dy, const float* w, void* dx, float* wpart, int N, int H, int W, int C,
                            int dtype, const float* in_scale, const float* in_shift, int in_act, const float* mean,
                            const float* invstd, float* bn_part, const void* addend, mpn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K1+K2  `2*x-1` + Conv2d_0 (3x3 stride 2 'SAME', 3 -> C0) fused
 * (detector/backbones/mobilenet_v1.py:41,53,56). images NHWC f32 in [0,1], or uint8
 * (images_u8=1: scaled by 1/255 first, create_pb.py:167). w = `Conv2d_0/weights` [3,3,3,C0] f32.
 */
int mpn_stem_conv_fwd(const void* images, int images_u8, const float* w, void* y, int N, int H,
                      int W, int C0, int dtype, mpn_stream_t stream);
/* The same + the batch-norm statistics of the (rounded) output, fused into the kernel's copy-out: stats_part
 * [mpn_stem_conv_fwd_num_parts()][2][C0] receives per-tile sum / sum of squares in the layout of mpn_bn_stats (finish with
 * mpn_bn_finalize; mobilenet_v1.py:56 applies batch-norm to Conv2d_0). num_parts == 0: not available for this C0 - run
 * mpn_bn_stats on the output instead. stats_part == NULL: plain forward. */
int mpn_stem_conv_fwd_num_parts(int N, int H, int W, int C0, int dtype);
int mpn_stem_conv_fwd_stats(const void* images, int images_u8, const float* w, void* y, int N, int H, int W, int C0,
                            int dtype, float* stats_part, mpn_stream_t stream);
int mpn_stem_conv_wgrad_num_parts(int N, int H, int W);
/* part [num_parts][27*C0] */
int mpn_stem_conv_bwd_weight(const void* images, int images_u8, const void* dy, float* part, int N,
                             int H, int W, int C0, int dtype, mpn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K9/K10  legacy bilinear up-sampling by an integer factor into a channel slice of the concat
 * tensor (tf.image.resize_bilinear + tf.concat, detector/keypoint_subnet.py:37,86), its transpose,
 * and the gradient of the FPN's nearest-2x upsample (detector/fpn.py:58-76).
 */
int mpn_bilinear_up_fwd(const void* x, void* y, int N, int h, int w, int C, int upsample,
                        int y_channel_offset, int y_channels_total, int dtype,
                        const float* in_scale, const float* in_shift, int in_act,
                        mpn_stream_t stream);
int mpn_bilinear_up_bwd(const void* dy, void* dx, int N, int h, int w, int C, int upsample,
                        int y_channel_offset, int y_channels_total, int dtype, mpn_stream_t stream);
/* src [N,2h,2w,C] -> dst [N,h,w,C] (2x2 sums); accumulate != 0 adds into dst */
int mpn_sumpool2x2(const void* src, void* dst, int N, int h, int w, int C, int accumulate,
                   int dtype, mpn_stream_t stream);
/* dst += src over n elements: joins the two gradient paths that meet at c2..c4 */
int mpn_add_inplace(void* dst, const void* src, long long n, int dtype, mpn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K11  `heatmaps` head: 1x1 conv Cin -> 18 + bias, f32 NHWC logits
 * (detector/keypoint_subnet.py:49-58); mode 1 = inference post-ops of create_pb.py:73-76.
 * w = `heatmaps/kernel` [1,1,Cin,18] f32, bias [18].
 */
int mpn_heatmap_head_fwd(const void* x, const float* w, const float* bias, long long M, int Cin,
                         int dtype, const float* in_scale, const float* in_shift, int in_act,
                         int mode, float* out, float* out_seg, mpn_stream_t stream);
int mpn_heatmap_head_bwd_num_parts(long long M);
/* dA [M][Cin]; part [num_parts][Cin*18 + 18] (dW then db) */
int mpn_heatmap_head_bwd(const void* x, const float* dlogits, const float* w, long long M, int Cin,
                         int dtype, const float* in_scale, const float* in_shift, int in_act,
                         void* dA, float* part, mpn_stream_t stream);
/* The same with the reduction pass of the batch-norm the head reads through (final_bn, keypoint_subnet.py:42-47) fused in:
 * dA comes out masked by that layer's activation and bn_part [mpn_heatmap_head_bwd_num_parts(M)][2][Cin] receives per-block
 * sums of g and g * x (raw x); finish with mpn_bn_bwd_finalize_raw + mpn_bn_bwd_apply. bf16, Cin = 16 / 32 / 64
 * (mpn_heatmap_head_bwd_bn_supported). */
int mpn_heatmap_head_bwd_bn_supported(int Cin, int dtype);
int mpn_heatmap_head_bwd_bn(const void* x, const float* dlogits, const float* w, long long M, int Cin, int dtype,
                            const float* in_scale, const float* in_shift, int in_act, void* dA, float* part,
                            float* bn_part, mpn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K12  fused losses + gradients (keypoints_model.py:43-90,141-178).
 * losses_out[8] = focal, regression, seg@2, seg@3, seg@4, seg@5, total, per_pixel_reg_loss.
 * dlogits / daux* may be NULL (evaluation). part: [mpn_keypoint_loss_num_parts()][8].
 */
int mpn_keypoint_loss_num_parts(int B, int h, int w);
int mpn_keypoint_loss(const float* logits, const float* heatmaps, const float* loss_masks,
                      const float* segmentation_masks, const int* num_boxes, const void* p2,
                      const void* p3, const void* p4, const void* p5, int p_channels, int p_dtype,
                      float* dlogits, float* daux2, float* daux3, float* daux4, float* daux5,
                      float* part, float* losses_out, int B, int h, int w, mpn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K13  optimizer step (keypoints_model.py:107-120): cosine-decayed learning rate,
 * tf.clip_by_value(g,-200,200), TF-1.15 Adam over a flat f32 arena. `step` (device int64 global
 * step, incremented by mpn_adam_prepare) and `hyper` (device f32[4]: lr_t, lr) stay on the
 * device so the whole step replays from a hipGraph.
 */
int mpn_adam_prepare(long long* step, float* hyper, double initial_learning_rate,
                     double decay_steps, double alpha, double beta1, double beta2,
                     mpn_stream_t stream);
int mpn_adam_step(float* params, const float* grads, float* m, float* v, long long n,
                  const float* hyper, float beta1, float beta2, float eps, float clip,
                  float grad_scale, mpn_stream_t stream);
/* mpn_adam_step that also writes 16-bit copies (fp16 / bf16 by cast_dtype) of up to 4 ranges of the UPDATED arena:
 * cast_dst[r][i] = (T) params[cast_begin[r] + i], i < cast_count[r] - the storage-dtype GEMM operand of a dense layer without a
 * second pass over its f32 master. begin / count multiples of 4, dst 8-byte aligned. */
int mpn_adam_step_cast(float* params, const float* grads, float* m, float* v, long long n, const float* hyper, float beta1,
                       float beta2, float eps, float clip, float grad_scale, int ncast, const long long* cast_begin,
                       const long long* cast_count, void* const* cast_dst, int cast_dtype, mpn_stream_t stream);
/* out[j] (+)= scale * sum_p part[p][j] in a fixed order (deterministic). `part` is scratch: large slabs are
 * reduced in two passes and the first pass folds range sums into the slab itself (its contents are clobbered). */
int mpn_reduce_partials(const float* part, int nparts, long long n, float* out, int accumulate,
                        float scale, mpn_stream_t stream);
/* All slab reductions of a training step in ONE launch (every weight-gradient kernel leaves a slab; 46 per step).
 * mpn_reduce_desc_fill writes one host-side job descriptor (mpn_reduce_desc_bytes() bytes; block_begin = running sum
 * of the returned block counts), the caller copies the table to the device once, mpn_reduce_partials_batched runs
 * out[j] = scale * sum_p part[p][j] for every job (fixed summation order; slabs are left intact). */
size_t mpn_reduce_desc_bytes(void);
int mpn_reduce_desc_fill(void* desc_host, const float* part, int nparts, long long n, float* out, float scale,
                         int block_begin);
int mpn_reduce_partials_batched(const void* descs_device, int ndesc, int total_blocks, mpn_stream_t stream);
int mpn_axpy(long long n, float a, const float* x, float* y, mpn_stream_t stream);
/* y[t][i] += a * x[t][i] over `count` tensors (host arrays of device pointers / element counts), one launch per 64 tensors:
 * the gradient of add_weight_decay's term for every regularised variable at once (keypoints_model.py:129-138). */
int mpn_axpy_batched(int count, const float* const* x, float* const* y, const long long* n, float a, mpn_stream_t stream);
/* acc[0] += scale * sum(w^2)/2 - `weight_decay * tf.nn.l2_loss(k)` of add_weight_decay (keypoints_model.py:129-138), the
 * term tf.losses.get_total_loss(add_regularization_losses=True) adds to the reported loss (keypoints_model.py:79).
 * One block, fixed summation order (f64): deterministic. */
int mpn_l2_loss_accumulate(long long n, const float* w, float scale, float* acc, mpn_stream_t stream);
/* The same over `count` tensors (host arrays of device pointers / element counts) in two launches: per-block f64 partial sums
 * into `workspace` (mpn_l2_loss_batched_workspace_bytes bytes), then one block adds them in a fixed order - the whole
 * regularisation term of a model (keypoints_model.py:129-138 sums every kernel) without one single-block launch per variable. */
size_t mpn_l2_loss_batched_workspace_bytes(int count, const long long* n);
int mpn_l2_loss_batched(int count, const float* const* w, const long long* n, float scale, float* acc, void* workspace,
                        size_t workspace_bytes, mpn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * L1  target-heatmap rendering (label producer of the keypoint path).
 * Replaces detector/input_pipeline/heatmap_creation.py:6-72 `get_heatmaps` (+ `get_kernel`
 * :75-86, `create_heatmap` :89-118), which the reference runs per image under the GIL via
 * tf.py_func (keypoints_detector_pipeline.py:86-90, prn_pipeline.py:91-95), for a whole
 * batch of equally sized images in one launch:
 *
 *   out[b,y,x,j] = max(0, max over visible persons p of image b of
 *                         float32(g_p[|y-cy_pj|] * g_p[|x-cx_pj|]),  |dy|,|dx| <= k_p)
 *   sigma_p = clip(0.007f * sqrtf(box area), 1, 4)            (float32, :33-37)
 *   k_p     = ceil(sqrt(-2 sigma^2 ln 0.01)), g_p[d] = exp(-d^2 / (2 sigma^2))  (float64, :78-84)
 *   cy, cx  = rint(float32(y / (height-1)) * (h-1)), likewise x  (:23-24,57,104-107)
 * Bit-identical to the reference (tests/golden/render_goldens.npz); peaks are exactly 1.0f.
 *
 *   keypoints    [P,17,3] int32 (y, x, visibility), persons of all images concatenated
 *   boxes        [P,4] f32 (ymin, xmin, ymax, xmax), absolute
 *   first_person [B+1] int32, device: persons of image b are first_person[b] .. first_person[b+1]-1
 *   width,height size of the (equally sized) images; h = ceil(height/downsample), w likewise
 *   out          [B,h,w,17] f32, every element written
 *   workspace    mpn_heatmap_render_workspace_bytes(P) bytes of scratch (no initialisation needed)
 * A blob whose centre lies outside the map is clipped (the numpy code raises or wraps there).
 */
size_t mpn_heatmap_render_workspace_bytes(int total_persons);
int mpn_heatmap_render(const int32_t* keypoints, const float* boxes, const int32_t* first_person,
                       int B, int total_persons, int width, int height, int downsample, float* out,
                       void* workspace, size_t workspace_bytes, mpn_stream_t stream);

/* Skinny "NT" GEMM split over K: part[s][M][N] (f32, s < mpn_gemm_nt_num_parts(K)) = A[M][K] * B[N][K]^T over the s-th K range.
 * A, B 16-bit (MPN_BF16 / MPN_F16), row-major with K contiguous - the order the reference's variables already have for
 * dH = dPre2 * W2^T in the pose residual network (prn.py:11-33: fc2's data gradient), so no transposed copy of the weights is
 * made. K % 8 == 0, N % 4 == 0; finish with mpn_reduce_partials(part, parts, M * N, out, ...). */
int mpn_gemm_nt_num_parts(int K);
int mpn_gemm_nt(const void* a, const void* b, float* part, int M, int N, int K, int dtype, mpn_stream_t stream);
/* ------------------------------------------------------------------------------------
 * L2  PRN - pose residual network (BASELINE config 5).
 * Replaces detector/prn.py:5-25 (flatten -> fc1 34272->1024 + ReLU -> fc2 1024->34272 + ReLU -> x + y) and the loss of
 * prn_model.py:16-30 (softmax over h*w per keypoint channel, tf.losses.log_loss eps 1e-7, mean). The four GEMMs of a
 * training step run on mpn_conv_fwd / mpn_conv_bwd_weight (K = 34272 contractions as split-K "weight gradients" whose
 * pixel axis is K); these entry points are the glue: K-major operand copies, bias + ReLU, the loss and its gradient.
 */
int mpn_transpose_cast(const void* in, int in_dtype, void* out, int out_dtype, int R, int C, mpn_stream_t stream);
int mpn_cast(const void* in, int in_dtype, void* out, int out_dtype, long long n, mpn_stream_t stream);
int mpn_bias_relu_fwd(const void* pre, int pre_dtype, const float* bias, void* y, int out_dtype, int R, int C,
                      mpn_stream_t stream);
int mpn_bias_relu_bwd(const void* y, int y_dtype, const float* dy, void* dpre, int dpre_dtype, float* dbias, int R,
                      int C, mpn_stream_t stream);
/* grad_scale: dlogits = grad_scale * dloss/dlogits (static loss scale of the fp16 build, 1 otherwise; the caller passes
 * 1 / grad_scale to mpn_adam_step) */
int mpn_prn_loss(const float* x, const void* y2, int y2_dtype, const float* labels, int B, int P, int C,
                 float* logits, float* dlogits, float* loss_part, float grad_scale, mpn_stream_t stream);
/* logits[i] = x[i] + y2[i]: the residual connection of detector/prn.py:24 at inference (create_pb.py:112) */
int mpn_prn_residual(const float* x, const void* y2, int y2_dtype, long long n, float* logits, mpn_stream_t stream);

/* PRN inference glue (create_pb.py:86-142): what sits between the sigmoid heatmaps / the detector's boxes and the
 * network, and between its logits and the exported `keypoint_scores` / `keypoint_positions`.
 *   mpn_heatmap_minmax  per (image, channel) min and max over the map (create_pb.py:90-92: M, m). minmax_keys: B*C*2
 *                       32-bit words (opaque order-preserving keys, decoded by mpn_prn_crop). heatmaps f32 [B,h,w,C], C <= 17.
 *   mpn_prn_crop        (heatmaps - m) / (M - m) * (M > threshold), then tf.image.crop_and_resize (bilinear, extrapolation
 *                       value 0; create_pb.py:93-109) of box n of image box_ind[n]: boxes f32 [nb,4] normalised
 *                       (y1,x1,y2,x2) -> crops f32 [nb,crop_h,crop_w,C]. box_ind outside [0,B) gives a zero crop.
 *   mpn_prn_decode      logits f32 [nb,crop_h,crop_w,C] -> softmax over the crop_h*crop_w positions of each channel
 *                       (create_pb.py:114-117): scores f32 [nb,C] = its maximum, positions f32 [nb,C,2] =
 *                       (y / crop_h, x / crop_w) of the first maximum (argmax_2d + scaler, create_pb.py:119-138).
 */
int mpn_heatmap_minmax(const float* heatmaps, int B, int h, int w, int C, void* minmax_keys, mpn_stream_t stream);
int mpn_prn_crop(const float* heatmaps, const void* minmax_keys, const float* boxes, const int* box_ind, int nb, int B,
                 int h, int w, int C, int crop_h, int crop_w, float threshold, float* crops, mpn_stream_t stream);
/* mpn_prn_crop for `nb` consecutive slots slot0.. of a detector's padded output (boxes f32 [B,max_boxes,4], num_boxes i32
 * [B], retinanet.py:60-84): slot s = box s % max_boxes of image s / max_boxes; padding slots and slots past the array give
 * zero crops - the [:n] slices, box_ind vectors and concat of create_pb.py:96-104 without materialising them. */
int mpn_prn_crop_slots(const float* heatmaps, const void* minmax_keys, const float* boxes, const int* num_boxes, int slot0,
                       int nb, int max_boxes, int B, int h, int w, int C, int crop_h, int crop_w, float threshold,
                       float* crops, mpn_stream_t stream);
int mpn_prn_decode(const float* logits, int nb, int crop_h, int crop_w, int C, float* scores, float* positions,
                   mpn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * L3  RetinaNet person-detector head (BASELINE config 4): detector/retinanet.py:13-217,
 * detector/box_predictor.py:6-142, detector/fpn.py:42-46, detector/training_target_creation.py:5-159,
 * detector/utils/box_utils.py:14-139, detector/utils/nms.py:6-61. Its convolutions / batch-norms / optimizer are the entry
 * points above; these are the detector-specific pieces. Level arrays have 5 entries (p3..p7); anchors are ordered like
 * reshape_and_concatenate (box_predictor.py:55-90): level, then y, x, then the 6 anchors of a location.
 *
 *   mpn_patchify3x3s2     x [N,H,W,C] -> patches [N,ceil(H/2),ceil(W/2),9*C] for conv2d_same(k=3, stride=2) (pad 1, VALID):
 *                         the 3x3 stride-2 convolutions p6 / p7 (fpn.py:43,45) = this gather (producer's affine + act applied
 *                         on the way, zero padding after it) + a 1x1 convolution over 9*C channels with the HWIO kernel viewed
 *                         as [1,1,9*C,Cout]. mpn_unpatchify3x3s2 is its transpose: dpatches -> dx [N,H,W,C].
 *   mpn_retina_match      get_training_targets for a batch: anchors f32 [A,4] (normalised), gt_boxes f32 [B,max_boxes,4],
 *                         num_boxes int32 [B] -> matches int32 [B,A] (-2 ignore, -1 background, else the box index), targets
 *                         f32 [B,A,4] (encode(), zeros where unmatched), num_matched int32[1] = count of matches >= 0.
 *                         float32 arithmetic in the reference's operation order; arg-max ties: first index (tf.argmax).
 *   mpn_retina_loss       focal loss (gamma, alpha; weights = not ignored) + smooth L1 (matched anchors) over raw tower outputs
 *                         logits[l] [B,h,w,8] (6 used) / boxes[l] [B,h,w,24] + biases, both normalised by max(num_matched, 1);
 *                         dlogits / dboxes (same layouts, may be NULL arrays) receive d(loc_w*loc + cls_w*cls); part
 *                         [mpn_retina_loss_num_parts][32] = partial sums of {cls loss, loc loss, dbias_cls[6], dbias_box[24]}
 *                         (finish with mpn_reduce_partials; the two losses still need the 1/normaliser).
 *   mpn_retina_nms        get_predictions: sigmoid, score >= threshold, decode + clip to [0,1], greedy NMS, zero padding:
 *                         out_boxes f32 [B,max_det,4], out_scores f32 [B,max_det], out_num int32 [B].
 *                         The workspace holds one candidate list of A slots per image, a counter per image and ONE int32
 *                         OVERFLOW word at byte mpn_retina_nms_overflow_offset(B, A): 0 after a good call, 1 when a list would
 *                         have grown past its A slots (the appends past the list are refused, the selection never reads past
 *                         it). A caller reads it WITH the outputs (the library never synchronises; from a captured graph it is
 *                         the only report there is) and treats 1 as MPN_ERR_WORKSPACE.
 */
int mpn_patchify3x3s2(const void* x, void* patches, int N, int H, int W, int C, int dtype, const float* in_scale,
                      const float* in_shift, int in_act, mpn_stream_t stream);
int mpn_unpatchify3x3s2(const void* dpatches, void* dx, int N, int H, int W, int C, int dtype, mpn_stream_t stream);
size_t mpn_retina_match_workspace_bytes(int B, int max_boxes);
int mpn_retina_match(const float* anchors, const float* gt_boxes, const int* num_boxes, int B, int A, int max_boxes,
                     float positives_threshold, float negatives_threshold, int* matches, float* targets,
                     int* num_matched, void* workspace, size_t workspace_bytes, mpn_stream_t stream);
int mpn_retina_loss_num_parts(int B, int A);
int mpn_retina_loss(const void* const* logits, const void* const* boxes, void* const* dlogits, void* const* dboxes,
                    const int* h, const int* w, int dtype, const float* cls_bias, const float* box_bias,
                    const int* matches, const float* targets, const int* num_matched, int B, float gamma, float alpha,
                    float localization_loss_weight, float classification_loss_weight, float* part,
                    mpn_stream_t stream);
/* The scalar losses of a step from the reduced partial sums (retinanet.py:128-144, person_detector_model.py:33-45):
 * sums f32 [32] = {cls, loc, dbias_cls[6], dbias_box[24]} (mpn_reduce_partials of mpn_retina_loss's parts);
 * losses f32 [4] = {localization = sums[1] / max(num_matched, 1), classification = sums[0] / max(num_matched, 1),
 * regularization (left as the caller accumulated it), total = loc_w * localization + cls_w * classification + regularization};
 * dbias_cls f32 [6] / dbias_box f32 [24] (may be NULL) receive the output convolutions' bias gradients. */
int mpn_retina_loss_finalize(const float* sums, const int* num_matched, float localization_loss_weight,
                             float classification_loss_weight, float* losses, float* dbias_cls, float* dbias_box,
                             mpn_stream_t stream);
size_t mpn_retina_nms_workspace_bytes(int B, int A);
size_t mpn_retina_nms_overflow_offset(int B, int A);
int mpn_retina_nms(const void* const* logits, const void* const* boxes, const int* h, const int* w, int dtype,
                   const float* cls_bias, const float* box_bias, const float* anchors, int B, float score_threshold,
                   float iou_threshold, int max_detections, float* out_boxes, float* out_scores, int* out_num,
                   void* workspace, size_t workspace_bytes, mpn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MPN_H_ */
