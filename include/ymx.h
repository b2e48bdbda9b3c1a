/* ymx — C ABI of the MI355X (gfx950) detector hot path.
 *
 * The reference (Jex-y/yolo, TensorFlow/Keras) has no native or FFI layer: its
 * hot path is reached through Python modules only.  Each entry point below
 * therefore cites the Python symbol of the reference whose device work it
 * carries; the Python package in yolo_amd/ keeps those symbols' names and
 * signatures and calls these functions through ctypes.
 *
 * Conventions
 *  - every pointer named x/w/y/boxes/scores/... is DEVICE memory owned by the
 *    caller; the library never allocates, frees or synchronises, never changes
 *    the current device, and launches on the stream it is given (a hipStream_t
 *    passed as void*; NULL = the default stream);
 *  - arguments documented as "host" are small host arrays read during the call;
 *  - activations are NHWC, contiguous unless a pixel stride (ld, in elements)
 *    is given; element types are named by YMX_F32 / YMX_BF16;
 *  - return value: YMX_OK or a negative YMX_E* code; ymx_last_error() returns a
 *    thread-local message for the last failing call on this thread.
 */
#ifndef YMX_H
#define YMX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YMX_VERSION 100

enum { YMX_F32 = 0, YMX_BF16 = 1, YMX_U8 = 2 /* image entry points only */ };
enum { YMX_ACT_NONE = 0, YMX_ACT_LEAKY = 1, YMX_ACT_MISH = 2 };
enum {
    YMX_OK = 0,
    YMX_EARG = -1,        /* bad argument (null pointer, non-positive size, misalignment) */
    YMX_EUNSUPPORTED = -2,/* shape / dtype combination no kernel covers */
    YMX_EWORKSPACE = -3,  /* workspace too small */
    YMX_ELAUNCH = -4      /* HIP reported a launch error */
};

int ymx_version(void);
const char *ymx_last_error(void);

/* Tuning knobs (no counterpart in the reference).  They are PER CALLING THREAD: a thread
 * that sets one changes only its own later calls, every thread starts from the defaults, and
 * the library keeps no other mutable state besides the thread-local error string (kernel
 * attributes it sets once per device are idempotent).  Mind frameworks that launch from threads
 * of their own: torch.autograd runs backward() on a worker thread, so an option set on the
 * thread that calls backward() does not reach the launches made inside it (the Python side
 * sets yolo_amd.training.options instead, which are applied on the launching thread).  "dma_min_tiles": the fewest 256x256
 * output tiles for which a bf16 convolution with more than 128 output channels runs on the
 * large-tile LDS-DMA kernel instead of the 128x128 kernel (default 160; 0 forces it, a huge
 * value disables it; ymx_op.min_tiles overrides it per op).  "dma_variant": which large-tile
 * kernel wide layers use (5, the default: input pixels resident in LDS for 3x3 stride 1,
 * phase-staggered wave groups otherwise; 1: two-stage 256x256 everywhere; 2: 256x128 tile, two workgroups per CU; 4:
 * phase-staggered wave groups everywhere).  "dma_stages": pipeline depth (3 or 4, default
 * 4) of the 32- and 64-channel narrow-layer tiles.  "dma_narrow": bf16 layers of <= 128
 * output channels with at least dma_min_tiles 256-pixel tiles run on the LDS-DMA kernels
 * (2, the default: with the input-resident kernel for 3x3 stride 1 and 32 or 64 input
 * channels; 1: without it) or on the register-staged kernel (0).  "halo_pixels": pixels per
 * tile of the resident-slab 3x3 kernel (0, the default, chooses per layer among 256, 192 and
 * both in one launch -- whole rounds of 256-pixel tiles on the chip's CUs, the remaining pixels
 * as 192-pixel tiles; 192 or 256 force one, 448 the mixed form where the launch has two rounds
 * or more; rows longer than 59 pixels always take 192).  "dma4_pixels": the same choice for the
 * phase-staggered kernel.  "tile_mix": 0 restricts the per-layer choice to the two pure forms
 * (default 1).  "halo_big_rounds": rounds of 256-pixel tiles of a mixed launch (0, the default:
 * priced per layer).
 * "wgrad_variant": 0 (the default) lets ymx_conv_wgrad choose: the all-taps LDS-DMA kernel for
 * 3x3 stride 1 on rows up to 459 pixels, the all-taps planar-staged kernel for 3x3 stride 2
 * with up to 32 input channels, else the per-tap LDS-DMA kernel, both for x and dz
 * under 2 GB, else the register-staged kernel; 1 forces the register-staged kernel, 2 the
 * per-tap LDS-DMA kernel (where the tensors allow it).  All forward kernels add the K terms in
 * the same order: forward results do not depend on these options (the weight-gradient kernels
 * slice the pixel sum differently and agree to fp32 rounding). */
int ymx_set_option(const char *name, long long value);

/* ---- convolution block ------------------------------------------------------
 * Replaces YOLOConv.call — layers.py:86-90 with the stack built at layers.py:46-81
 * ([ZeroPadding2D top/left] -> Conv2D -> [BatchNormalization] -> [Mish|LeakyReLU])
 * and the shortcut add of ResidialBlock.call, layers.py:100-106.
 *
 *   y = act( conv(x, w) * scale + shift ) + residual
 *
 * stride 1: SAME padding ((k-1)/2 each side), layers.py:55-57.
 * stride 2: one zero row/column on the top/left only, then VALID, layers.py:46-54.
 * scale == NULL means 1 (plain bias in shift: use_bias = not batchNormalization,
 * layers.py:61); with batch norm the caller folds it: scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale (layers.py:4-15 forces inference statistics).
 * residual == NULL means no shortcut.  act: YMX_ACT_*; leaky slope is 0.1
 * (layers.py:76).
 *
 * w is the PACKED kernel produced by ymx_pack_weights from the reference's HWIO
 * fp32 layout: an opaque buffer of ymx_packed_weight_bytes bytes, valid for this
 * library version only (rows = output channels padded to 128, columns = K padded to
 * 64, K ordered in blocks of 64 input channels, then filter taps, then the channels
 * of the block; Cin not a multiple of 8 keeps HWIO fp32).  dtype is the storage type
 * of x, residual and the MFMA operands (accumulation is always fp32).  scale/shift
 * are fp32 [Cout].  All kernels the library may choose for a shape add the K terms
 * in the same order: results do not depend on the tuning options or on the batch a
 * pixel is computed in.
 */
size_t ymx_packed_weight_bytes(int k, int Cin, int Cout, int dtype);
int ymx_pack_weights(const float *w_hwio, void *w_packed, int k, int Cin, int Cout, int dtype,
                     void *stream);

int ymx_conv2d_fused(const void *x, const void *w, const void *scale, const void *shift,
                     const void *residual, void *y, int B, int H, int W, int Cin, int Cout,
                     int k, int stride, int act, int dtype, void *stream);

/* Same, with explicit pixel strides (elements; 0 = dense), a separate input
 * element type for the direct (Cin not a multiple of 8) path, and an output
 * element type (the head's three linear output convs, detectors.py:20,33,43, are
 * written as fp32 for the decode that follows). */
int ymx_conv2d_fused_ex(const void *x, const void *w, const void *scale, const void *shift,
                        const void *residual, void *y, int B, int H, int W, int Cin, int Cout,
                        int k, int stride, int act, int dtype, int x_dtype, int out_dtype,
                        int x_ld, int res_ld, int y_ld, void *stream);

/* ---- upsample + concat ------------------------------------------------------
 * Replaces Upsample.call (layers.py:37-39: tf.image.resize bilinear x2, half-pixel
 * centres, edge clamp) fused with the tf.concat that follows it at
 * detectors.py:57-59 and 70-72:  y[..., :C1] = skip,  y[..., C1:] = up2(x).
 * skip [B,2H,2W,C1] (NULL with C1 = 0 for a bare upsample), x [B,H,W,C2],
 * y [B,2H,2W,C1+C2].  skip = NULL with C1 > 0: the first C1 channels of y are already in
 * place (the convolution that produces them wrote them there through its y_ld: a concat
 * without the copy) and only y[..., C1:] is written. */
int ymx_upsample2x_concat(const void *skip, const void *x, void *y, int B, int H, int W,
                          int C1, int C2, int dtype, void *stream);

/* ---- the backbone's first two layers in one launch ---------------------------
 * Replaces the pair YOLOConv((3,3,32)), YOLOConv((3,3,64), downsample=True) that opens
 * Darknet53 and CSPDarknet53 (backbones.py:11-12, 64-66; YOLOConv.call layers.py:77-90)
 * when nothing else reads the 32-channel map: it is kept in LDS tile by tile instead of being
 * written to memory and read back.  Results are bit-identical to two ymx_conv2d_fused_ex calls.
 *   x [B,H,W,3] fp32 (dense);  w0: the first layer's weights as ymx_pack_weights gives them for
 *   Cin = 3 (HWIO fp32);  w1: the second layer's bf16 packed weights (k = 3, Cin = 32, Cout = 64);
 *   scale*, shift*: per layer as in ymx_conv2d_fused (NULL = 1 / 0);  act: both layers' activation,
 *   YMX_ACT_MISH or YMX_ACT_LEAKY;  y [B, OH, OW, 64] bf16 with OH = (H - 2) / 2 + 1 (padding
 *   top/left as for every stride-2 layer), pixel stride y_ld elements (0 = 64, else a multiple of 8). */
int ymx_conv_stem_fused(const void *x, const void *w0, const void *scale0, const void *shift0,
                        const void *w1, const void *scale1, const void *shift1, void *y,
                        int B, int H, int W, int act, int y_ld, void *stream);

/* ---- a narrow ResidialBlock in one launch -------------------------------------
 * Replaces ResidialBlock(C/2, C) on a C-channel map, C = 64 or 128 (layers.py:92-106:
 * x + conv3x3(conv1x1(x)), both YOLOConv with bias or folded batch norm and the same activation;
 * the first three blocks of Darknet53, backbones.py:13-16) when nothing else reads the
 * C/2-channel map: it is kept in LDS tile by tile, and x is read once for both the 1x1
 * convolution and the shortcut.  Results are bit-identical to the two ymx_conv2d_fused_ex calls
 * of the block.
 *   x [B,H,W,C] bf16, pixel stride x_ld elements (0 = C);  w1 / w2: the bf16 packed weights of the
 *   1x1 (Cin = C, Cout = C/2) and the 3x3 (Cin = C/2, Cout = C) convolution;  scale*, shift* per
 *   layer as in ymx_conv2d_fused (NULL = 1 / 0);  act: YMX_ACT_MISH or YMX_ACT_LEAKY;
 *   y [B,H,W,C] bf16, pixel stride y_ld elements (0 = C), not x;  the strides are multiples of 8. */
int ymx_conv_resblock_fused(const void *x, const void *w1, const void *scale1, const void *shift1,
                            const void *w2, const void *scale2, const void *shift2, void *y,
                            int B, int H, int W, int C, int act, int x_ld, int y_ld, void *stream);

/* ---- whole-network executor -------------------------------------------------
 * Replaces the Python layer loops Darknet53.call (backbones.py:44-60) and
 * YoloV3.call (detectors.py:46-77): a pre-planned list of the two ops above,
 * launched back to back on one stream by a single call. */
typedef struct ymx_op {
    int32_t kind;                 /* 1 = conv2d_fused_ex, 2 = upsample2x_concat */
    int32_t B, H, W, Cin, Cout, k, stride, act;
    int32_t dtype, x_dtype, out_dtype;
    int32_t x_ld, res_ld, y_ld;
    int32_t C1, C2;               /* upsample2x_concat: channels of skip and x.  conv: C1 != 0 says that
                                     the output is read by the next op alone; ymx_run_program then
                                     runs the pair as ymx_conv_stem_fused or ymx_conv_resblock_fused
                                     when it has that shape and does not write this op's output */
    int32_t min_tiles;            /* conv only: fewest 256-wide tiles for which the large-tile LDS-DMA
                                     kernels are chosen; 0 = the library's "dma_min_tiles" option.
                                     Programs meant to share the GPU with another stream's program
                                     set it lower (fewer, larger tiles: less weight traffic). */
    const void *x, *w, *scale, *shift, *residual;   /* residual doubles as skip */
    void *y;
} ymx_op;
int ymx_run_program(const ymx_op *ops /* host */, int n_ops, void *stream);

/* Diagnostic (no counterpart in the reference): the kernel family ymx_run_program would
 * launch for a convolution op with the calling thread's options -- "halo", "halo192", "dma", "dma4",
 * "dma2", "slab2", "igemm", "first_rows" or "direct" (a static string) -- or NULL with
 * ymx_last_error() set when the op is invalid.  Nothing is launched.  Benchmarks use it to
 * attribute measured launch times to kernels without mirroring the selection rules. */
const char *ymx_conv_kernel_label(const ymx_op *op /* host */);

/* ---- decode (+ score filter) ------------------------------------------------
 * Replaces decode / decode_predict (utils.py:107-126, 133-140) for all scales at
 * once, and with filter != 0 also filter_bboxes (utils.py:142-167):
 *   xy = (sigmoid(t_xy)*xyscale - 0.5*(xyscale-1) + grid) * stride
 *   wh = exp(t_wh) * anchor            (anchors in pixels)
 *   score[c] = sigmoid(t_conf) * sigmoid(t_c)
 * Candidates are ordered scale by scale, and (row, col, anchor) inside a scale.
 *
 * filter == 0: boxes[b][n] = (cx,cy,w,h) pixels for every candidate, counts[b] = N.
 * filter != 0: only candidates with max_c score >= thr survive, in their original
 *   order (order-preserving compaction), boxes = (ymin,xmin,ymax,xmax)/image_size,
 *   src_idx[b][m] = index of survivor m in the unfiltered list, counts[b] = number
 *   of survivors (survivors beyond cap are counted but not written).
 *
 * head[s]: [B,S[s],S[s],3,5+C] (head_dtype);  boxes [B,cap,4] fp32;
 * scores fp32 CLASS-MAJOR [B,C,cap] (element (b,m,c) at (b*C+c)*cap+m; cap % 4 == 0);
 * src_idx [B,cap] (may be NULL when filter == 0); counts [B].
 * anchors host [n_scales][3][2], strides/xyscale host [n_scales].
 * Status word: the first int32 of the workspace is zeroed by the call and left nonzero by
 * the kernel if a workgroup gave up waiting (about a second) for the survivor count of the
 * segments before it, which a healthy device never does; counts and the compacted outputs
 * of that call are then invalid.  Callers that synchronise anyway may check it. */
size_t ymx_decode_workspace_bytes(int B, int n_scales, const int *S /* host */);
int ymx_decode_filter(const void *const *head /* host array of device ptrs */, int n_scales,
                      const int *S /* host */, int B, int C, int head_dtype,
                      const float *anchors, const float *strides, const float *xyscale,
                      int filter, float thr, float image_size, float *boxes, float *scores,
                      int *src_idx, int *counts, int cap, void *workspace, size_t ws_bytes,
                      void *stream);

/* ymx_decode_filter on conv maps whose grid cells are padded: cell (row, col) of head s
 * starts cell_ld[s] elements after the previous one (cell_ld[s] >= 3*(5+C), host array);
 * the 3*(5+C) values of a cell are dense, the rest of the stride is ignored.  The same
 * reference functions (utils.py:107-131, 142-167); lets the producing convolution write
 * 16-byte aligned rows (e.g. 256 for the 255 channels of 80 classes). */
int ymx_decode_filter_strided(const void *const *head, const int *cell_ld /* host */, int n_scales,
                              const int *S /* host */, int B, int C, int head_dtype,
                              const float *anchors, const float *strides, const float *xyscale,
                              int filter, float thr, float image_size, float *boxes, float *scores,
                              int *src_idx, int *counts, int cap, void *workspace, size_t ws_bytes,
                              void *stream);

/* decode (utils.py:107-126) with objectness and class probability kept apart, as
 * decode and decode_train (utils.py:128-131) return them: boxes [B,cap,4] xywh
 * pixels, conf [B,cap] = sigmoid(t_conf), prob class-major [B,C,cap] = sigmoid(t_c). */
int ymx_decode_split(const void *const *head, int n_scales, const int *S, int B, int C,
                     int head_dtype, const float *anchors, const float *strides,
                     const float *xyscale, float *boxes, float *conf, float *prob, int *counts,
                     int cap, void *stream);

/* filter_bboxes (utils.py:142-167) on already decoded tensors: xywh [B,N,4] fp32
 * dense; scores addressed as scores[b*sb + n*sn + c*sc] (element strides), so both
 * candidate-major and class-major inputs work.  Outputs as ymx_decode_filter with
 * filter != 0. */
size_t ymx_filter_workspace_bytes(int B, int N);
int ymx_filter_boxes(const float *xywh, const float *scores, long long sb, long long sn,
                     long long sc, int B, int N, int C, float thr, float image_size,
                     float *boxes, float *out_scores, int *src_idx, int *counts, int cap,
                     void *workspace, size_t ws_bytes, void *stream);

/* ---- combined NMS -----------------------------------------------------------
 * Replaces tf.image.combined_non_max_suppression as called by bbox_nms,
 * utils.py:184-192 (boxes shared by all classes, clip_boxes, no per-class padding).
 * Semantics (TensorFlow's kernel is not part of the reference tree, so they are
 * fixed here): per (image, class) candidates are those with score > score_thr,
 * visited by descending score, equal scores by ascending candidate index; a
 * candidate is kept iff its IoU with every box already kept for the class is
 * <= iou_thr; at most per_class are kept.  Per image the kept entries of all
 * classes are ordered by descending score, then ascending class, then ascending
 * index, and the first `total` are emitted with coordinates clipped to [0,1].
 *
 * boxes [B,cap,4] (ymin,xmin,ymax,xmax), scores class-major [B,C,cap] fp32,
 * counts [B] (NULL = cap candidates each), src_idx [B,cap] or NULL.
 * out_boxes [B,total,4], out_scores [B,total], out_classes [B,total] (fp32),
 * out_valid [B], out_keep_idx [B,total] (candidate index, mapped through src_idx
 * when given; -1 padding).  Limits: per_class <= 64, C*per_class <= 8192,
 * C <= 2048, cap <= 2097152, cap % 4 == 0. */
size_t ymx_nms_workspace_bytes(int B, int C, int per_class);
int ymx_nms_combined(const float *boxes, const float *scores, const int *counts,
                     const int *src_idx, int B, int cap, int C, int per_class, int total,
                     float iou_thr, float score_thr, float *out_boxes, float *out_scores,
                     float *out_classes, int *out_valid, int *out_keep_idx, void *workspace,
                     size_t ws_bytes, void *stream);

/* ymx_nms_combined for callers that cannot allocate or bring a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  Same arguments, semantics and results as
 * ymx_nms_combined, minus the workspace: the per-class survivor lists live in a buffer of
 * YMX_NMS_STATIC_SCRATCH_BYTES that is built into the library (one copy per GPU, held
 * until the process ends).  The call allocates and frees nothing.  Sizes whose
 * ymx_nms_workspace_bytes(B, C, per_class) exceeds the buffer are rejected with
 * YMX_EUNSUPPORTED.  All calls on one GPU share that GPU's buffer: calls that may overlap
 * in time on one GPU (different streams, different threads) race and must be serialised
 * by the caller; calls on one stream are safe back to back. */
#define YMX_NMS_STATIC_SCRATCH_BYTES (4 << 20)
int ymx_nms_combined_static(const float *boxes, const float *scores, const int *counts,
                            const int *src_idx, int B, int cap, int C, int per_class, int total,
                            float iou_thr, float score_thr, float *out_boxes, float *out_scores,
                            float *out_classes, int *out_valid, int *out_keep_idx, void *stream);

/* ---- stand-alone per-channel affine + activation -----------------------------
 * Replaces Mish.call (layers.py:21-22: x*tanh(softplus(x))) and the inference form
 * of BatchNormalization.call (layers.py:11-15) when those layers are used on their
 * own: y = act(x*scale[c] + shift[c]) on [n_pix, C]; scale/shift may be NULL. */
int ymx_affine_act(const void *x, const float *scale, const float *shift, void *y,
                   long long n_pix, int C, int act, int dtype, void *stream);

/* ---- SPP block ---------------------------------------------------------------
 * Replaces the three tf.nn.max_pool(ksize = 13 / 9 / 5, strides = 1, padding = 'SAME') and
 * the tf.concat around them in CSPDarknet53.call (backbones.py:254-259):
 *   y[..., 0:C] = maxpool13(x), y[..., C:2C] = maxpool9(x), y[..., 2C:3C] = maxpool5(x),
 *   y[..., 3C:4C] = x          (padding never wins a max, as in TensorFlow).
 * x [B,H,W,C], y [B,H,W,4C], dense; H*W <= 2048. */
int ymx_spp_concat(const void *x, void *y, int B, int H, int W, int C, int dtype, void *stream);

/* ---- YOLO loss of one scale, forward -----------------------------------------
 * Replaces yolo_loss (losses.py:10-46) as the closures of get_yolo_loss_at_scales call it
 * (losses.py:52-66): GIoU loss * (2 - w*h/input_size^2) on responsible cells, focal-weighted
 * objectness cross-entropy on responsible cells and on background cells whose best IoU
 * with any ground-truth box is below iou_loss_threshold, class cross-entropy on responsible
 * cells; each summed per image and averaged over the batch.
 *   conv   [B,S,S,3,5+C] fp32 raw head map (cell stride conv_ld elements, 0 = dense)
 *   pred   NULL: the decode of conv (decode_train, utils.py:107-131, with stride, xyscale and
 *          the scale's three anchors [3][2], host) is computed inside; else [B,S,S,3,5+C] fp32
 *          dense, of which (cx,cy,w,h,conf) are read, as yolo_loss takes it
 *   label  [B,S,S,3,5+C] fp32 (xywh, respond, class probabilities), dense within an image;
 *          label_stride elements from one image to the next (0 = dense)
 *   bboxes [B,n_boxes,4] fp32 (cx,cy,w,h), unused slots zero (128 in the reference);
 *          bbox_stride elements from one image to the next (0 = dense).  With the two strides
 *          both can be read in place from the reference's packed target rows
 *          [B, S*S*3*(5+C) + 128*4] (losses.py:58-62): no alignment is required of either
 *   out    [3] = (giou_loss, conf_loss, prob_loss); per_image [B][3] per-image sums or NULL
 * input_size = S * stride (losses.py:56).  Deterministic (no float atomics). */
size_t ymx_yolo_loss_workspace_bytes(int B, int S);
int ymx_yolo_loss(const float *conv, int conv_ld, const float *pred, const float *label,
                  long long label_stride, const float *bboxes, long long bbox_stride, int B, int S,
                  int C, int n_boxes, float stride, float xyscale,
                  const float *anchors /* host */, float iou_loss_threshold, float *out,
                  float *per_image, void *workspace, size_t ws_bytes, void *stream);

/* ---- YOLO loss of one scale, gradient ------------------------------------------
 * d(w0 * giou_loss + w1 * conf_loss + w2 * prob_loss) / d conv for the loss ymx_yolo_loss
 * computes with pred = NULL (decode inside), i.e. what back-propagation hands to the head's
 * output convolution in a training step (losses.py:10-66 differentiated; the background
 * mask of losses.py:30 is a comparison and carries no gradient).  Same arguments as
 * ymx_yolo_loss; term_weights device [3] (NULL = all 1), dconv [B,S,S,3,5+C] fp32 dense,
 * fully written (zero where no term depends on the value).  No workspace, deterministic. */
int ymx_yolo_loss_grad(const float *conv, int conv_ld, const float *label, long long label_stride,
                       const float *bboxes, long long bbox_stride, int B, int S, int C, int n_boxes,
                       float stride, float xyscale, const float *anchors /* host */,
                       float iou_loss_threshold, const float *term_weights /* device */,
                       float *dconv, void *stream);

/* ---- backward pass of the convolution block ------------------------------------
 * The reference trains through TensorFlow's automatic differentiation of YOLOConv /
 * ResidialBlock / Upsample (layers.py:33-106); these are the pieces of that derivative.
 * A training forward keeps the pre-activation z = conv(x) + bias next to the output
 * y = act(z) + residual: ymx_conv2d_fused_z writes both from one launch (layers without an
 * activation: ymx_conv2d_fused with act = 0, z = y).  Given dy:
 *   dz = dy * act'(z)                      ymx_act_backward   (act 0: dz = dy, no call needed)
 *   dbias[c] = sum_p dz[p][c]              ymx_bias_grad      (two ordered stages, deterministic)
 *   dkernel[r][s][ci][co] = sum_p x[p+(r,s)][ci] * dz[p][co]
 *                                          ymx_conv_wgrad     (HWIO fp32, the layout of the kernel;
 *                                          bf16 x and dz, channel counts multiples of 8; MFMA
 *                                          with transposing LDS reads; pixel slices summed in order)
 *   dx = conv(dz, kernel flipped in r,s and transposed in ci,co), for stride 2 on dz spread
 *        over a zero map of the input's size (even rows / columns): the forward entry points,
 *        with the kernel packed by ymx_pack_weights_dgrad (below)
 *   d residual = dy
 * ymx_upsample2x_backward is the adjoint of ymx_upsample2x_concat's resize half: dx [B,H,W,C]
 * from channels [c_off, c_off + C) of dy [B,2H,2W,dy_ld] (the skip half's gradient is the
 * other channel slice of dy as it is).  n / n_pix count elements / pixels; ld strides are in
 * elements, 0 = dense. */
int ymx_act_apply(const void *z, const void *residual, void *y, long long n, int act, int dtype, void *stream);
/* The SPP block (/root/reference/backbones.py:254-259) for a training step: the outputs of
 * ymx_spp_concat plus argmax [B,H,W,3,C] bytes, the row-major position inside its k x k window
 * (k = 13, 9, 5) of the first maximum each pooled value was taken from; the backward call
 * gives dx [B,H,W,C] from dy [B,H,W,4C]: the pass-through slice plus every pooled gradient at
 * its argmax (gathered per input element, fixed summation order). */
int ymx_spp_concat_train(const void *x, void *y, unsigned char *argmax, int B, int H, int W, int C, int dtype,
                         void *stream);
int ymx_spp_concat_backward(const void *dy, const unsigned char *argmax, void *dx, int B, int H, int W, int C,
                            int dtype, void *stream);
/* Batch normalisation in training mode (/root/reference/layers.py:4-15 under model.fit: Keras
 * normalises with the statistics of the batch and keeps moving averages for inference).  Dense
 * [n_pix, C] rows, bf16 or fp32; every sums vector holds 2 * C floats; workspaces of
 * ymx_bn_workspace_bytes(n_pix, C); sums are taken in two ordered stages (reproducible).
 *   ymx_bn_stats          sums = (sum_p u, sum_p u^2): the caller forms mean, the biased variance,
 *                         scale = gamma / sqrt(var + eps), shift = beta - mean * scale
 *   ymx_bn_act_forward    y = act(u * scale + shift) + residual (residual may be NULL)
 *   ymx_bn_act_backward   dz = dy * act'(u * scale + shift) stored in the tensors' type, and
 *                         sums = (sum_p dz, sum_p dz * u), from which dbeta = sum dz,
 *                         dgamma = s (sum dz u - mean sum dz), s = 1 / sqrt(var + eps)
 *   ymx_bn_backward_apply du = a[c] * dz + b[c] * u + c[c]; with N = n_pix the gradient of the
 *                         normalisation's input is a = gamma s, b = -gamma s^2 dgamma / N,
 *                         c = gamma s (s mean dgamma - dbeta) / N */
size_t ymx_bn_workspace_bytes(long long n_pix, int C);
int ymx_bn_stats(const void *u, long long n_pix, int C, int dtype, float *sums, void *workspace, size_t ws_bytes,
                 void *stream);
int ymx_bn_act_forward(const void *u, const float *scale, const float *shift, const void *residual, void *y,
                       long long n_pix, int C, int act, int dtype, void *stream);
int ymx_bn_act_backward(const void *u, const void *dy, const float *scale, const float *shift, void *dz,
                        long long n_pix, int C, int act, int dtype, float *sums, void *workspace, size_t ws_bytes,
                        void *stream);
int ymx_bn_backward_apply(const void *dz, const void *u, const float *a, const float *b, const float *c, void *du,
                          long long n_pix, int C, int dtype, void *stream);
int ymx_act_backward(const void *z, const void *dy, void *dz, long long n, int act, int dtype, void *stream);
/* ymx_act_backward and ymx_bias_grad in one pass over dense [n_pix, C] rows (bf16, act 1 or 2,
 * C / 8 a power of two up to 256; ..._workspace_bytes returns 0 for anything else). */
size_t ymx_act_backward_bias_workspace_bytes(long long n_pix, int C, int act, int dtype);
int ymx_act_backward_bias(const void *z, const void *dy, void *dz, long long n_pix, int C, int act, int dtype,
                          float *db, void *workspace, size_t ws_bytes, void *stream);
size_t ymx_bias_grad_workspace_bytes(long long n_pix, int C);
int ymx_bias_grad(const void *dz, long long n_pix, int C, int ld, int dtype, float *db,
                  void *workspace, size_t ws_bytes, void *stream);
size_t ymx_conv_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int k, int stride);
int ymx_conv_wgrad(const void *x, int x_ld, const void *dz, int dz_ld, float *dw, int B, int H, int W,
                   int Cin, int Cout, int k, int stride, void *workspace, size_t ws_bytes, void *stream);
/* Weight gradient of a 3x3 stride-1 layer with at most 3 input channels (the first layer of the
 * backbones, /root/reference/backbones.py:10): x [B,H,W,Cin] fp32 or bf16 as it is (no channel
 * padding), dz [B,H,W,dz_ld] bf16 (Cout a multiple of 8, at most 64), dw [3][3][Cin][Cout] fp32.
 * ymx_conv_wgrad_first_workspace_bytes returns 0 for shapes this entry point does not take (use
 * ymx_conv_wgrad on channel-padded tensors then).  Deterministic (ordered partial sums). */
size_t ymx_conv_wgrad_first_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int ymx_conv_wgrad_first(const void *x, int x_dtype, const void *dz, int dz_ld, float *dw, int B, int H, int W,
                         int Cin, int Cout, void *workspace, size_t ws_bytes, void *stream);
/* Which kernel ymx_conv_wgrad would take for dense tensors of this shape under the calling
 * thread's "wgrad_variant": "wgrad_slab" (all taps from one resident slab), "wgrad_s2" (all taps,
 * stride 2, planar staging), "wgrad_dma" (per tap, LDS-DMA) or "wgrad_reg" (per tap,
 * register-staged); NULL on a bad shape.  No counterpart
 * in the reference; for tests and tuning tools. */
const char *ymx_conv_wgrad_kernel_label(int B, int H, int W, int Cin, int Cout, int k, int stride);
int ymx_upsample2x_backward(const void *dy, void *dx, int B, int H, int W, int C, int dy_ld, int c_off,
                            int dtype, void *stream);
/* Packs, straight from the forward HWIO fp32 kernel [k][k][Cin][Cout], the kernel of the
 * input-gradient convolution: taps flipped, channel axes swapped, dz_channels (>= Cout, the
 * channel count dz is stored with; a multiple of 8) input channels of which those beyond Cout
 * are zero, Cin output channels.  w_packed has ymx_packed_weight_bytes(k, dz_channels, Cin,
 * dtype) bytes and feeds ymx_conv2d_fused(_ex) with Cin = dz_channels, Cout = Cin. */
int ymx_pack_weights_dgrad(const float *w_hwio, void *w_packed, int k, int Cin, int Cout, int dz_channels,
                           int dtype, void *stream);
/* ymx_pack_weights (into w_fwd) and ymx_pack_weights_dgrad (into w_dgrad) of one kernel in one
 * launch: what a training step does for every layer after each update of the weights.  Cin and
 * dz_channels must take the MFMA kernels (multiples of 8 for bf16, of 4 for fp32).  stride 2
 * (3x3, bf16): w_dgrad receives the form of ymx_pack_weights_dgrad_s2 below instead. */
int ymx_pack_weights_pair(const float *w_hwio, void *w_fwd, void *w_dgrad, int k, int Cin, int Cout,
                          int dz_channels, int stride, int dtype, void *stream);
/* Input gradient of a 3x3 stride-2 layer (layers.py:46-54: one zero row / column on the top / left,
 * VALID) without spreading dz over a zero map: input pixel (2 i + ph, 2 j + pw) only meets the
 * outputs (i + dr, j + dc), dr, dc in {0, 1}, so dx is a 2x2 convolution over dz with 4 Cin
 * output channels (one block of Cin per parity (ph, pw)) whose results are scattered
 * depth-to-space: 16 of the 36 multiply-adds per pixel the zero-map form spends (9 carry
 * kernel taps, the other 7 blocks of the packed kernel are zero).  ymx_pack_weights_dgrad_s2
 * packs that kernel from the forward HWIO fp32 kernel [3][3][Cin][Cout] (Cin a multiple of 8;
 * ymx_packed_weight_bytes(2, dz_channels, 4 * Cin, dtype) bytes; ymx_pack_weights_pair with
 * stride = 2 writes the same bytes); ymx_conv2d_dgrad_s2 writes dx [B,H,W,Cin] bf16 from dz
 * [B,OH,OW,dz_channels] bf16, OH = H / 2, OW = W / 2 (rounded down: the last row / column of an
 * odd-sized input takes no part in the forward pass; the caller zeroes it). */
int ymx_pack_weights_dgrad_s2(const float *w_hwio, void *w_packed, int Cin, int Cout, int dz_channels,
                              int dtype, void *stream);
int ymx_conv2d_dgrad_s2(const void *dz, const void *w_packed, void *dx, int B, int OH, int OW,
                        int dz_channels, int Cin, int H, int W, void *stream);
/* Training forward of a convolution block (/root/reference/layers.py:41-90 with the value in
 * front of the activation kept for the derivative): z = conv(x) * scale + shift rounded to bf16
 * [B,H',W',Cout], y = act(conv(x) * scale + shift) + residual (bf16, the same bits as
 * ymx_conv2d_fused gives), both dense.  bf16 compute; act 1 or 2; x_dtype as in
 * ymx_conv2d_fused_ex.  The LDS-DMA kernels store both from their epilogue; shapes that take
 * another kernel run as two launches (convolution into z, then ymx_act_apply). */
int ymx_conv2d_fused_z(const void *x, const void *w, const void *scale, const void *shift, const void *residual,
                       void *y, void *z, int B, int H, int W, int Cin, int Cout, int k, int stride, int act,
                       int x_dtype, void *stream);

/* ---- letterbox resize ---------------------------------------------------------
 * Replaces the image half of image_preprocess (utils.py:16-26): the image is resized by
 * scale = min(target/W, target/H) (bilinear, half-pixel centres, edges replicated: the
 * arithmetic of cv2.resize's default INTER_LINEAR on float images; 8-bit sources are
 * converted to float first, where OpenCV itself would use 11-bit fixed-point weights),
 * centred in a target x target square filled with 128, and divided by 256.
 *   images [B,H,W,3] fp32 or 8-bit (in_dtype YMX_F32 / YMX_U8), out [B,target,target,3]
 *   fp32 or bf16.  literal != 0 reproduces utils.py:21,25 as written: cv2.resize takes
 *   (width, height) but is handed (new_height, new_width), so a non-square image comes out
 *   with new_width rows and new_height columns and is pasted at rows pad_width.
 * The box half of image_preprocess is host arithmetic (dataset.letterbox_boxes). */
int ymx_letterbox(const void *images, int in_dtype, int B, int H, int W, void *out, int out_dtype,
                  int target, int literal, void *stream);

/* ---- box outlines -------------------------------------------------------------
 * Replaces the rectangle drawing of draw_bbox (utils.py:216-231) on a batch: for image b
 * the first valid[b] detections (boxes [B,max_boxes,4] normalised (ymin,xmin,ymax,xmax),
 * classes [B,max_boxes] fp32, as ymx_nms_combined writes them) are outlined in order, in
 * colors[class] (device, [C][3] 8-bit), `thickness` pixels wide (values below 1 draw 1),
 * on images [B,H,W,3] 8-bit, in place.  Corners are (int)(coordinate * size); a later box
 * paints over an earlier one.  No text labels: ymx_draw_boxes_labeled draws those. */
int ymx_draw_boxes(unsigned char *images, int B, int H, int W, const float *boxes,
                   const float *classes, const int *valid, int max_boxes,
                   const unsigned char *colors, int C, int thickness, void *stream);

/* The same with the text labels of draw_bbox (utils.py:233-240, the show_label branch):
 * labels [B,max_boxes,label_bytes] (device) holds one string per detection, zero-padded
 * (the reference's '%s: %.2f' % (class name, score), formatted by the caller; a string of
 * label_bytes characters needs no terminator, an empty one draws no label).  After its
 * outline each detection gets a rectangle filled with its class colour from (x1, y1) to
 * (x1 + text_width, y1 - text_height - 3), corners included, and the text in black with its
 * lowest row at y1 - 3.  The typeface is a built-in 5x7 bitmap in a 6-pixel cell, each font
 * pixel text_scale x text_scale image pixels (1..16): text_width = 6 * text_scale * length,
 * text_height = 7 * text_scale.  OpenCV's Hershey strokes and anti-aliasing
 * (cv2.putText, FONT_HERSHEY_SIMPLEX, LINE_AA) are not reproduced: the placement, colours
 * and painting order are the reference's, the glyph shapes are this library's.
 * label_bytes 1..256. */
int ymx_draw_boxes_labeled(unsigned char *images, int B, int H, int W, const float *boxes,
                           const float *classes, const int *valid, int max_boxes,
                           const unsigned char *colors, int C, int thickness,
                           const unsigned char *labels, int label_bytes, int text_scale,
                           void *stream);

/* ---- label assignment ---------------------------------------------------------
 * Replaces WIDERDataset.preprocess_true_boxes (dataset.py:387-482) for a batch: every
 * ground-truth box (xmin, ymin, xmax, ymax, class; boxes [B,max_boxes,5] fp32, counts [B] or
 * NULL = max_boxes each), in order, claims on each scale the anchors of its grid cell whose
 * IoU with it (in grid units) exceeds 0.3, or failing that the single best anchor of all
 * nine; a claimed label row becomes (cx, cy, w, h in pixels, 1, class probabilities smoothed
 * by 0.01) and the box is appended to that scale's ring of 128 ground-truth boxes.
 * labels: host array of 3 device pointers [B,S_i,S_i,3,5+C] fp32, S_i = input_size /
 * strides[i]; gt_boxes: host array of 3 device pointers [B,128,4] fp32; both are zeroed by
 * the call.  anchors_grid: host [3][3][2], anchor sizes in grid units (pixels / stride).
 * Arithmetic in double, as the reference's numpy code.  Boxes whose class is outside
 * [0, C) or whose cell falls outside the grid are skipped (the reference raises). */
int ymx_assign_labels(const float *boxes, const int *counts, int B, int max_boxes, int input_size,
                      int C, const float *strides /* host */, const float *anchors_grid /* host */,
                      float *const *labels /* host */, float *const *gt_boxes /* host */, void *stream);

/* ---- box IoU ----------------------------------------------------------------
 * Replaces bbox_iou (utils.py:33-63) and, with giou != 0, bbox_giou
 * (utils.py:65-105) on n pairs of centre-format (cx,cy,w,h) fp32 boxes that the
 * caller has already broadcast to a common shape; zero unions divide to 0
 * (tf.math.divide_no_nan). */
int ymx_bbox_iou(const float *b1, const float *b2, float *out, long long n, int giou,
                 void *stream);

/* ---- CRC-32C of a host buffer ---------------------------------------------------
 * The checksum of TFRecord framing (to_tfrecord, dataset.py:64-126, writes through
 * tf.io.TFRecordWriter): Castagnoli polynomial, reflected, initial value and final xor
 * 0xFFFFFFFF; crc = the checksum of the bytes before `data` (0 for none), so a buffer can be
 * fed in pieces.  Host memory, no device work. */
unsigned ymx_crc32c(const void *data, size_t n, unsigned crc);

#ifdef __cplusplus
}
#endif
#endif /* YMX_H */
