// Internal declarations shared by the kernel translation units and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stddef.h>
#include "../../include/ymx.h"

namespace ymx {

// ---- error reporting (thread-local message behind ymx_last_error) ------------
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);
static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// ---- per-device kernel attributes ---------------------------------------------
struct LdsReserve { std::atomic<unsigned long long> done{0}; };
int reserve_lds(LdsReserve &st, const void *kern, int bytes, const char *what);

// ---- element helpers ---------------------------------------------------------
typedef uint16_t bf16_t;   // raw bfloat16 bits in memory

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// Round to nearest even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and
// keeps NaN a NaN.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// 16-byte streaming stores of a layer's output: the non-temporal hint keeps 100s of MB of
// activations from displacing the weights and the input tiles other workgroups are still
// re-reading in L2 (measured: -1.6 % on the detector step).
typedef __attribute__((ext_vector_type(4))) unsigned int st_u32x4_t;
typedef __attribute__((ext_vector_type(4))) float st_f32x4_t;
__device__ __forceinline__ void store_stream16(void *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const st_u32x4_t v = {a, b, c, d};
    __builtin_nontemporal_store(v, (st_u32x4_t *)p);
}
__device__ __forceinline__ void store_stream16(void *p, float a, float b, float c, float d) {
    const st_f32x4_t v = {a, b, c, d};
    __builtin_nontemporal_store(v, (st_f32x4_t *)p);
}

template <typename T> struct elem;
template <> struct elem<float> {
    static constexpr int code = YMX_F32;
    __device__ static __forceinline__ float load(const float *p) { return *p; }
    __device__ static __forceinline__ void store(float *p, float v) { *p = v; }
};
template <> struct elem<bf16_t> {
    static constexpr int code = YMX_BF16;
    __device__ static __forceinline__ float load(const bf16_t *p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store(bf16_t *p, float v) { *p = f32_to_bf16(v); }
};

// Workgroups are dealt round-robin over the 8 XCDs, so ids b and b+8 share an L2.
// Give every XCD one contiguous run of logical tile ids (bijective for any grid).
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// x * tanh(softplus(x)) with one exp and one reciprocal:
//   tanh(log(1+e)) = n/(n+2),  n = e*(e+2),  e = exp(x).
// v_exp_f32 and v_rcp_f32 are 1-ulp instructions; the clamp keeps n finite (for
// x >= 20 the ratio is 1 to fp32 precision, so the result is x), and the product form
// keeps full relative accuracy for very negative x, where the result is ~ x*e^x.
//
// Floating-point contraction is switched off inside these helpers and the affine step is an
// explicit fma: every convolution kernel, whatever its tiling and however the compiler
// sees the surrounding code, must produce the same bits for the same accumulator.
__device__ __forceinline__ float affine_f(float v, float sc, float sh) { return __builtin_fmaf(v, sc, sh); }
__device__ __forceinline__ float mish_f(float x) {
#pragma clang fp contract(off)
    const float e = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.44269504088896f);
    const float n = e * (e + 2.0f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
}
// Two values at once: the multiplies and adds become packed fp32 instructions (v_pk_mul_f32,
// v_pk_add_f32: two IEEE-exact results per issue slot), the exponential and the reciprocal
// stay scalar.  Bit-identical to mish_f on each element.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t mish2_f(f32x2_t x) {
#pragma clang fp contract(off)
    const f32x2_t c = {fminf(x.x, 20.0f), fminf(x.y, 20.0f)};
    const f32x2_t t = c * 1.44269504088896f;
    const f32x2_t e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const f32x2_t n = e * (e + 2.0f);
    const f32x2_t d = n + 2.0f;
    const f32x2_t r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return x * (n * r);
}
__device__ __forceinline__ float act_f(float v, int act) {
#pragma clang fp contract(off)
    if (act == YMX_ACT_MISH) return mish_f(v);
    if (act == YMX_ACT_LEAKY) return v > 0.0f ? v : 0.1f * v;
    return v;
}
template <int ACT> __device__ __forceinline__ float act_c(float v) {
#pragma clang fp contract(off)
    if constexpr (ACT == YMX_ACT_MISH) return mish_f(v);
    if constexpr (ACT == YMX_ACT_LEAKY) return v > 0.0f ? v : 0.1f * v;
    return v;
}
// act(v * sc + sh) for a pair (the affine step is one packed fma, as in affine_f).
__device__ __forceinline__ f32x2_t affine2(f32x2_t v, f32x2_t sc, f32x2_t sh) { return __builtin_elementwise_fma(v, sc, sh); }
template <int ACT> __device__ __forceinline__ f32x2_t act2(f32x2_t x) {
    if constexpr (ACT == YMX_ACT_MISH) return mish2_f(x);
    f32x2_t o = {act_c<ACT>(x.x), act_c<ACT>(x.y)};
    return o;
}
template <int ACT> __device__ __forceinline__ f32x2_t affine_act2(f32x2_t v, f32x2_t sc, f32x2_t sh) {
    return act2<ACT>(affine2(v, sc, sh));
}
// 1 / (1 + e^-x) with the hardware exponential and reciprocal (1 ulp each): four instructions
// instead of the dozen a correctly rounded division takes; the decode kernel spends most of
// its arithmetic here.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---- launchers (one per kernel family; defined next to the kernels) -----------
struct ConvArgs {
    const void *x, *w, *res;
    const float *scale, *shift;
    void *y;
    void *z;                     // bf16 [P][Cout], dense, or null: second output, the value before the activation
                                 // (what a training step keeps for the derivative); LDS-DMA kernels only
    int B, H, W, Cin, Cout, OH, OW;
    int k, stride, pad;
    int x_ld, res_ld, y_ld;      // pixel strides in elements
    int P;                       // B*OH*OW output pixels
    int K;                       // k*k*Cin
    int Kw;                      // row stride of the packed weights (elements)
    int act;
    int dtype, x_dtype, out_dtype;
    unsigned x_bytes;            // extent of x in bytes (buffer-load bounds), <= 2^31
    unsigned w_bytes;            // extent of the packed weights in bytes
    unsigned res_bytes;          // extent of the shortcut input in bytes (0: none, or beyond 4 GiB)
    unsigned ohw_m, ohw_s;       // p / (OH*OW) and p / OW for 0 <= p < 2^31 without a division:
    unsigned ow_m, ow_s;         //   m == 0 ? p : __umulhi(p, m) >> s   (set_fastdiv below)
    // K order of the implicit GEMM (= column order of the packed weights): channel blocks
    // of CB channels outermost, then the k*k filter taps, then the channels of the block
    //   K index = blk * (k*k*CB) + tap * CB + c,   input channel = blk * CB + c
    // with CB = 64 when Cin is a multiple of 64 and CB = Cin (one block) otherwise.  Every
    // MFMA kernel walks K in this order, so all of them add the same terms in the same
    // order; the block-major order lets a kernel keep one block's input pixels in LDS and
    // reuse them for all taps.
    // Depth-to-space output (d2s_c != 0; the input gradient of a stride-2 layer): the Cout = 4 * d2s_c
    // channels of output pixel (b, oh, ow) are the d2s_c channels of the four pixels
    // (2 oh + ph, 2 ow + pw), block q = 2 ph + pw, of y [B, d2s_H, d2s_W, d2s_c]
    int d2s_c, d2s_H, d2s_W;
    unsigned d2s_m, d2s_s;       // fastdiv constants of d2s_c
    int CB, kb;                  // channel block, k*k*CB
    unsigned kb_m, kb_s;         // fastdiv constants: K index / kb, (K index % kb) / CB, tap / k
    unsigned cb_m, cb_s;
    unsigned k_m, k_s;
};

// Multiply-shift division of a 31-bit dividend by d >= 1: with l = ceil(log2 d) and
// m = ceil(2^(31+l) / d) (< 2^32), floor(n / d) = (n * m) >> (31 + l) for all n < 2^31,
// because the excess n * (m*d - 2^(31+l)) / (d * 2^(31+l)) stays below 1/d.  d = 1 is
// flagged by m = 0.
static inline void set_fastdiv(unsigned d, unsigned &m, unsigned &s) {
    if (d <= 1) { m = 0; s = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    m = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
    s = l - 1;
}
__device__ __forceinline__ int fastdiv(int n, unsigned m, unsigned s) {
    return m == 0 ? n : (int)(__umulhi((unsigned)n, m) >> s);
}

static inline int conv_channel_block(int Cin) { return Cin % 64 == 0 ? 64 : Cin; }

// K index (a multiple of the 16-byte chunk) -> element offset of its tap and channel
// relative to a pixel's (ih0, iw0, 0), and the tap's bit for the validity masks of
// pixel_geometry; indices past the end of K get no bit (their loads read zeros).
__device__ __forceinline__ void k_decode(const ConvArgs &a, int kpos, int &tapoff, unsigned &tapbit) {
    const int blk = fastdiv(kpos, a.kb_m, a.kb_s), rem = kpos - blk * a.kb;
    const int tap = fastdiv(rem, a.cb_m, a.cb_s), ci = blk * a.CB + (rem - tap * a.CB);
    const int tr = fastdiv(tap, a.k_m, a.k_s), tc = tap - tr * a.k;
    tapoff = (tr * a.W + tc) * a.x_ld + ci;
    tapbit = kpos < a.K ? (1u << tap) : 0u;
}

// Where output pixel p reads its input: element offset of (b, ih0, iw0, channel 0) -- it may
// be negative -- and the bit mask (bit r*k+c) of the taps that fall inside the image.
__device__ __forceinline__ void pixel_geometry(const ConvArgs &a, int p, int &off, unsigned &mask) {
    off = 0;
    mask = 0;
    if (p < a.P) {
        const int b = fastdiv(p, a.ohw_m, a.ohw_s), rem = p - b * (a.OH * a.OW);
        const int oh = fastdiv(rem, a.ow_m, a.ow_s), ow = rem - oh * a.OW;
        const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;
        off = ((b * a.H + ih0) * a.W + iw0) * a.x_ld;
        const int clo = max(0, -iw0), chi = max(0, min(a.k, a.W - iw0));
        const unsigned cols = ((1u << chi) - 1u) & ~((1u << clo) - 1u);
        for (int r = 0; r < a.k; ++r)
            mask |= ((unsigned)(ih0 + r) < (unsigned)a.H ? cols : 0u) << (r * a.k);
    }
}

// Element offset in y of channels co.. (a run that does not cross a multiple of d2s_c) of output pixel p.
__device__ __forceinline__ size_t conv_out_offset(const ConvArgs &a, int p, int co) {
    if (a.d2s_c == 0) return (size_t)p * a.y_ld + co;
    const int b = fastdiv(p, a.ohw_m, a.ohw_s), rem = p - b * (a.OH * a.OW);
    const int oh = fastdiv(rem, a.ow_m, a.ow_s), ow = rem - oh * a.OW;
    const int q = fastdiv(co, a.d2s_m, a.d2s_s), ci = co - q * a.d2s_c;
    return (((size_t)(b * a.d2s_H + 2 * oh + (q >> 1)) * a.d2s_W) + 2 * ow + (q & 1)) * a.d2s_c + ci;
}

// Packed weight geometry, shared by pack and conv.
static inline bool conv_uses_mfma(int Cin, int dtype) {
    return dtype == YMX_BF16 ? (Cin % 8 == 0) : (Cin % 4 == 0);
}
static inline int packed_kw(int k, int Cin) { return ((k * k * Cin + 63) / 64) * 64; }
static inline int packed_rows(int Cout) { return ((Cout + 127) / 128) * 128; }

int launch_conv_igemm(const ConvArgs &a, hipStream_t s);
int launch_conv_igemm_dma(const ConvArgs &a, hipStream_t s);
int launch_conv_igemm_dma2(const ConvArgs &a, hipStream_t s);
int launch_conv_igemm_dma4(const ConvArgs &a, hipStream_t s);
int launch_conv_igemm_halo(const ConvArgs &a, hipStream_t s);
bool conv_halo_eligible(const ConvArgs &a);
int conv_halo_tile_pixels(const ConvArgs &a);         // 256, 192, or 448 = both forms in one launch
struct TilePlan { int pixels, big_tiles, big_pixels; };   // mixed form (448): the first big_tiles workgroups cover big_pixels pixels with 256-pixel tiles
TilePlan plan_wide_tiles(long long P, long long ntq, double f192, int forced_pixels, int forced_rounds);
TilePlan conv_halo_plan(const ConvArgs &a);
TilePlan conv_dma4_plan(const ConvArgs &a);
extern thread_local int g_halo_pixels, g_halo_big_rounds, g_tile_mix;
int conv_dma4_tile_pixels(const ConvArgs &a);
extern thread_local int g_dma4_pixels;
extern thread_local int g_wgrad_variant;
int launch_conv_igemm_slab2(const ConvArgs &a, hipStream_t s);
bool conv_slab2_eligible(const ConvArgs &a);
extern thread_local int g_dma2_stages;     // per calling thread, like every tuning option
bool conv_first_rows_eligible(const ConvArgs &a);
bool conv_stem_supported(int act);
int launch_conv_stem(const float *x, const float *w0, const float *scale0, const float *shift0, const void *w1, const float *scale1,
                     const float *shift1, void *y, int B, int H, int W, int act, int y_ld, hipStream_t s);
bool conv_resblock_supported(int act);
int launch_conv_resblock(const void *x, int x_ld, const void *w1, const float *scale1, const float *shift1, const void *w2,
                         const float *scale2, const float *shift2, void *y, int y_ld, int B, int H, int W, int act, hipStream_t s);
int launch_conv_resblock128(const void *x, int x_ld, const void *w1, const float *scale1, const float *shift1, const void *w2,
                            const float *scale2, const float *shift2, void *y, int y_ld, int B, int H, int W, int act, hipStream_t s);
int launch_conv_direct(const ConvArgs &a, hipStream_t s);
int launch_pack_weights(const float *w_hwio, void *w_packed, int k, int Cin, int Cout, int dtype,
                        hipStream_t s, int src_cout = 0);
int launch_pack_weights_pair(const float *w_hwio, void *w_fwd, void *w_bwd, int k, int Cin, int Cout, int dz_channels, int stride,
                             int dtype, hipStream_t s);
int launch_pack_weights_dgrad_s2(const float *w_hwio, void *w_packed, int Cin, int Cout, int dz_channels, int dtype, hipStream_t s);
int launch_upsample2x_concat(const void *skip, const void *x, void *y, int B, int H, int W, int C1,
                             int C2, int dtype, hipStream_t s);
int launch_affine_act(const void *x, const float *scale, const float *shift, void *y, long long n, int C, int act,
                      int dtype, hipStream_t s, const void *res = nullptr);
int launch_bbox_iou(const float *b1, const float *b2, float *out, long long n, int giou,
                    hipStream_t s);
int launch_spp_concat(const void *x, void *y, int B, int H, int W, int C, int dtype, hipStream_t s);
int launch_spp_train(const void *x, void *y, unsigned char *arg, int B, int H, int W, int C, int dtype, hipStream_t s);
int launch_spp_bwd(const void *dy, const unsigned char *arg, void *dx, int B, int H, int W, int C, int dtype, hipStream_t s);
int launch_letterbox(const void *src, int in_u8, int B, int H, int W, void *dst, int out_bf16, int T, int literal, hipStream_t s);
int launch_draw_boxes(unsigned char *img, int B, int H, int W, const float *boxes, const float *classes, const int *valid, int M,
                      const unsigned char *colors, int C, int thick, const unsigned char *labels, int L, int scale,
                      hipStream_t s);
int launch_act(const void *z, const void *other, void *y, size_t n, int act, int dtype, int mode, hipStream_t s);
size_t bias_grad_workspace_bytes(long long P, int C);
bool act_bwd_bias_supported(long long P, int C, int act, int dtype);
size_t act_bwd_bias_workspace_bytes(long long P, int C);
int launch_act_bwd_bias(const void *z, const void *dy, void *dz, long long P, int C, int act, float *db, void *ws, hipStream_t s);
int launch_bias_grad(const void *dz, long long P, int C, int ld, int dtype, float *db, void *ws, hipStream_t s);
size_t bn_workspace_bytes(long long P, int C);
int launch_bn_reduce(const void *u, const void *dy, void *dz, const float *scale, const float *shift, long long P, int C, int act,
                     int dtype, float *sums, void *ws, hipStream_t s);
int launch_bn_bwd_apply(const void *dz, const void *u, const float *a, const float *b, const float *c, void *du, long long P, int C,
                        int dtype, hipStream_t s);
size_t conv_wgrad_workspace_bytes(int k, int Cin, int Cout, long long P);
size_t conv_wgrad_s2_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int launch_conv_wgrad(const void *x, int x_ld, const void *dz, int dz_ld, float *dw, int B, int H, int W, int Cin, int Cout, int k,
                      int stride, void *ws, hipStream_t s);
const char *conv_wgrad_label(int B, int H, int W, int Cin, int Cout, int k, int stride);
bool conv_wgrad_first_supported(int B, int H, int W, int Cin, int Cout);
size_t conv_wgrad_first_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int launch_conv_wgrad_first(const void *x, int x_f32, const void *dz, int dz_ld, float *dw, int B, int H, int W, int Cin, int Cout, void *ws,
                            hipStream_t s);
int launch_upsample2x_bwd(const void *dy, void *dx, int B, int H, int W, int C, int dy_ld, int c_off, int dtype, hipStream_t s);
size_t yolo_loss_workspace_bytes(int B, int S);
int launch_yolo_loss_grad(const float *conv, int conv_ld, const float *label, long long label_stride, const float *bboxes,
                          long long bbox_stride, int B, int S, int C, int NB, float stride, float xyscale, const float *anchors,
                          float iou_thr, const float *weights, float *dconv, hipStream_t s);
int launch_yolo_loss(const float *conv, int conv_ld, const float *pred, const float *label, long long label_stride,
                     const float *bboxes, long long bbox_stride, int B, int S, int C, int NB, float stride, float xyscale, const float *anchors, float iou_thr, float *out,
                     float *per_image, void *workspace, hipStream_t s);

// A decode workgroup (256 threads) owns a segment of DEC_SEG consecutive candidates of one
// scale of one image, walked in DEC_SS sub-steps of 64.
#ifndef YMX_DEC_SS
#define YMX_DEC_SS 1
#endif
constexpr int DEC_SS = YMX_DEC_SS, DEC_SEG = 64 * DEC_SS;

struct DecodeArgs {
    const void *head[3];
    unsigned head_bytes[3];      // byte extent of each fp32 map if it can be addressed as one buffer (< 4 GiB), else 0
    int S[3];
    int seg_begin[4];            // first segment of each scale (segments of DEC_SEG candidates)
    int cand_begin[4];           // first candidate of each scale in the concatenated list
    float stride[3], xyscale[3], anchor[3][3][2];
    int cell_ld[3];              // elements from one grid cell to the next (>= 3*(5+C)); 0 = dense
    int max_pad;                 // largest cell_ld - 3*(5+C) over the scales (sizes the staging buffer)
    int n_scales, B, C, N, cap, head_dtype;
    int filter;                  // 0: decode only, 1: decode+filter, 2: filter decoded input
    float thr, image_size;
    const float *in_xywh, *in_scores;    // filter == 2
    long long sb, sn, sc;
    float *boxes, *scores;
    float *conf;                 // filter == 0 only: if set, scores = class prob alone and conf[b][n] = objectness
    int *src_idx, *counts;
    unsigned long long *state;   // [B][nseg] look-back words
    int *ticket;                 // [B]
    int *err;                    // [1]
    int nseg;
};
int launch_decode(const DecodeArgs &a, hipStream_t s);

struct NmsArgs {
    const float *boxes, *scores;
    const int *counts, *src_idx;
    int B, cap, C, per_class, total;
    float iou_thr, score_thr;
    float *out_boxes, *out_scores, *out_classes;
    int *out_valid, *out_keep;
    unsigned long long *kept_keys;   // [B][C][per_class]
    int *kept_cnt;                   // [B][C]
};
int launch_nms(const NmsArgs &a, hipStream_t s);
size_t nms_workspace_bytes(int B, int C, int per_class);
// argument checks + workspace carving + launch of ymx_nms_combined, shared by every NMS entry point
int nms_entry(const float *boxes, const float *scores, const int *counts, const int *src_idx, int B, int cap, int C,
              int per_class, int total, float iou_thr, float score_thr, float *out_boxes, float *out_scores,
              float *out_classes, int *out_valid, int *out_keep_idx, void *workspace, size_t ws_bytes, hipStream_t s);

}  // namespace ymx
