// extern "C" surface of libymx: argument validation, kernel selection, error strings.
// No device memory is allocated or freed here and nothing synchronises; the only
// runtime calls besides launches are hipMemsetAsync on caller-provided workspace.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "common.h"

namespace ymx {

static thread_local std::string g_err;
static thread_local float *g_split_conf = nullptr;   // set only inside ymx_decode_split
static thread_local const int *g_cell_ld = nullptr;  // set only inside ymx_decode_filter_strided

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(YMX_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return YMX_OK;
}


// Tuning options (ymx_set_option) are per calling thread: a thread that changes one affects
// only its own later calls, so concurrent callers never see each other's settings and
// nothing here needs a lock.  Every thread starts from the defaults.
struct Tuning {
    long long variant = 5;      // wide layers: 5 = resident-slab kernel for 3x3 stride 1, phase-staggered kernel
                                // otherwise; 1, 2, 4: see ymx_set_option
    long long narrow = 2;       // layers of <= 128 output channels: 2 = LDS-DMA kernels (resident input slab for
                                // 3x3 stride 1 with 32/64 input channels), 1 = without the slab kernel, 0 = register-staged
    long long min_tiles = -1;   // < 0: the default below
};
static thread_local Tuning t_opt;
thread_local int g_dma2_stages = 4;    // pipeline depth of the 32- and 64-channel narrow tiles (3 or 4)

// Fewest 256x256 tiles for which the large-tile kernel is chosen (one workgroup per CU,
// 256 CUs).  YMX_DMA_MIN_TILES, read once per process, overrides the default of 160 for
// experiments (a huge value disables the kernel, 0 forces it).
static long long default_min_tiles() {
    static const long long v = [] {
        const char *e = getenv("YMX_DMA_MIN_TILES");
        return e ? atoll(e) : 160ll;
    }();
    return v;
}
static long long dma_min_tiles() { return t_opt.min_tiles >= 0 ? t_opt.min_tiles : default_min_tiles(); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel.  The
// caller keeps one LdsReserve per kernel instantiation; bit d records that device d has
// been set up.  Setting the attribute twice is harmless, so two threads racing here are too.
int reserve_lds(LdsReserve &st, const void *kern, int bytes, const char *what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return fail(YMX_ELAUNCH, "%s: no current device", what);
    }
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && ((st.done.load(std::memory_order_acquire) >> dev) & 1ull)) return YMX_OK;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(YMX_ELAUNCH, "%s: cannot reserve %d bytes of LDS", what, bytes);
    }
    if (tracked) st.done.fetch_or(1ull << dev, std::memory_order_release);
    return YMX_OK;
}

static int conv_impl(const void *x, const void *w, const void *scale, const void *shift,
                     const void *residual, void *y, int B, int H, int W, int Cin, int Cout, int k,
                     int stride, int act, int dtype, int x_dtype, int out_dtype, int x_ld, int res_ld,
                     int y_ld, hipStream_t s, int op_min_tiles = 0, const char **label = nullptr, void *z = nullptr,
                     const int *d2s = nullptr) {
    // label != nullptr: validate and select only; *label names the kernel family, nothing is launched
    // z != nullptr: the value before the activation is stored too (bf16, dense); kernels without
    // that second output run as two launches, convolution into z and then y = act(z) + residual
    // d2s = {H', W', C}: a 2x2 window (taps at +0 / +1, zeros beyond the bottom / right edge) whose Cout = 4 C
    // outputs per pixel are scattered depth-to-space into y [B, H', W', C]
    if (!x || !w || !y) return fail(YMX_EARG, "conv2d: null x, w or y");
    if (op_min_tiles < 0) return fail(YMX_EARG, "conv2d: negative min_tiles");
    const long long min_tiles = op_min_tiles > 0 ? op_min_tiles : dma_min_tiles();
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return fail(YMX_EARG, "conv2d: non-positive size");
    if (k < 1 || k > 5 || ((k & 1) == 0 && !(d2s && k == 2))) return fail(YMX_EUNSUPPORTED, "conv2d: kernel size %d (odd, <= 5)", k);
    if (stride != 1 && stride != 2) return fail(YMX_EUNSUPPORTED, "conv2d: stride %d (1 or 2)", stride);
    if (act != YMX_ACT_NONE && act != YMX_ACT_LEAKY && act != YMX_ACT_MISH)
        return fail(YMX_EARG, "conv2d: activation code %d", act);
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "conv2d: dtype %d", dtype);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.res = residual; a.y = y;
    a.scale = (const float *)scale; a.shift = (const float *)shift;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.k = k; a.stride = stride;
    if (d2s) {
        if (stride != 1 || act != YMX_ACT_NONE || residual || dtype != YMX_BF16 || out_dtype != YMX_BF16 || y_ld > 0 || z)
            return fail(YMX_EARG, "conv2d: the depth-to-space form is bf16, stride 1, linear, without shortcut");
        if (!conv_uses_mfma(Cin, dtype)) return fail(YMX_EARG, "conv2d: the depth-to-space form needs input channels in multiples of 8, got %d", Cin);
        if (d2s[2] <= 0 || d2s[2] % 8 || Cout != 4 * d2s[2] || d2s[0] < 2 * H || d2s[0] > 2 * H + 1 || d2s[1] < 2 * W || d2s[1] > 2 * W + 1)
            return fail(YMX_EARG, "conv2d: depth-to-space output [%d,%d,%d] does not match %d x %d x %d", d2s[0], d2s[1], d2s[2], H, W, Cout);
        a.pad = 0; a.OH = H; a.OW = W;
        a.d2s_H = d2s[0]; a.d2s_W = d2s[1]; a.d2s_c = d2s[2];
        set_fastdiv((unsigned)a.d2s_c, a.d2s_m, a.d2s_s);
    } else if (stride == 1) {
        a.pad = (k - 1) / 2; a.OH = H; a.OW = W;                   // SAME
    } else {
        a.pad = 1;                                                 // top/left only, then VALID
        if (H + 1 < k || W + 1 < k) return fail(YMX_EARG, "conv2d: input smaller than kernel");
        a.OH = (H + 1 - k) / 2 + 1; a.OW = (W + 1 - k) / 2 + 1;
    }
    a.x_ld = x_ld > 0 ? x_ld : Cin;
    a.res_ld = res_ld > 0 ? res_ld : Cout;
    a.y_ld = y_ld > 0 ? y_ld : Cout;
    if (a.x_ld < Cin || a.y_ld < Cout || a.res_ld < Cout) return fail(YMX_EARG, "conv2d: pixel stride below channel count");
    const long long P = (long long)B * a.OH * a.OW;
    // pixel counts and input offsets are 32-bit in the kernels; output offsets are size_t
    if (P > 0x7fffffffLL || (long long)B * H * W * a.x_ld > 0x7fffffffLL)
        return fail(YMX_EUNSUPPORTED, "conv2d: more than 2^31 output pixels or input elements");
    a.P = (int)P;
    a.K = k * k * Cin;
    set_fastdiv((unsigned)(a.OH * a.OW), a.ohw_m, a.ohw_s);
    set_fastdiv((unsigned)a.OW, a.ow_m, a.ow_s);
    a.CB = conv_channel_block(Cin);
    a.kb = k * k * a.CB;
    set_fastdiv((unsigned)a.kb, a.kb_m, a.kb_s);
    set_fastdiv((unsigned)a.CB, a.cb_m, a.cb_s);
    set_fastdiv((unsigned)k, a.k_m, a.k_s);
    a.act = act;
    a.dtype = dtype; a.x_dtype = x_dtype; a.out_dtype = out_dtype;
    if (z) {
        if (dtype != YMX_BF16 || out_dtype != YMX_BF16 || act == YMX_ACT_NONE || a.y_ld != Cout || !aligned16(z))
            return fail(YMX_EARG, "conv2d: the second output needs bf16, an activation, dense y and a 16-byte aligned z");
    }
    // the second output inside the kernel: the 16-byte store path of the LDS-DMA epilogue
    const bool z_fused = z && Cout % 8 == 0 && (!residual || a.res_ld % 8 == 0);
    auto two_pass = [&](int (*launch)(const ConvArgs &, hipStream_t)) {
        ConvArgs c = a;
        c.y = z; c.res = nullptr; c.res_bytes = 0; c.act = YMX_ACT_NONE; c.y_ld = Cout;
        const int rc = launch(c, s);
        if (rc) return rc;
        if (residual && a.res_ld != Cout) return fail(YMX_EUNSUPPORTED, "conv2d: second output with a strided residual");
        return launch_act(z, residual, y, (size_t)P * Cout, act, YMX_BF16, 0, s);
    };
    if (conv_uses_mfma(Cin, dtype)) {
        if (x_dtype != dtype) return fail(YMX_EUNSUPPORTED, "conv2d: MFMA path needs x_dtype == dtype");
        const int kch = dtype == YMX_BF16 ? 8 : 4;
        if (a.x_ld % kch) return fail(YMX_EARG, "conv2d: x pixel stride %d not a multiple of %d", a.x_ld, kch);
        if (!aligned16(x) || !aligned16(w)) return fail(YMX_EARG, "conv2d: x and w must be 16-byte aligned");
        if (!aligned16(y) || (residual && !aligned16(residual))) return fail(YMX_EARG, "conv2d: y/residual must be 16-byte aligned");
        a.Kw = packed_kw(k, Cin);
        const long long xb = (((long long)B * H * W - 1) * a.x_ld + Cin) * (dtype == YMX_BF16 ? 2 : 4);
        if (xb > 0x80000000LL) return fail(YMX_EUNSUPPORTED, "conv2d: input larger than 2 GiB (%lld bytes)", xb);
        a.x_bytes = (unsigned)xb;
        a.w_bytes = (unsigned)((size_t)packed_rows(Cout) * a.Kw * (dtype == YMX_BF16 ? 2 : 4));
        if (residual) {
            const long long rb = ((P - 1) * a.res_ld + Cout) * (dtype == YMX_BF16 ? 2 : 4);
            a.res_bytes = rb < 0x80000000LL ? (unsigned)rb : 0u;
        }
        // Large-tile LDS-DMA kernel when the layer is wide enough to fill 256-channel tiles
        // and has enough of them to occupy the chip; the 128x128 kernel otherwise.
        if (dtype == YMX_BF16 && Cout > 128) {
            const long long tiles = ((long long)(a.P + 255) / 256) * ((Cout + 255) / 256);
            if (tiles >= min_tiles) {
                // 3x3 stride 1: input pixels resident in LDS across the taps; everything else:
                // phase-staggered wave groups (2-5 % ahead of the two-stage kernel on every wide
                // 1x1 and stride-2 layer of the detector since its loop has four intervals)
                int which = (int)t_opt.variant;
                if (which == 5) which = conv_halo_eligible(a) ? 5 : 4;
                static const char *const names[6] = {"", "dma", "dma2", "", "dma4", "halo"};
                if (label) {
                    *label = (which == 5 && conv_halo_tile_pixels(a) == 192) ? "halo192" :
                             (which == 5 && conv_halo_tile_pixels(a) == 448) ? "halo_mixed" :
                             (which == 4 && conv_dma4_tile_pixels(a) == 192) ? "dma4_192" :
                             (which == 4 && conv_dma4_tile_pixels(a) == 448) ? "dma4_mixed" : names[which];
                    return YMX_OK;
                }
                int (*const fn)(const ConvArgs &, hipStream_t) = which == 1 ? launch_conv_igemm_dma : which == 2 ? launch_conv_igemm_dma2 :
                                                                  which == 4 ? launch_conv_igemm_dma4 : launch_conv_igemm_halo;
                if (z && !(z_fused && (!residual || a.res_bytes))) return two_pass(fn);
                a.z = z;
                return fn(a, s);
            }
        }
        // Narrow layers (the first backbone stages): memory-bound; the several-workgroups-
        // per-CU LDS-DMA kernel with channel tiles of 32/64/128.
        if (dtype == YMX_BF16 && Cout <= 128 && t_opt.narrow && (a.P + 255) / 256 >= min_tiles) {
            const bool slab = t_opt.narrow == 2 && conv_slab2_eligible(a);
            if (label) { *label = slab ? "slab2" : "dma2"; return YMX_OK; }
            int (*const fn)(const ConvArgs &, hipStream_t) = slab ? launch_conv_igemm_slab2 : launch_conv_igemm_dma2;
            if (z && !(z_fused && (!residual || a.res_bytes))) return two_pass(fn);
            a.z = z;
            return fn(a, s);
        }
        if (label) { *label = "igemm"; return YMX_OK; }
        return z ? two_pass(launch_conv_igemm) : launch_conv_igemm(a, s);
    }
    if (!aligned16(y)) return fail(YMX_EARG, "conv2d: y must be 16-byte aligned");
    a.Kw = Cout;
    {
        // extent of x for the bounds-checked loads of the first-layer kernel (0 = unknown:
        // inputs beyond 2 GiB fall back to the plain direct kernel)
        const long long xb = (((long long)B * H * W - 1) * a.x_ld + Cin) * (x_dtype == YMX_BF16 ? 2 : 4);
        a.x_bytes = (xb <= 0x80000000LL && aligned16(x)) ? (unsigned)xb : 0u;
    }
    if (label) { *label = conv_first_rows_eligible(a) ? "first_rows" : "direct"; return YMX_OK; }
    return z ? two_pass(launch_conv_direct) : launch_conv_direct(a, s);
}

}  // namespace ymx

using namespace ymx;

extern "C" {

int ymx_version(void) { return YMX_VERSION; }

int ymx_set_option(const char *name, long long value) {
    if (!name) return fail(YMX_EARG, "set_option: null name");
    if (!strcmp(name, "dma_min_tiles")) {
        if (value < 0) return fail(YMX_EARG, "set_option: dma_min_tiles must be >= 0");
        t_opt.min_tiles = value;
        return YMX_OK;
    }
    if (!strcmp(name, "dma_variant")) {
        if (value < 1 || value > 5 || value == 3) return fail(YMX_EARG, "set_option: dma_variant must be 1, 2, 4 or 5");
        t_opt.variant = value;
        return YMX_OK;
    }
    if (!strcmp(name, "dma_stages")) {
        if (value != 3 && value != 4) return fail(YMX_EARG, "set_option: dma_stages must be 3 or 4");
        g_dma2_stages = (int)value;
        return YMX_OK;
    }
    if (!strcmp(name, "halo_pixels")) {
        if (value != 0 && value != 192 && value != 256 && value != 448)
            return fail(YMX_EARG, "set_option: halo_pixels must be 0, 192, 256 or 448 (both forms in one launch)");
        g_halo_pixels = (int)value;
        return YMX_OK;
    }
    if (!strcmp(name, "tile_mix")) {
        if (value != 0 && value != 1) return fail(YMX_EARG, "set_option: tile_mix must be 0 or 1");
        g_tile_mix = (int)value;
        return YMX_OK;
    }
    if (!strcmp(name, "halo_big_rounds")) {
        if (value < 0 || value > 1000000) return fail(YMX_EARG, "set_option: halo_big_rounds must be 0 (all but the last round) or a positive count");
        g_halo_big_rounds = (int)value;
        return YMX_OK;
    }
    if (!strcmp(name, "dma4_pixels")) {
        if (value != 0 && value != 192 && value != 256 && value != 448)
            return fail(YMX_EARG, "set_option: dma4_pixels must be 0, 192, 256 or 448 (both forms in one launch)");
        g_dma4_pixels = (int)value;
        return YMX_OK;
    }
    if (!strcmp(name, "wgrad_variant")) {
        if (value < 0 || value > 2) return fail(YMX_EARG, "set_option: wgrad_variant must be 0 (automatic), 1 (register-staged) or 2 (per-tap LDS-DMA)");
        g_wgrad_variant = (int)value;
        return YMX_OK;
    }
    if (!strcmp(name, "dma_narrow")) {
        if (value < 0 || value > 2) return fail(YMX_EARG, "set_option: dma_narrow must be 0, 1 or 2");
        t_opt.narrow = value;
        return YMX_OK;
    }
    return fail(YMX_EARG, "set_option: unknown option '%s'", name);
}
const char *ymx_last_error(void) { return g_err.c_str(); }

size_t ymx_packed_weight_bytes(int k, int Cin, int Cout, int dtype) {
    if (k <= 0 || Cin <= 0 || Cout <= 0) return 0;
    if (!conv_uses_mfma(Cin, dtype)) return (size_t)k * k * Cin * Cout * 4;
    return (size_t)packed_rows(Cout) * packed_kw(k, Cin) * (dtype == YMX_BF16 ? 2 : 4);
}

int ymx_pack_weights(const float *w_hwio, void *w_packed, int k, int Cin, int Cout, int dtype, void *stream) {
    if (!w_hwio || !w_packed) return fail(YMX_EARG, "pack_weights: null pointer");
    if (k <= 0 || Cin <= 0 || Cout <= 0) return fail(YMX_EARG, "pack_weights: non-positive size");
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "pack_weights: dtype %d", dtype);
    return launch_pack_weights(w_hwio, w_packed, k, Cin, Cout, dtype, (hipStream_t)stream);
}

int ymx_pack_weights_dgrad(const float *w_hwio, void *w_packed, int k, int Cin, int Cout, int dz_channels, int dtype,
                           void *stream) {
    if (!w_hwio || !w_packed) return fail(YMX_EARG, "pack_weights_dgrad: null pointer");
    if (k <= 0 || Cin <= 0 || Cout <= 0) return fail(YMX_EARG, "pack_weights_dgrad: non-positive size");
    if (dz_channels < Cout) return fail(YMX_EARG, "pack_weights_dgrad: dz_channels %d below Cout %d", dz_channels, Cout);
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "pack_weights_dgrad: dtype %d", dtype);
    return launch_pack_weights(w_hwio, w_packed, k, dz_channels, Cin, dtype, (hipStream_t)stream, Cout);
}

int ymx_pack_weights_pair(const float *w_hwio, void *w_fwd, void *w_dgrad, int k, int Cin, int Cout, int dz_channels, int stride,
                          int dtype, void *stream) {
    if (!w_hwio || !w_fwd || !w_dgrad) return fail(YMX_EARG, "pack_weights_pair: null pointer");
    if (k <= 0 || Cin <= 0 || Cout <= 0) return fail(YMX_EARG, "pack_weights_pair: non-positive size");
    if (dz_channels < Cout) return fail(YMX_EARG, "pack_weights_pair: dz_channels %d below Cout %d", dz_channels, Cout);
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "pack_weights_pair: dtype %d", dtype);
    if (stride != 1 && !(stride == 2 && k == 3 && dtype == YMX_BF16)) return fail(YMX_EARG, "pack_weights_pair: stride %d (1, or 2 for 3x3 bf16)", stride);
    return launch_pack_weights_pair(w_hwio, w_fwd, w_dgrad, k, Cin, Cout, dz_channels, stride, dtype, (hipStream_t)stream);
}

int ymx_pack_weights_dgrad_s2(const float *w_hwio, void *w_packed, int Cin, int Cout, int dz_channels, int dtype, void *stream) {
    if (!w_hwio || !w_packed) return fail(YMX_EARG, "pack_weights_dgrad_s2: null pointer");
    if (Cin <= 0 || Cin % 8 || Cout <= 0 || dz_channels < Cout) return fail(YMX_EARG, "pack_weights_dgrad_s2: Cin %d (a multiple of 8), Cout %d, dz_channels %d", Cin, Cout, dz_channels);
    if (dtype != YMX_BF16) return fail(YMX_EARG, "pack_weights_dgrad_s2: dtype %d", dtype);
    return launch_pack_weights_dgrad_s2(w_hwio, w_packed, Cin, Cout, dz_channels, dtype, (hipStream_t)stream);
}

int ymx_conv2d_dgrad_s2(const void *dz, const void *w_packed, void *dx, int B, int OH, int OW, int dz_channels, int Cin, int H, int W,
                        void *stream) {
    const int d2s[3] = {H, W, Cin};
    return conv_impl(dz, w_packed, nullptr, nullptr, nullptr, dx, B, OH, OW, dz_channels, 4 * Cin, 2, 1, YMX_ACT_NONE, YMX_BF16, YMX_BF16,
                     YMX_BF16, 0, 0, 0, (hipStream_t)stream, 0, nullptr, nullptr, d2s);
}

int ymx_conv2d_fused(const void *x, const void *w, const void *scale, const void *shift, const void *residual,
                     void *y, int B, int H, int W, int Cin, int Cout, int k, int stride, int act, int dtype,
                     void *stream) {
    return conv_impl(x, w, scale, shift, residual, y, B, H, W, Cin, Cout, k, stride, act, dtype, dtype, dtype, 0,
                     0, 0, (hipStream_t)stream);
}

int ymx_conv2d_fused_ex(const void *x, const void *w, const void *scale, const void *shift, const void *residual,
                        void *y, int B, int H, int W, int Cin, int Cout, int k, int stride, int act, int dtype,
                        int x_dtype, int out_dtype, int x_ld, int res_ld, int y_ld, void *stream) {
    if ((x_dtype != YMX_F32 && x_dtype != YMX_BF16) || (out_dtype != YMX_F32 && out_dtype != YMX_BF16))
        return fail(YMX_EARG, "conv2d: x_dtype %d / out_dtype %d", x_dtype, out_dtype);
    return conv_impl(x, w, scale, shift, residual, y, B, H, W, Cin, Cout, k, stride, act, dtype, x_dtype, out_dtype,
                     x_ld, res_ld, y_ld, (hipStream_t)stream);
}

int ymx_conv2d_fused_z(const void *x, const void *w, const void *scale, const void *shift, const void *residual, void *y,
                       void *z, int B, int H, int W, int Cin, int Cout, int k, int stride, int act, int x_dtype, void *stream) {
    if (!z) return fail(YMX_EARG, "conv2d_fused_z: null z");
    if (x_dtype != YMX_F32 && x_dtype != YMX_BF16) return fail(YMX_EARG, "conv2d_fused_z: x_dtype %d", x_dtype);
    return conv_impl(x, w, scale, shift, residual, y, B, H, W, Cin, Cout, k, stride, act, YMX_BF16, x_dtype, YMX_BF16, 0, 0, 0,
                     (hipStream_t)stream, 0, nullptr, z);
}

int ymx_upsample2x_concat(const void *skip, const void *x, void *y, int B, int H, int W, int C1, int C2, int dtype,
                          void *stream) {
    if (!x || !y) return fail(YMX_EARG, "upsample2x_concat: null pointer");       // (skip == NULL with C1 > 0: those channels of y are in place)
    if (B <= 0 || H <= 0 || W <= 0 || C1 < 0 || C2 <= 0) return fail(YMX_EARG, "upsample2x_concat: bad size");
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "upsample2x_concat: dtype %d", dtype);
    const int kch = dtype == YMX_BF16 ? 8 : 4;
    if (C1 % kch || C2 % kch) return fail(YMX_EUNSUPPORTED, "upsample2x_concat: channels must be multiples of %d", kch);
    if (!aligned16(x) || !aligned16(y) || (skip && !aligned16(skip))) return fail(YMX_EARG, "upsample2x_concat: 16-byte alignment");
    return launch_upsample2x_concat(skip, x, y, B, H, W, C1, C2, dtype, (hipStream_t)stream);
}

int ymx_conv_stem_fused(const void *x, const void *w0, const void *scale0, const void *shift0, const void *w1,
                        const void *scale1, const void *shift1, void *y, int B, int H, int W, int act, int y_ld, void *stream) {
    if (!x || !w0 || !w1 || !y) return fail(YMX_EARG, "conv_stem_fused: null pointer");
    if (B <= 0 || H < 2 || W < 2) return fail(YMX_EARG, "conv_stem_fused: B=%d H=%d W=%d", B, H, W);
    if (y_ld != 0 && (y_ld < 64 || (y_ld & 7))) return fail(YMX_EARG, "conv_stem_fused: y_ld=%d (0, or a multiple of 8 not below 64)", y_ld);
    if (!aligned16(w1) || !aligned16(y) || (((uintptr_t)x | (uintptr_t)w0) & 3))
        return fail(YMX_EARG, "conv_stem_fused: w1 and y must be 16-byte aligned, x and w0 4-byte aligned");
    return launch_conv_stem((const float *)x, (const float *)w0, (const float *)scale0, (const float *)shift0, w1,
                            (const float *)scale1, (const float *)shift1, y, B, H, W, act, y_ld, (hipStream_t)stream);
}

int ymx_conv_resblock_fused(const void *x, const void *w1, const void *scale1, const void *shift1, const void *w2,
                            const void *scale2, const void *shift2, void *y, int B, int H, int W, int C, int act, int x_ld, int y_ld,
                            void *stream) {
    if (!x || !w1 || !w2 || !y) return fail(YMX_EARG, "conv_resblock_fused: null pointer");
    if (B <= 0 || H <= 0 || W <= 0) return fail(YMX_EARG, "conv_resblock_fused: B=%d H=%d W=%d", B, H, W);
    if (C != 64 && C != 128) return fail(YMX_EUNSUPPORTED, "conv_resblock_fused: %d channels (64 or 128)", C);
    if ((x_ld != 0 && (x_ld < C || (x_ld & 7))) || (y_ld != 0 && (y_ld < C || (y_ld & 7))))
        return fail(YMX_EARG, "conv_resblock_fused: x_ld=%d, y_ld=%d (each 0, or a multiple of 8 not below %d)", x_ld, y_ld, C);
    if (!aligned16(x) || !aligned16(w1) || !aligned16(w2) || !aligned16(y))
        return fail(YMX_EARG, "conv_resblock_fused: x, w1, w2 and y must be 16-byte aligned");
    if (x == y) return fail(YMX_EARG, "conv_resblock_fused: y must not be x (neighbouring tiles read x after this one has written y)");
    return (C == 64 ? launch_conv_resblock : launch_conv_resblock128)(
        x, x_ld, w1, (const float *)scale1, (const float *)shift1, w2, (const float *)scale2, (const float *)shift2, y, y_ld, B, H, W, act,
        (hipStream_t)stream);
}

// Op i is the 1x1 convolution C -> C/2 of a ResidialBlock (C = 64 or 128), marked as read by op i + 1 alone, and
// op i + 1 the 3x3 convolution C/2 -> C that reads it and adds op i's input: the block the fused kernels take.
static bool resblock_pair(const ymx_op &o, const ymx_op &n) {
    const int C = o.Cin;
    return o.kind == 1 && n.kind == 1 && o.C1 != 0 && (C == 64 || C == 128) &&
           o.Cout == C / 2 && o.k == 1 && o.stride == 1 && o.dtype == YMX_BF16 && o.x_dtype == YMX_BF16 &&
           o.out_dtype == YMX_BF16 && !o.residual && (o.y_ld == 0 || o.y_ld == C / 2) &&
           n.Cin == C / 2 && n.Cout == C && n.k == 3 && n.stride == 1 && n.dtype == YMX_BF16 && n.x_dtype == YMX_BF16 &&
           n.out_dtype == YMX_BF16 && n.residual == o.x && n.x == o.y && n.y != o.x && n.B == o.B && n.H == o.H && n.W == o.W &&
           (n.x_ld == 0 || n.x_ld == C / 2) && (n.res_ld == 0 ? (o.x_ld == 0 || o.x_ld == C) : n.res_ld == (o.x_ld ? o.x_ld : C)) &&
           o.act == n.act && conv_resblock_supported(o.act);
}

// Op i is the network's first layer on an fp32 image, marked as read by op i + 1 alone, and op
// i + 1 is the 3x3 stride-2 layer to 64 channels that reads it: the pair the fused kernel takes.
static bool stem_pair(const ymx_op &o, const ymx_op &n) {
    return o.kind == 1 && n.kind == 1 && o.C1 != 0 &&
           o.Cin == 3 && o.Cout == 32 && o.k == 3 && o.stride == 1 && o.dtype == YMX_BF16 && o.x_dtype == YMX_F32 &&
           o.out_dtype == YMX_BF16 && !o.residual && (o.x_ld == 0 || o.x_ld == 3) && (o.y_ld == 0 || o.y_ld == 32) &&
           n.Cin == 32 && n.Cout == 64 && n.k == 3 && n.stride == 2 && n.dtype == YMX_BF16 && n.x_dtype == YMX_BF16 &&
           n.out_dtype == YMX_BF16 && !n.residual && n.x == o.y && n.B == o.B && n.H == o.H && n.W == o.W &&
           (n.x_ld == 0 || n.x_ld == 32) && o.act == n.act && conv_stem_supported(o.act) && o.H >= 2 && o.W >= 2;
}

int ymx_run_program(const ymx_op *ops, int n_ops, void *stream) {
    if (!ops || n_ops < 0) return fail(YMX_EARG, "run_program: bad program");
    for (int i = 0; i < n_ops; ++i) {
        const ymx_op &o = ops[i];
        int rc = YMX_EUNSUPPORTED;
        if (i + 1 < n_ops && stem_pair(o, ops[i + 1])) {
            const ymx_op &n = ops[i + 1];
            rc = ymx_conv_stem_fused(o.x, o.w, o.scale, o.shift, n.w, n.scale, n.shift, n.y, o.B, o.H, o.W, o.act, n.y_ld, stream);
            if (rc == YMX_OK) { ++i; continue; }
        }
        if (rc == YMX_EUNSUPPORTED && i + 1 < n_ops && resblock_pair(o, ops[i + 1])) {
            const ymx_op &n = ops[i + 1];
            rc = ymx_conv_resblock_fused(o.x, o.w, o.scale, o.shift, n.w, n.scale, n.shift, n.y, o.B, o.H, o.W, o.Cin, o.act, o.x_ld, n.y_ld, stream);
            if (rc == YMX_OK) { ++i; continue; }
        }
        if (rc != YMX_EUNSUPPORTED) {
            // the fused launch failed for a reason the two separate launches would share
        } else if (o.kind == 1)
            rc = conv_impl(o.x, o.w, o.scale, o.shift, o.residual, o.y, o.B, o.H, o.W, o.Cin, o.Cout, o.k, o.stride,
                           o.act, o.dtype, o.x_dtype, o.out_dtype, o.x_ld, o.res_ld, o.y_ld, (hipStream_t)stream, o.min_tiles);
        else if (o.kind == 2)
            rc = ymx_upsample2x_concat(o.residual, o.x, o.y, o.B, o.H, o.W, o.C1, o.C2, o.dtype, stream);
        else
            rc = fail(YMX_EARG, "run_program: op %d has unknown kind %d", i, o.kind);
        if (rc != YMX_OK) {
            const std::string inner = g_err;
            return fail(rc, "run_program: op %d: %s", i, inner.c_str());
        }
    }
    return YMX_OK;
}

const char *ymx_conv_kernel_label(const ymx_op *op) {
    if (!op || op->kind != 1) { fail(YMX_EARG, "conv_kernel_label: not a convolution op"); return nullptr; }
    const ymx_op &o = *op;
    const char *label = nullptr;
    const int rc = conv_impl(o.x, o.w, o.scale, o.shift, o.residual, o.y, o.B, o.H, o.W, o.Cin, o.Cout, o.k, o.stride,
                             o.act, o.dtype, o.x_dtype, o.out_dtype, o.x_ld, o.res_ld, o.y_ld, nullptr, o.min_tiles, &label);
    return rc == YMX_OK ? label : nullptr;
}

// workspace layout of the decode/filter calls: [err:4 int][ticket: B int, padded to 16 B][state: B*nseg u64]
static size_t decode_ws_bytes(int B, int nseg) {
    const size_t head = ((size_t)(4 + B) * 4 + 15) & ~(size_t)15;
    return head + (size_t)B * nseg * 8;
}
static int count_segments(int n_scales, const int *S) {
    int n = 0;
    for (int i = 0; i < n_scales; ++i) n += (3 * S[i] * S[i] + DEC_SEG - 1) / DEC_SEG;
    return n;
}

size_t ymx_decode_workspace_bytes(int B, int n_scales, const int *S) {
    if (B <= 0 || n_scales <= 0 || n_scales > 3 || !S) return 0;
    return decode_ws_bytes(B, count_segments(n_scales, S));
}

static int setup_ws(DecodeArgs &a, void *workspace, size_t ws_bytes, hipStream_t s) {
    const size_t need = decode_ws_bytes(a.B, a.nseg);
    if (!workspace || ws_bytes < need) return fail(YMX_EWORKSPACE, "decode: workspace %zu < %zu bytes", ws_bytes, need);
    if (!aligned16(workspace)) return fail(YMX_EARG, "decode: workspace must be 16-byte aligned");
    const size_t head = ((size_t)(4 + a.B) * 4 + 15) & ~(size_t)15;
    a.err = (int *)workspace;
    a.ticket = (int *)workspace + 4;
    a.state = (unsigned long long *)((char *)workspace + head);
    if (hipMemsetAsync(workspace, 0, need, s) != hipSuccess) {
        (void)hipGetLastError();
        return fail(YMX_ELAUNCH, "decode: hipMemsetAsync failed");
    }
    return YMX_OK;
}

int ymx_decode_filter(const void *const *head, int n_scales, const int *S, int B, int C, int head_dtype,
                      const float *anchors, const float *strides, const float *xyscale, int filter, float thr,
                      float image_size, float *boxes, float *scores, int *src_idx, int *counts, int cap,
                      void *workspace, size_t ws_bytes, void *stream) {
    if (!head || !S || !anchors || !strides || !xyscale || !boxes || !scores || !counts)
        return fail(YMX_EARG, "decode_filter: null pointer");
    if (n_scales < 1 || n_scales > 3) return fail(YMX_EUNSUPPORTED, "decode_filter: 1..3 scales, got %d", n_scales);
    if (B <= 0 || C <= 0 || cap <= 0 || (cap & 3)) return fail(YMX_EARG, "decode_filter: bad B/C/cap (cap %% 4 == 0)");
    if (filter && !src_idx) return fail(YMX_EARG, "decode_filter: src_idx required when filtering");
    if (!aligned16(boxes) || !aligned16(scores)) return fail(YMX_EARG, "decode_filter: boxes/scores must be 16-byte aligned");
    DecodeArgs a;
    memset(&a, 0, sizeof(a));
    int nseg = 0, ncand = 0;
    for (int i = 0; i < n_scales; ++i) {
        if (!head[i] || S[i] <= 0) return fail(YMX_EARG, "decode_filter: scale %d missing", i);
        a.head[i] = head[i]; a.S[i] = S[i];
        a.seg_begin[i] = nseg; a.cand_begin[i] = ncand;
        nseg += (3 * S[i] * S[i] + DEC_SEG - 1) / DEC_SEG; ncand += 3 * S[i] * S[i];
        a.stride[i] = strides[i]; a.xyscale[i] = xyscale[i];
        for (int j = 0; j < 6; ++j) (&a.anchor[i][0][0])[j] = anchors[i * 6 + j];
    }
    a.seg_begin[n_scales] = nseg; a.cand_begin[n_scales] = ncand;
    a.n_scales = n_scales; a.B = B; a.C = C; a.N = ncand; a.cap = cap; a.head_dtype = head_dtype;
    a.filter = filter ? 1 : 0; a.thr = thr; a.image_size = image_size;
    a.boxes = boxes; a.scores = scores; a.src_idx = src_idx; a.counts = counts; a.nseg = nseg;
    a.conf = filter ? nullptr : g_split_conf;
    if (g_cell_ld) {
        for (int i = 0; i < n_scales; ++i) {
            const int pad = g_cell_ld[i] - 3 * (5 + C);
            if (pad < 0) return fail(YMX_EARG, "decode_filter: cell stride %d below 3*(5+C) = %d", g_cell_ld[i], 3 * (5 + C));
            if (pad > 4096) return fail(YMX_EUNSUPPORTED, "decode_filter: cell stride %d too large", g_cell_ld[i]);
            a.cell_ld[i] = g_cell_ld[i];
            if (pad > a.max_pad) a.max_pad = pad;
        }
    }
    for (int i = 0; i < n_scales; ++i) {
        const int ld = a.cell_ld[i] ? a.cell_ld[i] : 3 * (5 + C);
        const unsigned long long bytes = ((unsigned long long)B * S[i] * S[i] - 1) * ld * 4 + 3ull * (5 + C) * 4;
        a.head_bytes[i] = (head_dtype == YMX_F32 && bytes < 0xffffff00ull) ? (unsigned)bytes : 0u;
    }
    if (!filter && cap < ncand) return fail(YMX_EARG, "decode_filter: cap %d < %d candidates", cap, ncand);
    if (filter) {
        const int rc = setup_ws(a, workspace, ws_bytes, (hipStream_t)stream);
        if (rc) return rc;
    }
    return launch_decode(a, (hipStream_t)stream);
}

int ymx_decode_filter_strided(const void *const *head, const int *cell_ld, int n_scales, const int *S, int B, int C,
                              int head_dtype, const float *anchors, const float *strides, const float *xyscale,
                              int filter, float thr, float image_size, float *boxes, float *scores, int *src_idx,
                              int *counts, int cap, void *workspace, size_t ws_bytes, void *stream) {
    if (!cell_ld) return fail(YMX_EARG, "decode_filter_strided: null cell_ld");
    g_cell_ld = cell_ld;
    const int rc = ymx_decode_filter(head, n_scales, S, B, C, head_dtype, anchors, strides, xyscale, filter, thr,
                                     image_size, boxes, scores, src_idx, counts, cap, workspace, ws_bytes, stream);
    g_cell_ld = nullptr;
    return rc;
}

int ymx_decode_split(const void *const *head, int n_scales, const int *S, int B, int C, int head_dtype,
                     const float *anchors, const float *strides, const float *xyscale, float *boxes, float *conf,
                     float *prob, int *counts, int cap, void *stream) {
    if (!conf) return fail(YMX_EARG, "decode_split: null conf");
    g_split_conf = conf;
    const int rc = ymx_decode_filter(head, n_scales, S, B, C, head_dtype, anchors, strides, xyscale, 0, 0.0f, 1.0f, boxes,
                                     prob, nullptr, counts, cap, nullptr, 0, stream);
    g_split_conf = nullptr;
    return rc;
}

size_t ymx_filter_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return decode_ws_bytes(B, (N + DEC_SEG - 1) / DEC_SEG);
}

int ymx_filter_boxes(const float *xywh, const float *scores, long long sb, long long sn, long long sc, int B, int N,
                     int C, float thr, float image_size, float *boxes, float *out_scores, int *src_idx, int *counts,
                     int cap, void *workspace, size_t ws_bytes, void *stream) {
    if (!xywh || !scores || !boxes || !out_scores || !src_idx || !counts) return fail(YMX_EARG, "filter_boxes: null pointer");
    if (B <= 0 || N <= 0 || C <= 0 || cap <= 0 || (cap & 3)) return fail(YMX_EARG, "filter_boxes: bad B/N/C/cap");
    if (sb < 0 || sn < 0 || sc < 0) return fail(YMX_EUNSUPPORTED, "filter_boxes: negative strides");
    if (!aligned16(xywh) || !aligned16(boxes) || !aligned16(out_scores)) return fail(YMX_EARG, "filter_boxes: 16-byte alignment");
    DecodeArgs a;
    memset(&a, 0, sizeof(a));
    a.n_scales = 1; a.S[0] = 1; a.seg_begin[0] = 0; a.cand_begin[0] = 0;
    a.nseg = (N + DEC_SEG - 1) / DEC_SEG; a.seg_begin[1] = a.nseg; a.cand_begin[1] = N;
    a.B = B; a.C = C; a.N = N; a.cap = cap; a.filter = 2; a.thr = thr; a.image_size = image_size;
    a.in_xywh = xywh; a.in_scores = scores; a.sb = sb; a.sn = sn; a.sc = sc;
    a.boxes = boxes; a.scores = out_scores; a.src_idx = src_idx; a.counts = counts;
    const int rc = setup_ws(a, workspace, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
    return launch_decode(a, (hipStream_t)stream);
}

size_t ymx_nms_workspace_bytes(int B, int C, int per_class) { return nms_workspace_bytes(B, C, per_class); }

int ymx_nms_combined(const float *boxes, const float *scores, const int *counts, const int *src_idx, int B, int cap,
                     int C, int per_class, int total, float iou_thr, float score_thr, float *out_boxes,
                     float *out_scores, float *out_classes, int *out_valid, int *out_keep_idx, void *workspace,
                     size_t ws_bytes, void *stream) {
    return nms_entry(boxes, scores, counts, src_idx, B, cap, C, per_class, total, iou_thr, score_thr, out_boxes,
                     out_scores, out_classes, out_valid, out_keep_idx, workspace, ws_bytes, (hipStream_t)stream);
}

int ymx_spp_concat(const void *x, void *y, int B, int H, int W, int C, int dtype, void *stream) {
    if (!x || !y) return fail(YMX_EARG, "spp_concat: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(YMX_EARG, "spp_concat: non-positive size");
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "spp_concat: dtype %d", dtype);
    if (C % (dtype == YMX_BF16 ? 8 : 4)) return fail(YMX_EUNSUPPORTED, "spp_concat: channels must be a multiple of %d", dtype == YMX_BF16 ? 8 : 4);
    if (B > 65535) return fail(YMX_EUNSUPPORTED, "spp_concat: batch above 65535");
    if (!aligned16(x) || !aligned16(y)) return fail(YMX_EARG, "spp_concat: 16-byte alignment");
    return launch_spp_concat(x, y, B, H, W, C, dtype, (hipStream_t)stream);
}

int ymx_spp_concat_train(const void *x, void *y, unsigned char *argmax, int B, int H, int W, int C, int dtype, void *stream) {
    if (!x || !y || !argmax) return fail(YMX_EARG, "spp_concat_train: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(YMX_EARG, "spp_concat_train: non-positive size");
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "spp_concat_train: dtype %d", dtype);
    return launch_spp_train(x, y, argmax, B, H, W, C, dtype, (hipStream_t)stream);
}

int ymx_spp_concat_backward(const void *dy, const unsigned char *argmax, void *dx, int B, int H, int W, int C, int dtype, void *stream) {
    if (!dy || !dx || !argmax) return fail(YMX_EARG, "spp_concat_backward: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(YMX_EARG, "spp_concat_backward: non-positive size");
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "spp_concat_backward: dtype %d", dtype);
    return launch_spp_bwd(dy, argmax, dx, B, H, W, C, dtype, (hipStream_t)stream);
}

int ymx_act_apply(const void *z, const void *residual, void *y, long long n, int act, int dtype, void *stream) {
    if (!z || !y) return fail(YMX_EARG, "act_apply: null pointer");
    if (n <= 0) return fail(YMX_EARG, "act_apply: non-positive size");
    if (act < 0 || act > 2) return fail(YMX_EARG, "act_apply: activation %d", act);
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "act_apply: dtype %d", dtype);
    return launch_act(z, residual, y, (size_t)n, act, dtype, 0, (hipStream_t)stream);
}

int ymx_act_backward(const void *z, const void *dy, void *dz, long long n, int act, int dtype, void *stream) {
    if (!z || !dy || !dz) return fail(YMX_EARG, "act_backward: null pointer");
    if (n <= 0) return fail(YMX_EARG, "act_backward: non-positive size");
    if (act < 0 || act > 2) return fail(YMX_EARG, "act_backward: activation %d", act);
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "act_backward: dtype %d", dtype);
    return launch_act(z, dy, dz, (size_t)n, act, dtype, 1, (hipStream_t)stream);
}

size_t ymx_act_backward_bias_workspace_bytes(long long n_pix, int C, int act, int dtype) {
    return act_bwd_bias_supported(n_pix, C, act, dtype) ? act_bwd_bias_workspace_bytes(n_pix, C) : 0;
}

int ymx_act_backward_bias(const void *z, const void *dy, void *dz, long long n_pix, int C, int act, int dtype, float *db,
                          void *workspace, size_t ws_bytes, void *stream) {
    if (!z || !dy || !dz || !db) return fail(YMX_EARG, "act_backward_bias: null pointer");
    if (!act_bwd_bias_supported(n_pix, C, act, dtype))
        return fail(YMX_EUNSUPPORTED, "act_backward_bias: bf16, leaky or mish, and a power-of-two number (<= 256) of 8-channel chunks per row are required (C = %d)", C);
    if (!aligned16(z) || !aligned16(dy) || !aligned16(dz)) return fail(YMX_EARG, "act_backward_bias: 16-byte alignment");
    const size_t need = act_bwd_bias_workspace_bytes(n_pix, C);
    if (!workspace || ws_bytes < need || (((uintptr_t)workspace) & 3)) return fail(YMX_EWORKSPACE, "act_backward_bias: workspace %zu < %zu bytes", ws_bytes, need);
    return launch_act_bwd_bias(z, dy, dz, n_pix, C, act, db, workspace, (hipStream_t)stream);
}

size_t ymx_bias_grad_workspace_bytes(long long n_pix, int C) { return (n_pix > 0 && C > 0) ? bias_grad_workspace_bytes(n_pix, C) : 0; }

int ymx_bias_grad(const void *dz, long long n_pix, int C, int ld, int dtype, float *db, void *workspace, size_t ws_bytes, void *stream) {
    if (!dz || !db) return fail(YMX_EARG, "bias_grad: null pointer");
    if (n_pix <= 0 || C <= 0) return fail(YMX_EARG, "bias_grad: non-positive size");
    if (ld != 0 && ld < C) return fail(YMX_EARG, "bias_grad: row stride below the channel count");
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "bias_grad: dtype %d", dtype);
    const size_t need = bias_grad_workspace_bytes(n_pix, C);
    if (!workspace || ws_bytes < need) return fail(YMX_EWORKSPACE, "bias_grad: workspace %zu < %zu bytes", ws_bytes, need);
    return launch_bias_grad(dz, n_pix, C, ld ? ld : C, dtype, db, workspace, (hipStream_t)stream);
}

static long long wgrad_pixels(int B, int H, int W, int k, int stride) {
    const int oh = stride == 1 ? H : (H + 1 - k) / 2 + 1, ow = stride == 1 ? W : (W + 1 - k) / 2 + 1;
    return (long long)B * oh * ow;
}

static size_t wgrad_ws_need(int B, int H, int W, int Cin, int Cout, int k, int stride) {
    size_t need = conv_wgrad_workspace_bytes(k, Cin, Cout, wgrad_pixels(B, H, W, k, stride));
    if (k == 3 && stride == 2 && Cin % 8 == 0 && Cout % 8 == 0 && Cin <= 32) {
        const size_t s2 = conv_wgrad_s2_workspace_bytes(B, H, W, Cin, Cout);
        need = s2 > need ? s2 : need;
    }
    return need;
}

size_t ymx_conv_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int k, int stride) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (k != 1 && k != 3) || (stride != 1 && stride != 2)) return 0;
    return wgrad_ws_need(B, H, W, Cin, Cout, k, stride);
}

int ymx_conv_wgrad(const void *x, int x_ld, const void *dz, int dz_ld, float *dw, int B, int H, int W, int Cin, int Cout, int k,
                   int stride, void *workspace, size_t ws_bytes, void *stream) {
    if (!x || !dz || !dw) return fail(YMX_EARG, "conv_wgrad: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return fail(YMX_EARG, "conv_wgrad: non-positive size");
    if ((k != 1 && k != 3) || (stride != 1 && stride != 2)) return fail(YMX_EUNSUPPORTED, "conv_wgrad: k=%d stride=%d", k, stride);
    if (stride == 2 && ((H + 1 - k) < 0 || (W + 1 - k) < 0)) return fail(YMX_EARG, "conv_wgrad: input smaller than the kernel");
    if (Cin % 8 || Cout % 8) return fail(YMX_EUNSUPPORTED, "conv_wgrad: channel counts must be multiples of 8 (pad them), got %d -> %d", Cin, Cout);
    if ((x_ld != 0 && (x_ld < Cin || x_ld % 8)) || (dz_ld != 0 && (dz_ld < Cout || dz_ld % 8))) return fail(YMX_EARG, "conv_wgrad: pixel strides must be multiples of 8 elements");
    if (!aligned16(x) || !aligned16(dz) || (((uintptr_t)dw) & 3)) return fail(YMX_EARG, "conv_wgrad: alignment");
    if ((long long)B * H * W * (x_ld ? x_ld : Cin) >= (1LL << 40)) return fail(YMX_EUNSUPPORTED, "conv_wgrad: input too large");
    const size_t need = wgrad_ws_need(B, H, W, Cin, Cout, k, stride);
    if (need && (!workspace || ws_bytes < need)) return fail(YMX_EWORKSPACE, "conv_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    if (need && (((uintptr_t)workspace) & 3)) return fail(YMX_EARG, "conv_wgrad: workspace must be 4-byte aligned");
    return launch_conv_wgrad(x, x_ld, dz, dz_ld, dw, B, H, W, Cin, Cout, k, stride, workspace, (hipStream_t)stream);
}

size_t ymx_conv_wgrad_first_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    return conv_wgrad_first_workspace_bytes(B, H, W, Cin, Cout);
}

int ymx_conv_wgrad_first(const void *x, int x_dtype, const void *dz, int dz_ld, float *dw, int B, int H, int W, int Cin, int Cout,
                         void *workspace, size_t ws_bytes, void *stream) {
    if (!x || !dz || !dw || !workspace) return fail(YMX_EARG, "conv_wgrad_first: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return fail(YMX_EARG, "conv_wgrad_first: non-positive size");
    if (x_dtype != YMX_F32 && x_dtype != YMX_BF16) return fail(YMX_EARG, "conv_wgrad_first: x_dtype %d", x_dtype);
    if (!conv_wgrad_first_supported(B, H, W, Cin, Cout))
        return fail(YMX_EUNSUPPORTED, "conv_wgrad_first: %d -> %d channels at width %d (at most 3 input channels, 64 output channels in multiples of 8)", Cin, Cout, W);
    if (dz_ld != 0 && (dz_ld < Cout || dz_ld % 8)) return fail(YMX_EARG, "conv_wgrad_first: dz pixel stride must be a multiple of 8 elements");
    if (!aligned16(dz) || (((uintptr_t)dw) & 3) || (((uintptr_t)workspace) & 3)) return fail(YMX_EARG, "conv_wgrad_first: alignment");
    const size_t need = conv_wgrad_first_workspace_bytes(B, H, W, Cin, Cout);
    if (ws_bytes < need) return fail(YMX_EWORKSPACE, "conv_wgrad_first: workspace %zu < %zu bytes", ws_bytes, need);
    return launch_conv_wgrad_first(x, x_dtype == YMX_F32 ? 1 : 0, dz, dz_ld, dw, B, H, W, Cin, Cout, workspace, (hipStream_t)stream);
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), eight bytes per step: host code, for
// the record framing of yolo_amd/tfrecord.py (records hold megabytes of tensor data).
unsigned ymx_crc32c(const void *data, size_t n, unsigned crc) {
    static unsigned table[8][256];
    static std::atomic<int> ready{0};
    if (!ready.load(std::memory_order_acquire)) {
        unsigned t[8][256];
        for (unsigned i = 0; i < 256; ++i) {
            unsigned c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (unsigned i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = t[0][t[k - 1][i] & 0xffu] ^ (t[k - 1][i] >> 8);
        memcpy(table, t, sizeof(t));             // every thread writes the same values
        ready.store(1, std::memory_order_release);
    }
    const unsigned char *p = (const unsigned char *)data;
    unsigned c = ~crc;
    if (!p) return crc;
    while (n >= 8) {
        unsigned lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = table[7][lo & 0xffu] ^ table[6][(lo >> 8) & 0xffu] ^ table[5][(lo >> 16) & 0xffu] ^ table[4][lo >> 24] ^
            table[3][hi & 0xffu] ^ table[2][(hi >> 8) & 0xffu] ^ table[1][(hi >> 16) & 0xffu] ^ table[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = table[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
    return ~c;
}

const char *ymx_conv_wgrad_kernel_label(int B, int H, int W, int Cin, int Cout, int k, int stride) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (k != 1 && k != 3) || (stride != 1 && stride != 2)) {
        fail(YMX_EARG, "conv_wgrad_kernel_label: bad shape");
        return nullptr;
    }
    return conv_wgrad_label(B, H, W, Cin, Cout, k, stride);
}

int ymx_upsample2x_backward(const void *dy, void *dx, int B, int H, int W, int C, int dy_ld, int c_off, int dtype, void *stream) {
    if (!dy || !dx) return fail(YMX_EARG, "upsample2x_backward: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || c_off < 0) return fail(YMX_EARG, "upsample2x_backward: bad size");
    if (dy_ld != 0 && dy_ld < c_off + C) return fail(YMX_EARG, "upsample2x_backward: row stride below the channels read");
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "upsample2x_backward: dtype %d", dtype);
    return launch_upsample2x_bwd(dy, dx, B, H, W, C, dy_ld ? dy_ld : c_off + C, c_off, dtype, (hipStream_t)stream);
}

int ymx_yolo_loss_grad(const float *conv, int conv_ld, const float *label, long long label_stride, const float *bboxes,
                       long long bbox_stride, int B, int S, int C, int n_boxes, float stride, float xyscale, const float *anchors,
                       float iou_loss_threshold, const float *term_weights, float *dconv, void *stream) {
    if (!conv || !label || !bboxes || !anchors || !dconv) return fail(YMX_EARG, "yolo_loss_grad: null pointer");
    if (B <= 0 || S <= 0 || C <= 0 || n_boxes <= 0) return fail(YMX_EARG, "yolo_loss_grad: non-positive size");
    if (B > 65535) return fail(YMX_EUNSUPPORTED, "yolo_loss_grad: batch above 65535");
    if (conv_ld != 0 && conv_ld < 3 * (5 + C)) return fail(YMX_EARG, "yolo_loss_grad: cell stride %d below 3*(5+C) = %d", conv_ld, 3 * (5 + C));
    if (label_stride != 0 && label_stride < (long long)3 * S * S * (5 + C)) return fail(YMX_EARG, "yolo_loss_grad: label image stride below one image");
    if (bbox_stride != 0 && bbox_stride < (long long)4 * n_boxes) return fail(YMX_EARG, "yolo_loss_grad: box image stride below one image");
    if ((((uintptr_t)conv) | ((uintptr_t)label) | ((uintptr_t)bboxes) | ((uintptr_t)dconv)) & 3) return fail(YMX_EARG, "yolo_loss_grad: pointers must be 4-byte aligned");
    return launch_yolo_loss_grad(conv, conv_ld, label, label_stride, bboxes, bbox_stride, B, S, C, n_boxes, stride, xyscale, anchors,
                                 iou_loss_threshold, term_weights, dconv, (hipStream_t)stream);
}

int ymx_letterbox(const void *images, int in_dtype, int B, int H, int W, void *out, int out_dtype, int target, int literal,
                  void *stream) {
    if (!images || !out) return fail(YMX_EARG, "letterbox: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || target <= 0) return fail(YMX_EARG, "letterbox: non-positive size");
    if (in_dtype != YMX_F32 && in_dtype != YMX_U8) return fail(YMX_EARG, "letterbox: source dtype %d (fp32 or 8-bit)", in_dtype);
    if (out_dtype != YMX_F32 && out_dtype != YMX_BF16) return fail(YMX_EARG, "letterbox: output dtype %d", out_dtype);
    if ((size_t)B * H * W >= ((size_t)1 << 40) || (size_t)B * target * target >= ((size_t)1 << 40)) return fail(YMX_EUNSUPPORTED, "letterbox: too many pixels");
    return launch_letterbox(images, in_dtype == YMX_U8, B, H, W, out, out_dtype == YMX_BF16, target, literal != 0, (hipStream_t)stream);
}

int ymx_draw_boxes(unsigned char *images, int B, int H, int W, const float *boxes, const float *classes, const int *valid,
                   int max_boxes, const unsigned char *colors, int C, int thickness, void *stream) {
    if (!images || !boxes || !classes || !valid || !colors) return fail(YMX_EARG, "draw_boxes: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || max_boxes <= 0 || C <= 0) return fail(YMX_EARG, "draw_boxes: non-positive size");
    return launch_draw_boxes(images, B, H, W, boxes, classes, valid, max_boxes, colors, C, thickness, nullptr, 0, 1,
                             (hipStream_t)stream);
}

int ymx_draw_boxes_labeled(unsigned char *images, int B, int H, int W, const float *boxes, const float *classes,
                           const int *valid, int max_boxes, const unsigned char *colors, int C, int thickness,
                           const unsigned char *labels, int label_bytes, int text_scale, void *stream) {
    if (!images || !boxes || !classes || !valid || !colors || !labels) return fail(YMX_EARG, "draw_boxes_labeled: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || max_boxes <= 0 || C <= 0) return fail(YMX_EARG, "draw_boxes_labeled: non-positive size");
    if (label_bytes < 1 || label_bytes > 256) return fail(YMX_EARG, "draw_boxes_labeled: label_bytes=%d (1..256)", label_bytes);
    if (text_scale < 1 || text_scale > 16) return fail(YMX_EARG, "draw_boxes_labeled: text_scale=%d (1..16)", text_scale);
    return launch_draw_boxes(images, B, H, W, boxes, classes, valid, max_boxes, colors, C, thickness, labels, label_bytes,
                             text_scale, (hipStream_t)stream);
}

size_t ymx_yolo_loss_workspace_bytes(int B, int S) { return yolo_loss_workspace_bytes(B, S); }

int ymx_yolo_loss(const float *conv, int conv_ld, const float *pred, const float *label, long long label_stride,
                  const float *bboxes, long long bbox_stride, int B, int S, int C, int n_boxes, float stride, float xyscale, const float *anchors, float iou_loss_threshold,
                  float *out, float *per_image, void *workspace, size_t ws_bytes, void *stream) {
    if (!conv || !label || !bboxes || !anchors || !out) return fail(YMX_EARG, "yolo_loss: null pointer");
    if (B <= 0 || S <= 0 || C <= 0 || n_boxes <= 0) return fail(YMX_EARG, "yolo_loss: non-positive size");
    if (B > 65535) return fail(YMX_EUNSUPPORTED, "yolo_loss: batch above 65535");
    if (conv_ld != 0 && conv_ld < 3 * (5 + C)) return fail(YMX_EARG, "yolo_loss: cell stride %d below 3*(5+C) = %d", conv_ld, 3 * (5 + C));
    if (label_stride != 0 && label_stride < (long long)3 * S * S * (5 + C)) return fail(YMX_EARG, "yolo_loss: label image stride below one image");
    if (bbox_stride != 0 && bbox_stride < (long long)4 * n_boxes) return fail(YMX_EARG, "yolo_loss: box image stride below one image");
    if ((((uintptr_t)conv) | ((uintptr_t)label) | ((uintptr_t)bboxes) | ((uintptr_t)pred)) & 3) return fail(YMX_EARG, "yolo_loss: pointers must be 4-byte aligned");
    const size_t need = yolo_loss_workspace_bytes(B, S);
    if (!workspace || ws_bytes < need) return fail(YMX_EWORKSPACE, "yolo_loss: workspace %zu < %zu bytes", ws_bytes, need);
    if (((uintptr_t)workspace) & 3) return fail(YMX_EARG, "yolo_loss: workspace must be 4-byte aligned");
    return launch_yolo_loss(conv, conv_ld, pred, label, label_stride, bboxes, bbox_stride, B, S, C, n_boxes, stride, xyscale, anchors, iou_loss_threshold,
                            out, per_image, workspace, (hipStream_t)stream);
}

int ymx_affine_act(const void *x, const float *scale, const float *shift, void *y, long long n_pix, int C, int act,
                   int dtype, void *stream) {
    if (!x || !y) return fail(YMX_EARG, "affine_act: null pointer");
    if (n_pix < 0 || C <= 0) return fail(YMX_EARG, "affine_act: bad size");
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "affine_act: dtype %d", dtype);
    if (act != YMX_ACT_NONE && act != YMX_ACT_LEAKY && act != YMX_ACT_MISH) return fail(YMX_EARG, "affine_act: activation code %d", act);
    if (n_pix == 0) return YMX_OK;
    return launch_affine_act(x, scale, shift, y, n_pix * C, C, act, dtype, (hipStream_t)stream);
}

static int bn_args_ok(const char *what, long long n_pix, int C, int dtype) {
    if (n_pix <= 0 || C <= 0) return fail(YMX_EARG, "%s: bad size", what);
    if (dtype != YMX_F32 && dtype != YMX_BF16) return fail(YMX_EARG, "%s: dtype %d", what, dtype);
    return YMX_OK;
}
size_t ymx_bn_workspace_bytes(long long n_pix, int C) { return (n_pix > 0 && C > 0) ? bn_workspace_bytes(n_pix, C) : 0; }

int ymx_bn_stats(const void *u, long long n_pix, int C, int dtype, float *sums, void *workspace, size_t ws_bytes, void *stream) {
    if (!u || !sums || !workspace) return fail(YMX_EARG, "bn_stats: null pointer");
    if (const int rc = bn_args_ok("bn_stats", n_pix, C, dtype)) return rc;
    if (ws_bytes < bn_workspace_bytes(n_pix, C)) return fail(YMX_EARG, "bn_stats: workspace of %zu bytes, %zu needed", ws_bytes, bn_workspace_bytes(n_pix, C));
    return launch_bn_reduce(u, nullptr, nullptr, nullptr, nullptr, n_pix, C, 0, dtype, sums, workspace, (hipStream_t)stream);
}

int ymx_bn_act_forward(const void *u, const float *scale, const float *shift, const void *residual, void *y, long long n_pix, int C,
                       int act, int dtype, void *stream) {
    if (!u || !y || !scale || !shift) return fail(YMX_EARG, "bn_act_forward: null pointer");
    if (const int rc = bn_args_ok("bn_act_forward", n_pix, C, dtype)) return rc;
    if (act != YMX_ACT_NONE && act != YMX_ACT_LEAKY && act != YMX_ACT_MISH) return fail(YMX_EARG, "bn_act_forward: activation code %d", act);
    return launch_affine_act(u, scale, shift, y, n_pix * C, C, act, dtype, (hipStream_t)stream, residual);
}

int ymx_bn_act_backward(const void *u, const void *dy, const float *scale, const float *shift, void *dz, long long n_pix, int C, int act,
                        int dtype, float *sums, void *workspace, size_t ws_bytes, void *stream) {
    if (!u || !dy || !dz || !scale || !shift || !sums || !workspace) return fail(YMX_EARG, "bn_act_backward: null pointer");
    if (const int rc = bn_args_ok("bn_act_backward", n_pix, C, dtype)) return rc;
    if (act != YMX_ACT_NONE && act != YMX_ACT_LEAKY && act != YMX_ACT_MISH) return fail(YMX_EARG, "bn_act_backward: activation code %d", act);
    if (ws_bytes < bn_workspace_bytes(n_pix, C)) return fail(YMX_EARG, "bn_act_backward: workspace of %zu bytes, %zu needed", ws_bytes, bn_workspace_bytes(n_pix, C));
    return launch_bn_reduce(u, dy, dz, scale, shift, n_pix, C, act, dtype, sums, workspace, (hipStream_t)stream);
}

int ymx_bn_backward_apply(const void *dz, const void *u, const float *a, const float *b, const float *c, void *du, long long n_pix, int C,
                          int dtype, void *stream) {
    if (!dz || !u || !a || !b || !c || !du) return fail(YMX_EARG, "bn_backward_apply: null pointer");
    if (const int rc = bn_args_ok("bn_backward_apply", n_pix, C, dtype)) return rc;
    return launch_bn_bwd_apply(dz, u, a, b, c, du, n_pix, C, dtype, (hipStream_t)stream);
}

int ymx_bbox_iou(const float *b1, const float *b2, float *out, long long n, int giou, void *stream) {
    if (!b1 || !b2 || !out) return fail(YMX_EARG, "bbox_iou: null pointer");
    if (n < 0) return fail(YMX_EARG, "bbox_iou: negative count");
    if (n == 0) return YMX_OK;
    if (!aligned16(b1) || !aligned16(b2)) return fail(YMX_EARG, "bbox_iou: boxes must be 16-byte aligned");
    return launch_bbox_iou(b1, b2, out, n, giou, (hipStream_t)stream);
}

}  // extern "C"
