// Image-side utilities of the data path:
//   - letterbox resize of utils.image_preprocess (/root/reference/utils.py:16-26): bilinear
//     resize that keeps the aspect ratio, centred in a square filled with 128, divided by 256;
//   - box outlines of utils.draw_bbox (/root/reference/utils.py:216-231).
#include <algorithm>

#include "common.h"

namespace ymx {

// ---- letterbox ---------------------------------------------------------------------------------
// One thread per output pixel.  Inside the pasted region the value is the bilinear sample of
// the source with half-pixel centres and edge replication, the arithmetic OpenCV's
// INTER_LINEAR uses for float images: the source coordinate in double, f = (float)((d + 0.5) *
// (src / dst) - 0.5), i = floor(f), weight f - i (0 at a clamped edge); rows first
// (S[i] * (1 - fx) + S[i + 1] * fx) then columns, each product and sum rounded to fp32 (no fused
// multiply-add, so that a plain fp32 restatement gives the same bits).  8-bit sources are
// converted to float first.  Outside the region 128; everything divided by 256.
struct LetterboxArgs {
    const void *src;
    void *dst;
    int B, H, W, T;
    int r0, c0, rh, rw;          // pasted region: rows [r0, r0 + rh), columns [c0, c0 + rw)
    double sy, sx;               // source / destination size ratio per axis
    int in_u8, out_bf16;
};

__device__ __forceinline__ void lin_coord(int d, double ratio, int n, int &i0, int &i1, float &f) {
#pragma clang fp contract(off)
    float v = (float)(((double)d + 0.5) * ratio - 0.5);
    int i = (int)floorf(v);
    f = v - (float)i;
    if (i < 0) { i = 0; f = 0.0f; }
    if (i >= n - 1) { i = n - 1; f = 0.0f; }
    i0 = i;
    i1 = i + 1 < n ? i + 1 : n - 1;
}

__global__ void __launch_bounds__(256)
letterbox_kernel(const LetterboxArgs a) {
#pragma clang fp contract(off)
    const size_t total = (size_t)a.B * a.T * a.T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % a.T);
        const size_t r = i / a.T;
        const int oy = (int)(r % a.T), b = (int)(r / a.T);
        float v[3] = {128.0f, 128.0f, 128.0f};
        const int dy = oy - a.r0, dx = ox - a.c0;
        if (dy >= 0 && dy < a.rh && dx >= 0 && dx < a.rw) {
            int y0, y1, x0, x1;
            float fy, fx;
            lin_coord(dy, a.sy, a.H, y0, y1, fy);
            lin_coord(dx, a.sx, a.W, x0, x1, fx);
            const float ax0 = 1.0f - fx, by0 = 1.0f - fy;
            const size_t base = (size_t)b * a.H * a.W;
            const size_t p00 = (base + (size_t)y0 * a.W + x0) * 3, p01 = (base + (size_t)y0 * a.W + x1) * 3;
            const size_t p10 = (base + (size_t)y1 * a.W + x0) * 3, p11 = (base + (size_t)y1 * a.W + x1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s00, s01, s10, s11;
                if (a.in_u8) {
                    const unsigned char *s = (const unsigned char *)a.src;
                    s00 = (float)s[p00 + c]; s01 = (float)s[p01 + c]; s10 = (float)s[p10 + c]; s11 = (float)s[p11 + c];
                } else {
                    const float *s = (const float *)a.src;
                    s00 = s[p00 + c]; s01 = s[p01 + c]; s10 = s[p10 + c]; s11 = s[p11 + c];
                }
                const float top = s00 * ax0 + s01 * fx;          // contraction is off in this function
                const float bot = s10 * ax0 + s11 * fx;
                v[c] = top * by0 + bot * fy;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float o = v[c] * (1.0f / 256.0f);
            if (a.out_bf16) ((bf16_t *)a.dst)[i * 3 + c] = f32_to_bf16(o);
            else ((float *)a.dst)[i * 3 + c] = o;
        }
    }
}

int launch_letterbox(const void *src, int in_u8, int B, int H, int W, void *dst, int out_bf16, int T, int literal, hipStream_t s) {
    // utils.py:19-24, in Python's double arithmetic
    const double scale = std::min((double)T / (double)W, (double)T / (double)H);
    const int new_h = (int)(scale * (double)H), new_w = (int)(scale * (double)W);
    if (new_h < 1 || new_w < 1) return fail(YMX_EUNSUPPORTED, "letterbox: a %dx%d image vanishes at target size %d", H, W, T);
    const int pad_h = (T - new_h) / 2, pad_w = (T - new_w) / 2;
    LetterboxArgs a;
    a.src = src; a.dst = dst; a.B = B; a.H = H; a.W = W; a.T = T;
    if (literal) {
        // utils.py:21,25 as written: cv2.resize takes (width, height), so the resized image
        // has new_width rows and new_height columns, and it is pasted at rows pad_width...
        a.rh = new_w; a.rw = new_h; a.r0 = pad_w; a.c0 = pad_h;
    } else {
        a.rh = new_h; a.rw = new_w; a.r0 = pad_h; a.c0 = pad_w;
    }
    a.sy = (double)H / (double)a.rh;
    a.sx = (double)W / (double)a.rw;
    a.in_u8 = in_u8; a.out_bf16 = out_bf16;
    const size_t total = (size_t)B * T * T;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(letterbox_kernel, dim3(grid), dim3(256), 0, s, a);
    return check_launch("letterbox");
}

// ---- box outlines --------------------------------------------------------------------------
// One thread per pixel; the boxes of the image are visited in order and the last outline that
// covers the pixel wins, which is what drawing them one after another gives.  Corners are
// (int)(coordinate * size) as utils.py:220-223; an outline of thickness t covers, on each of
// the four sides, t pixel lines around the side ((t - 1) / 2 towards lower coordinates, t / 2
// towards higher ones), over the whole extent of the box
// enlarged by the same amounts.  Detections with a class outside [0, C] are skipped
// (utils.py:218 tests "> num_classes"; class == C has no colour and is skipped here too).
struct DrawArgs {
    unsigned char *img;
    const float *boxes, *classes;
    const int *valid;
    const unsigned char *colors;
    int B, H, W, M, C, thick;
};

__global__ void __launch_bounds__(256)
draw_boxes_kernel(const DrawArgs a) {
    const size_t total = (size_t)a.B * a.H * a.W;
    const int lo = (a.thick - 1) / 2, hi = a.thick / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % a.W);
        const size_t r = i / a.W;
        const int y = (int)(r % a.H), b = (int)(r / a.H);
        int nb = a.valid[b];
        nb = nb > a.M ? a.M : nb;
        int chosen = -1;
        for (int j = 0; j < nb; ++j) {
            const int cls = (int)a.classes[(size_t)b * a.M + j];
            if (cls < 0 || cls >= a.C) continue;
            const float *bx = a.boxes + ((size_t)b * a.M + j) * 4;
            const int y1 = (int)(bx[0] * (float)a.H), x1 = (int)(bx[1] * (float)a.W);
            const int y2 = (int)(bx[2] * (float)a.H), x2 = (int)(bx[3] * (float)a.W);
            const int ya = min(y1, y2), yb = max(y1, y2), xa = min(x1, x2), xb = max(x1, x2);
            const bool in_outer = x >= xa - lo && x <= xb + hi && y >= ya - lo && y <= yb + hi;
            if (!in_outer) continue;
            const bool on_v = (x >= xa - lo && x <= xa + hi) || (x >= xb - lo && x <= xb + hi);
            const bool on_h = (y >= ya - lo && y <= ya + hi) || (y >= yb - lo && y <= yb + hi);
            if (on_v || on_h) chosen = cls;
        }
        if (chosen >= 0) {
            unsigned char *p = a.img + i * 3;
            p[0] = a.colors[chosen * 3]; p[1] = a.colors[chosen * 3 + 1]; p[2] = a.colors[chosen * 3 + 2];
        }
    }
}

int launch_draw_boxes(unsigned char *img, int B, int H, int W, const float *boxes, const float *classes, const int *valid, int M,
                      const unsigned char *colors, int C, int thick, hipStream_t s) {
    DrawArgs a;
    a.img = img; a.boxes = boxes; a.classes = classes; a.valid = valid; a.colors = colors;
    a.B = B; a.H = H; a.W = W; a.M = M; a.C = C; a.thick = thick < 1 ? 1 : thick;
    const size_t total = (size_t)B * H * W;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(draw_boxes_kernel, dim3(grid), dim3(256), 0, s, a);
    return check_launch("draw_boxes");
}

}  // namespace ymx
