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

// ---- box outlines and labels ----------------------------------------------------------------
// One thread per pixel; the boxes of the image are visited in order and the last thing drawn
// over the pixel wins, which is what drawing them one after another gives.  Corners are
// (int)(coordinate * size) as utils.py:220-223; an outline of thickness t covers, on each of
// the four sides, t pixel lines around the side ((t - 1) / 2 towards lower coordinates, t / 2
// towards higher ones), over the whole extent of the box
// enlarged by the same amounts.  Detections with a class outside [0, C] are skipped
// (utils.py:218 tests "> num_classes"; class == C has no colour and is skipped here too).
//
// Labels (utils.py:233-240): after its outline, a detection with a non-empty label gets a
// rectangle filled with the class colour from the corner c1 = (x1, y1) to
// (x1 + text_width, y1 - text_height - 3), both corners included, and the text in black on
// it with its lowest pixel row at y1 - 3 (cv2.putText's origin (x1, y1 - 2) is the baseline,
// drawn below the glyph bodies).  The typeface is this library's own: a 5x7 bitmap in a 6-wide
// cell, every font pixel a scale x scale square, so text_width = 6 * scale * length and
// text_height = 7 * scale (OpenCV's Hershey strokes are not reproduced).
// Codes outside 32..126 show as a hollow box.
__constant__ unsigned char kFont5x7[95][5] = {   // column-major, bit r of a column = row r from the top
    {0x00,0x00,0x00,0x00,0x00}, {0x00,0x00,0x5F,0x00,0x00}, {0x00,0x07,0x00,0x07,0x00}, {0x14,0x7F,0x14,0x7F,0x14}, {0x24,0x2A,0x7F,0x2A,0x12},
    {0x23,0x13,0x08,0x64,0x62}, {0x36,0x49,0x55,0x22,0x50}, {0x00,0x05,0x03,0x00,0x00}, {0x00,0x1C,0x22,0x41,0x00}, {0x00,0x41,0x22,0x1C,0x00},
    {0x14,0x08,0x3E,0x08,0x14}, {0x08,0x08,0x3E,0x08,0x08}, {0x00,0x50,0x30,0x00,0x00}, {0x08,0x08,0x08,0x08,0x08}, {0x00,0x60,0x60,0x00,0x00},
    {0x20,0x10,0x08,0x04,0x02}, {0x3E,0x51,0x49,0x45,0x3E}, {0x00,0x42,0x7F,0x40,0x00}, {0x42,0x61,0x51,0x49,0x46}, {0x21,0x41,0x45,0x4B,0x31},
    {0x18,0x14,0x12,0x7F,0x10}, {0x27,0x45,0x45,0x45,0x39}, {0x3C,0x4A,0x49,0x49,0x30}, {0x01,0x71,0x09,0x05,0x03}, {0x36,0x49,0x49,0x49,0x36},
    {0x06,0x49,0x49,0x29,0x1E}, {0x00,0x36,0x36,0x00,0x00}, {0x00,0x56,0x36,0x00,0x00}, {0x08,0x14,0x22,0x41,0x00}, {0x14,0x14,0x14,0x14,0x14},
    {0x00,0x41,0x22,0x14,0x08}, {0x02,0x01,0x51,0x09,0x06}, {0x32,0x49,0x79,0x41,0x3E}, {0x7E,0x11,0x11,0x11,0x7E}, {0x7F,0x49,0x49,0x49,0x36},
    {0x3E,0x41,0x41,0x41,0x22}, {0x7F,0x41,0x41,0x22,0x1C}, {0x7F,0x49,0x49,0x49,0x41}, {0x7F,0x09,0x09,0x09,0x01}, {0x3E,0x41,0x49,0x49,0x7A},
    {0x7F,0x08,0x08,0x08,0x7F}, {0x00,0x41,0x7F,0x41,0x00}, {0x20,0x40,0x41,0x3F,0x01}, {0x7F,0x08,0x14,0x22,0x41}, {0x7F,0x40,0x40,0x40,0x40},
    {0x7F,0x02,0x0C,0x02,0x7F}, {0x7F,0x04,0x08,0x10,0x7F}, {0x3E,0x41,0x41,0x41,0x3E}, {0x7F,0x09,0x09,0x09,0x06}, {0x3E,0x41,0x51,0x21,0x5E},
    {0x7F,0x09,0x19,0x29,0x46}, {0x46,0x49,0x49,0x49,0x31}, {0x01,0x01,0x7F,0x01,0x01}, {0x3F,0x40,0x40,0x40,0x3F}, {0x1F,0x20,0x40,0x20,0x1F},
    {0x3F,0x40,0x38,0x40,0x3F}, {0x63,0x14,0x08,0x14,0x63}, {0x07,0x08,0x70,0x08,0x07}, {0x61,0x51,0x49,0x45,0x43}, {0x00,0x7F,0x41,0x41,0x00},
    {0x02,0x04,0x08,0x10,0x20}, {0x00,0x41,0x41,0x7F,0x00}, {0x04,0x02,0x01,0x02,0x04}, {0x40,0x40,0x40,0x40,0x40}, {0x00,0x01,0x02,0x04,0x00},
    {0x20,0x54,0x54,0x54,0x78}, {0x7F,0x48,0x44,0x44,0x38}, {0x38,0x44,0x44,0x44,0x20}, {0x38,0x44,0x44,0x48,0x7F}, {0x38,0x54,0x54,0x54,0x18},
    {0x08,0x7E,0x09,0x01,0x02}, {0x0C,0x52,0x52,0x52,0x3E}, {0x7F,0x08,0x04,0x04,0x78}, {0x00,0x44,0x7D,0x40,0x00}, {0x20,0x40,0x44,0x3D,0x00},
    {0x7F,0x10,0x28,0x44,0x00}, {0x00,0x41,0x7F,0x40,0x00}, {0x7C,0x04,0x18,0x04,0x78}, {0x7C,0x08,0x04,0x04,0x78}, {0x38,0x44,0x44,0x44,0x38},
    {0x7C,0x14,0x14,0x14,0x08}, {0x08,0x14,0x14,0x18,0x7C}, {0x7C,0x08,0x04,0x04,0x08}, {0x48,0x54,0x54,0x54,0x20}, {0x04,0x3F,0x44,0x40,0x20},
    {0x3C,0x40,0x40,0x20,0x7C}, {0x1C,0x20,0x40,0x20,0x1C}, {0x3C,0x40,0x30,0x40,0x3C}, {0x44,0x28,0x10,0x28,0x44}, {0x0C,0x50,0x50,0x50,0x3C},
    {0x44,0x64,0x54,0x4C,0x44}, {0x00,0x08,0x36,0x41,0x00}, {0x00,0x00,0x7F,0x00,0x00}, {0x00,0x41,0x36,0x08,0x00}, {0x08,0x04,0x08,0x10,0x08},
};

struct DrawArgs {
    unsigned char *img;
    const float *boxes, *classes;
    const int *valid;
    const unsigned char *colors;
    const unsigned char *labels;    // [B, M, L] zero-padded, or null
    int B, H, W, M, C, thick, L, scale;
};

__global__ void __launch_bounds__(256)
draw_boxes_kernel(const DrawArgs a) {
    const size_t total = (size_t)a.B * a.H * a.W;
    const int lo = (a.thick - 1) / 2, hi = a.thick / 2;
    const int cell = 6 * a.scale, th = 7 * a.scale;
    constexpr int BLACK = -2;
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
            if (in_outer) {
                const bool on_v = (x >= xa - lo && x <= xa + hi) || (x >= xb - lo && x <= xb + hi);
                const bool on_h = (y >= ya - lo && y <= ya + hi) || (y >= yb - lo && y <= yb + hi);
                if (on_v || on_h) chosen = cls;
            }
            if (!a.labels) continue;
            const int dx = x - x1, dy = y - (y1 - th - 3);      // offsets inside the label rectangle
            if (dx < 0 || dx > cell * a.L || dy < 0 || dy > th + 3) continue;
            const unsigned char *lab = a.labels + ((size_t)b * a.M + j) * a.L;
            if (lab[0] == 0) continue;
            const int k = dx / cell, cx = dx - k * cell;
            const unsigned char ch = k < a.L ? lab[k] : (unsigned char)0;
            if (ch == 0) {
                // right edge of the rectangle: the column x1 + text_width itself
                if (cx == 0 && lab[k - 1] != 0) chosen = cls;
                continue;
            }
            chosen = cls;
            const int gy = dy - 1;                              // glyph rows are dy = 1 .. th
            if (gy >= 0 && gy < th && cx < 5 * a.scale) {
                const int code = ch >= 32 && ch <= 126 ? ch - 32 : -1;
                const int col = cx / a.scale, row = gy / a.scale;
                const unsigned bits = code >= 0 ? kFont5x7[code][col]
                                                : ((col == 0 || col == 4) ? 0x7Fu : 0x41u);
                if ((bits >> row) & 1u) chosen = BLACK;
            }
        }
        if (chosen >= 0) {
            unsigned char *p = a.img + i * 3;
            p[0] = a.colors[chosen * 3]; p[1] = a.colors[chosen * 3 + 1]; p[2] = a.colors[chosen * 3 + 2];
        } else if (chosen == BLACK) {
            unsigned char *p = a.img + i * 3;
            p[0] = 0; p[1] = 0; p[2] = 0;
        }
    }
}

int launch_draw_boxes(unsigned char *img, int B, int H, int W, const float *boxes, const float *classes, const int *valid, int M,
                      const unsigned char *colors, int C, int thick, const unsigned char *labels, int L, int scale,
                      hipStream_t s) {
    DrawArgs a;
    a.img = img; a.boxes = boxes; a.classes = classes; a.valid = valid; a.colors = colors;
    a.B = B; a.H = H; a.W = W; a.M = M; a.C = C; a.thick = thick < 1 ? 1 : thick;
    a.labels = (labels && L > 0) ? labels : nullptr;
    a.L = a.labels ? L : 0; a.scale = scale < 1 ? 1 : scale;
    const size_t total = (size_t)B * H * W;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(draw_boxes_kernel, dim3(grid), dim3(256), 0, s, a);
    return check_launch("draw_boxes");
}

}  // namespace ymx
