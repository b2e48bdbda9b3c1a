// Bandwidth-bound elementwise kernels: bilinear x2 upsample fused with the channel
// concat that follows it in the head, and centre-format box IoU / GIoU.
#include "common.h"

namespace ymx {

// One thread per (output pixel, 16-byte channel chunk).  Chunks below C1 copy the
// skip tensor; the rest interpolate x with half-pixel centres and edge clamp:
//   src = (dst + 0.5)/2 - 0.5;  lo = max(floor(src), 0);  hi = min(ceil(src), n-1);
//   out = top + (bottom - top) * fy,  top = tl + (tr - tl) * fx   (fp32 arithmetic).
template <typename T>
__global__ void __launch_bounds__(256)
upsample2x_concat_kernel(const T *__restrict__ skip, const T *__restrict__ x, T *__restrict__ y,
                         int B, int H, int W, int C1, int C2) {
    constexpr int KCH = 16 / (int)sizeof(T);
    const int CT = C1 + C2, nch = CT / KCH, nch1 = C1 / KCH;
    const int OH = 2 * H, OW = 2 * W;
    // skip == nullptr with C1 > 0: the first C1 channels of y are already in place (their producer wrote
    // them there); only the resized channels are visited
    const int first = skip ? 0 : nch1, nloop = nch - first;
    const size_t total = (size_t)B * OH * OW * nloop;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ch = first + (int)(i % nloop);
        const size_t pix = i / nloop;
        uint4 out;
        if (ch < nch1) {
            out = *(const uint4 *)(skip + pix * C1 + (size_t)ch * KCH);
        } else {
            const int ox = (int)(pix % OW);
            const size_t t2 = pix / OW;
            const int oy = (int)(t2 % OH), b = (int)(t2 / OH);
            const float sy = (oy + 0.5f) * 0.5f - 0.5f, sx = (ox + 0.5f) * 0.5f - 0.5f;
            const float fy0 = floorf(sy), fx0 = floorf(sx);
            const int y0 = max((int)fy0, 0), y1 = min((int)ceilf(sy), H - 1);
            const int x0 = max((int)fx0, 0), x1 = min((int)ceilf(sx), W - 1);
            const float fy = sy - fy0, fx = sx - fx0;
            const int c = (ch - nch1) * KCH;
            const T *base = x + (size_t)b * H * W * C2 + c;
            const uint4 tl = *(const uint4 *)(base + ((size_t)y0 * W + x0) * C2);
            const uint4 tr = *(const uint4 *)(base + ((size_t)y0 * W + x1) * C2);
            const uint4 bl = *(const uint4 *)(base + ((size_t)y1 * W + x0) * C2);
            const uint4 br = *(const uint4 *)(base + ((size_t)y1 * W + x1) * C2);
            auto lerp = [&](float a, float bq, float c2, float d) {
                const float top = a + (bq - a) * fx, bot = c2 + (d - c2) * fx;
                return top + (bot - top) * fy;
            };
            if constexpr (sizeof(T) == 2) {
                const uint32_t *a = (const uint32_t *)&tl, *bq = (const uint32_t *)&tr;
                const uint32_t *c2 = (const uint32_t *)&bl, *d = (const uint32_t *)&br;
                uint32_t o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = lerp(__uint_as_float(a[k] << 16), __uint_as_float(bq[k] << 16),
                                          __uint_as_float(c2[k] << 16), __uint_as_float(d[k] << 16));
                    const float hi = lerp(__uint_as_float(a[k] & 0xffff0000u), __uint_as_float(bq[k] & 0xffff0000u),
                                          __uint_as_float(c2[k] & 0xffff0000u), __uint_as_float(d[k] & 0xffff0000u));
                    o[k] = pack2_bf16(lo, hi);
                }
                out = make_uint4(o[0], o[1], o[2], o[3]);
            } else {
                const float *a = (const float *)&tl, *bq = (const float *)&tr;
                const float *c2 = (const float *)&bl, *d = (const float *)&br;
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = lerp(a[k], bq[k], c2[k], d[k]);
                out = make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
            }
        }
        store_stream16(y + pix * CT + (size_t)ch * KCH, out.x, out.y, out.z, out.w);
    }
}

int launch_upsample2x_concat(const void *skip, const void *x, void *y, int B, int H, int W, int C1,
                             int C2, int dtype, hipStream_t s) {
    const int kch = dtype == YMX_BF16 ? 8 : 4;
    const size_t total = (size_t)B * 2 * H * 2 * W * (((skip ? C1 : 0) + C2) / kch);
    const size_t blocks = (total + 255) / 256;
    const int grid = (int)(blocks < 16384 ? blocks : 16384);
    if (dtype == YMX_BF16)
        hipLaunchKernelGGL((upsample2x_concat_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t *)skip,
                           (const bf16_t *)x, (bf16_t *)y, B, H, W, C1, C2);
    else
        hipLaunchKernelGGL((upsample2x_concat_kernel<float>), dim3(grid), dim3(256), 0, s, (const float *)skip,
                           (const float *)x, (float *)y, B, H, W, C1, C2);
    return check_launch("upsample2x_concat");
}

// IoU / GIoU of centre-format boxes, one pair per thread.
__global__ void __launch_bounds__(256)
bbox_iou_kernel(const float4 *__restrict__ b1, const float4 *__restrict__ b2, float *__restrict__ out,
                long long n, int giou) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float4 a = b1[i], b = b2[i];
        const float area1 = a.z * a.w, area2 = b.z * b.w;
        const float ax0 = a.x - a.z * 0.5f, ay0 = a.y - a.w * 0.5f, ax1 = a.x + a.z * 0.5f, ay1 = a.y + a.w * 0.5f;
        const float bx0 = b.x - b.z * 0.5f, by0 = b.y - b.w * 0.5f, bx1 = b.x + b.z * 0.5f, by1 = b.y + b.w * 0.5f;
        const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.0f);
        const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.0f);
        const float inter = iw * ih;
        const float uni = area1 + area2 - inter;
        float r = uni != 0.0f ? inter / uni : 0.0f;
        if (giou) {
            const float ew = fmaxf(ax1, bx1) - fminf(ax0, bx0), eh = fmaxf(ay1, by1) - fminf(ay0, by0);
            const float earea = ew * eh;
            r -= earea != 0.0f ? (earea - uni) / earea : 0.0f;
        }
        out[i] = r;
    }
}

int launch_bbox_iou(const float *b1, const float *b2, float *out, long long n, int giou, hipStream_t s) {
    const long long blocks = (n + 255) / 256;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(bbox_iou_kernel, dim3(grid), dim3(256), 0, s, (const float4 *)b1, (const float4 *)b2, out, n, giou);
    return check_launch("bbox_iou");
}

// y = act(x * scale[c] + shift[c]) on an NHWC tensor viewed as [n_pix][C]; scale and
// shift may be null (1 and 0).  Serves the stand-alone Mish and BatchNormalization
// layers; inside the network both are fused into the convolution epilogue instead.
template <typename T>
__global__ void __launch_bounds__(256)
affine_act_kernel(const T *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                  const T *__restrict__ res, T *__restrict__ y, long long n, int C, int act) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        float v = elem<T>::load(x + i);
        v = v * (scale ? scale[c] : 1.0f) + (shift ? shift[c] : 0.0f);
        v = act_f(v, act);
        if (res) v += elem<T>::load(res + i);                       // the shortcut of a ResidialBlock, after the activation
        elem<T>::store(y + i, v);
    }
}

int launch_affine_act(const void *x, const float *scale, const float *shift, void *y, long long n, int C, int act,
                      int dtype, hipStream_t s, const void *res) {
    const long long blocks = (n + 255) / 256;
    const int grid = (int)(blocks < 16384 ? blocks : 16384);
    if (dtype == YMX_BF16)
        hipLaunchKernelGGL((affine_act_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t *)x, scale, shift,
                           (const bf16_t *)res, (bf16_t *)y, n, C, act);
    else
        hipLaunchKernelGGL((affine_act_kernel<float>), dim3(grid), dim3(256), 0, s, (const float *)x, scale, shift,
                           (const float *)res, (float *)y, n, C, act);
    return check_launch("affine_act");
}

}  // namespace ymx
