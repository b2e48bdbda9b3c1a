// Direct (non-MFMA) fused convolution and the weight packing kernels.
//
// The direct kernel serves input-channel counts that cannot form 16-byte K chunks,
// which in Darknet53 is the first layer only (Cin = 3, K = 27, 32 outputs:
// backbones.py:11).  That layer moves 64 B of output per 27 multiply-adds per
// channel, so it is bandwidth- and VALU-bound, not MFMA-shaped: one lane per output
// pixel, 32 output channels per lane in registers, weights read through the scalar
// path (their address is the same in every lane).
//
// Weight layout for this path: the reference's own HWIO order, i.e. a dense
// [K = k*k*Cin][Cout] fp32 matrix.
#include "common.h"

namespace ymx {

template <typename XT, typename OT>
__global__ void __launch_bounds__(256)
conv_direct_kernel(const ConvArgs a) {
    constexpr int CO = 32;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int q0 = blockIdx.y * CO;
    if (p >= a.P) return;
    const XT *__restrict__ x = (const XT *)a.x;
    const float *__restrict__ w = (const float *)a.w;
    const int ohw = a.OH * a.OW;
    const int b = p / ohw, rem = p - b * ohw;
    const int oh = rem / a.OW, ow = rem - oh * a.OW;
    const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;

    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.0f;

    for (int r = 0; r < a.k; ++r) {
        const int ih = ih0 + r;
        for (int c = 0; c < a.k; ++c) {
            const int iw = iw0 + c;
            const bool ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const XT *xp = x + ((size_t)(b * a.H + (ok ? ih : 0)) * a.W + (ok ? iw : 0)) * a.x_ld;
            const float *wp = w + (size_t)((r * a.k + c) * a.Cin) * a.Cout;
            for (int ci = 0; ci < a.Cin; ++ci) {
                const float xv = ok ? elem<XT>::load(xp + ci) : 0.0f;
                const float *wr = wp + (size_t)ci * a.Cout;
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const int cc = min(q0 + co, a.Cout - 1);   // uniform: scalar load
                    acc[co] = fmaf(xv, wr[cc], acc[co]);
                }
            }
        }
    }

    OT *__restrict__ y = (OT *)a.y + (size_t)p * a.y_ld;
    const OT *__restrict__ res = a.res ? (const OT *)a.res + (size_t)p * a.res_ld : nullptr;
    float v[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const int cc = min(q0 + co, a.Cout - 1);
        const float sc = a.scale ? a.scale[cc] : 1.0f;
        const float sh = a.shift ? a.shift[cc] : 0.0f;
        v[co] = act_f(acc[co] * sc + sh, a.act);
        if (res) v[co] += elem<OT>::load(res + cc);
    }
    const bool full = q0 + CO <= a.Cout;
    if (full && sizeof(OT) == 2 && (a.y_ld & 7) == 0 && (a.Cout & 7) == 0) {
        uint4 *yp = (uint4 *)(y + q0);
#pragma unroll
        for (int g = 0; g < CO / 8; ++g)
            yp[g] = make_uint4(pack2_bf16(v[8 * g + 0], v[8 * g + 1]), pack2_bf16(v[8 * g + 2], v[8 * g + 3]),
                               pack2_bf16(v[8 * g + 4], v[8 * g + 5]), pack2_bf16(v[8 * g + 6], v[8 * g + 7]));
    } else if (full && sizeof(OT) == 4 && (a.y_ld & 3) == 0 && (a.Cout & 3) == 0) {
        float4 *yp = (float4 *)(y + q0);
#pragma unroll
        for (int g = 0; g < CO / 4; ++g) yp[g] = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    } else {
#pragma unroll
        for (int co = 0; co < CO; ++co)
            if (q0 + co < a.Cout) elem<OT>::store(y + q0 + co, v[co]);
    }
}

int launch_conv_direct(const ConvArgs &a, hipStream_t s) {
    dim3 grid((a.P + 255) / 256, (a.Cout + 31) / 32), block(256);
    // the shortcut, when present, has the output's element type on this path
    if (a.x_dtype == YMX_F32 && a.out_dtype == YMX_F32)
        hipLaunchKernelGGL((conv_direct_kernel<float, float>), grid, block, 0, s, a);
    else if (a.x_dtype == YMX_F32 && a.out_dtype == YMX_BF16)
        hipLaunchKernelGGL((conv_direct_kernel<float, bf16_t>), grid, block, 0, s, a);
    else if (a.x_dtype == YMX_BF16 && a.out_dtype == YMX_BF16)
        hipLaunchKernelGGL((conv_direct_kernel<bf16_t, bf16_t>), grid, block, 0, s, a);
    else if (a.x_dtype == YMX_BF16 && a.out_dtype == YMX_F32)
        hipLaunchKernelGGL((conv_direct_kernel<bf16_t, float>), grid, block, 0, s, a);
    else
        return fail(YMX_EUNSUPPORTED, "conv_direct: x_dtype %d / out_dtype %d", a.x_dtype, a.out_dtype);
    return check_launch("conv_direct");
}

// ---- weight packing ------------------------------------------------------------
// MFMA path: HWIO fp32 [K][Cout] -> [rows = Cout rounded up to 128][Kw = K rounded up
// to 64] of T, row-major, zero padded; K order (r, c, ci) is HWIO's own flattening.
template <typename T>
__global__ void __launch_bounds__(256)
pack_weights_kernel(const float *__restrict__ w, T *__restrict__ out, int K, int Cout, int Kw, int rows) {
    const size_t n = (size_t)rows * Kw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i / Kw), kk = (int)(i - (size_t)co * Kw);
        const float v = (co < Cout && kk < K) ? w[(size_t)kk * Cout + co] : 0.0f;
        elem<T>::store(out + i, v);
    }
}
// Direct path: same [K][Cout] fp32 layout; with bf16 compute the values are rounded
// to bf16 precision so that every layer of a bf16 network sees bf16 weights.
__global__ void __launch_bounds__(256)
copy_weights_kernel(const float *__restrict__ w, float *__restrict__ out, size_t n, int round_bf16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = w[i];
        out[i] = round_bf16 ? bf16_to_f32(f32_to_bf16(v)) : v;
    }
}

int launch_pack_weights(const float *w_hwio, void *w_packed, int k, int Cin, int Cout, int dtype,
                        hipStream_t s) {
    const int K = k * k * Cin;
    if (!conv_uses_mfma(Cin, dtype)) {
        const size_t n = (size_t)K * Cout;
        const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(copy_weights_kernel, dim3(grid), dim3(256), 0, s, w_hwio, (float *)w_packed, n,
                           dtype == YMX_BF16 ? 1 : 0);
        return check_launch("copy_weights");
    }
    const int Kw = packed_kw(k, Cin), rows = packed_rows(Cout);
    const size_t n = (size_t)rows * Kw;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == YMX_BF16)
        hipLaunchKernelGGL((pack_weights_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, w_hwio, (bf16_t *)w_packed, K, Cout, Kw, rows);
    else
        hipLaunchKernelGGL((pack_weights_kernel<float>), dim3(grid), dim3(256), 0, s, w_hwio, (float *)w_packed, K, Cout, Kw, rows);
    return check_launch("pack_weights");
}

}  // namespace ymx
