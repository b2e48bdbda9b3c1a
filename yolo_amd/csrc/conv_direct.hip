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
        v[co] = act_f(affine_f(acc[co], sc, sh), a.act);
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

// ---- first layer on MFMA: 3x3, Cin = 3, Cout = 32, stride 1, bf16 network ----------
// K = 27 fits one mfma_f32_16x16x32_bf16 (5 zero columns).  The 27 inputs of a pixel are
// three runs of 9 contiguous values (3 pixels x 3 channels of one image row), so the
// pixel operand is gathered straight from global memory: lane (pixel lane&15, k-group
// lane>>4) loads k = 8g .. 8g+7, converts to bf16 and packs.  Weights are the A operand
// and stay in registers for the whole kernel; output row i of channel tile t holds
// channel 8*(i>>2) + 4*t + (i&3), so after both tiles a lane owns 8 consecutive channels
// of its pixel and the epilogue (bias, activation) ends in one 16-byte store per pixel.
// The layer writes 64 B per pixel for 27 MACs per channel: it is HBM-bound (709 MB per
// batch-64 step at 416x416), and the gather costs ~1/8 of the VALU work of the scalar form.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_first_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_first_t;

template <typename XT>
__global__ void __launch_bounds__(256)
conv_first_mfma_kernel(const ConvArgs a, const int nchunk) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, li = lane & 15;
    const XT *__restrict__ x = (const XT *)a.x;
    const float *__restrict__ w = (const float *)a.w;          // [27][32] fp32 (HWIO)
    bf16_t *__restrict__ y = (bf16_t *)a.y;

    // A fragments: tile t, row li -> channel 8*(li>>2) + 4t + (li&3); k = 8g + j
    bf16x8_first_t af[2];
    int koff[8];            // element offset of k = 8g+j relative to the pixel's (b, oh, ow, 0)
    int kdr[8], kdc[8];     // row / column displacement of that tap; a huge kdr marks k >= 27
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        const int r = k / 9, c = (k - 9 * r) / 3, ci = k - 9 * r - 3 * c;
        const bool kin = k < 27;
        kdr[j] = kin ? r - 1 : (1 << 28);
        kdc[j] = c - 1;
        koff[j] = ((r - 1) * a.W + (c - 1)) * a.x_ld + ci;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ch = 8 * (li >> 2) + 4 * t + (li & 3);
            af[t][j] = (__bf16)(kin ? w[k * 32 + ch] : 0.0f);
        }
    }
    float sc[8], sh[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        sc[r] = a.scale ? a.scale[8 * g + r] : 1.0f;
        sh[r] = a.shift ? a.shift[8 * g + r] : 0.0f;
    }
    const int hw = a.OH * a.OW;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    for (int chunk = wave_global; chunk < nchunk; chunk += nwaves) {
        // 64 pixels per wave per iteration: four 16-pixel B fragments
        bf16x8_first_t bf[4];
        int pix[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int p = chunk * 64 + f * 16 + li;
            pix[f] = p;
            const int pp = p < a.P ? p : a.P - 1;
            const int b = pp / hw, rem = pp - b * hw;
            const int oh = rem / a.OW, ow = rem - oh * a.OW;
            const XT *base = x + ((size_t)(b * a.H + oh) * a.W + ow) * a.x_ld;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = (unsigned)(oh + kdr[j]) < (unsigned)a.H && (unsigned)(ow + kdc[j]) < (unsigned)a.W;
                const float v = ok ? elem<XT>::load(base + koff[j]) : 0.0f;
                bf[f][j] = (__bf16)v;
            }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            f32x4_first_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[f], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[f], d1, 0, 0, 0);
            if (pix[f] < a.P) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = act_f(affine_f(d0[r], sc[r], sh[r]), a.act);
                    v[4 + r] = act_f(affine_f(d1[r], sc[4 + r], sh[4 + r]), a.act);
                }
                store_stream16(y + (size_t)pix[f] * a.y_ld + 8 * g, pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]),
                               pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
            }
        }
    }
}

// Same layer, fp32 image, one wave per 64-pixel run of one output row.  All pixel index
// arithmetic is per wave (no per-lane division).  The K axis is re-ordered so that every
// lane's operand is one contiguous run of the image: k-group g = tap row g holds the first
// 8 of that row's 9 values (3 pixels x 3 channels), read as three 12-byte buffer loads
// (12-byte pixels are only dword aligned; out-of-range offsets return zeros); the 9th
// value of each row goes to a second MFMA (slot 0 of group g).  Weights are permuted to
// match when the A fragments are built.  The loads of a wave's next run are issued before
// the arithmetic of the current one (the layer is HBM-bound: 0.13 GB in, 0.71 GB out per
// 64 images), and the activation is a template parameter so the per-value code is
// straight-line.
template <int ACT>
__global__ void __launch_bounds__(256)
conv_first_rows_kernel(const ConvArgs a, const int chunks_per_row, const int nchunk) {
    typedef __attribute__((ext_vector_type(3))) unsigned int u3;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, li = lane & 15;
    const float *__restrict__ w = (const float *)a.w;          // [27][32] fp32 (HWIO)
    bf16_t *__restrict__ y = (bf16_t *)a.y;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);

    bf16x8_first_t a1[2], a2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ch = 8 * (li >> 2) + 4 * t + (li & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a1[t][j] = (__bf16)(g < 3 ? w[(g * 9 + j) * 32 + ch] : 0.0f);
            a2[t][j] = (__bf16)((g < 3 && j == 0) ? w[(g * 9 + 8) * 32 + ch] : 0.0f);
        }
    }
    float sc[8], sh[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        sc[r] = a.scale ? a.scale[8 * g + r] : 1.0f;
        sh[r] = a.shift ? a.shift[8 * g + r] : 0.0f;
    }
    // Every wave owns a CONTIGUOUS range of chunks (a few output rows of one image): the three
    // input rows of an output row are then re-read by the same wave a few chunks later, out of
    // its CU's cache, instead of by three workgroups on three XCDs (measured before this:
    // 442 MB fetched for a 133 MB batch of images).
    const int nwaves = gridDim.x * 4;
    const int wave_global = xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int per = (nchunk + nwaves - 1) / nwaves;
    const int first = wave_global * per;
    const int last = min(first + per, nchunk);

    // this lane's three pixels (tap row g, columns ow-1 .. ow+1) of each 16-pixel block f
    auto issue = [&](int chunk, u3 (&p)[4][3]) {
        const bool live = chunk < nchunk;
        const int rowid = live ? chunk / chunks_per_row : 0;    // b * OH + oh
        const int ow0 = (chunk - rowid * chunks_per_row) * 64;
        const int b = rowid / a.OH, oh = rowid - b * a.OH;
        const int ih = oh + g - 1;
        const bool row_ok = live && g < 3 && (unsigned)ih < (unsigned)a.H;
        const int rowoff = ((b * a.H + ih) * a.W + ow0 - 1) * 3;      // (b, ih, ow0 - 1, 0) of the dense fp32 image
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const unsigned off = row_ok ? (unsigned)(rowoff + (f * 16 + li) * 3) * 4u : 0x80000000u;
            p[f][0] = __builtin_amdgcn_raw_buffer_load_b96(xr, off, 0, 0);
            p[f][1] = __builtin_amdgcn_raw_buffer_load_b96(xr, off + 12u, 0, 0);
            p[f][2] = __builtin_amdgcn_raw_buffer_load_b96(xr, off + 24u, 0, 0);
        }
    };
    auto compute = [&](int chunk, const u3 (&p)[4][3]) {
        const int rowid = chunk / chunks_per_row;
        const int ow0 = (chunk - rowid * chunks_per_row) * 64;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int ow = ow0 + f * 16 + li;
            const bool l_ok = ow >= 1, r_ok = ow + 1 < a.W;
            bf16x8_first_t b1, b2;
            b1[0] = (__bf16)(l_ok ? __uint_as_float(p[f][0][0]) : 0.0f);
            b1[1] = (__bf16)(l_ok ? __uint_as_float(p[f][0][1]) : 0.0f);
            b1[2] = (__bf16)(l_ok ? __uint_as_float(p[f][0][2]) : 0.0f);
            b1[3] = (__bf16)__uint_as_float(p[f][1][0]);
            b1[4] = (__bf16)__uint_as_float(p[f][1][1]);
            b1[5] = (__bf16)__uint_as_float(p[f][1][2]);
            b1[6] = (__bf16)(r_ok ? __uint_as_float(p[f][2][0]) : 0.0f);
            b1[7] = (__bf16)(r_ok ? __uint_as_float(p[f][2][1]) : 0.0f);
#pragma unroll
            for (int j = 1; j < 8; ++j) b2[j] = (__bf16)0.0f;
            b2[0] = (__bf16)(r_ok ? __uint_as_float(p[f][2][2]) : 0.0f);
            f32x4_first_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0], b1, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[1], b1, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0], b2, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[1], b2, d1, 0, 0, 0);
            if (ow < a.OW) {
                // two values per step: the activation's multiplies and adds are packed fp32
                // instructions (this layer is bound by its 354 M Mish evaluations, not by memory)
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2_t o0 = affine_act2<ACT>(f32x2_t{d0[r], d0[r + 1]}, f32x2_t{sc[r], sc[r + 1]}, f32x2_t{sh[r], sh[r + 1]});
                    const f32x2_t o1 = affine_act2<ACT>(f32x2_t{d1[r], d1[r + 1]}, f32x2_t{sc[4 + r], sc[5 + r]}, f32x2_t{sh[4 + r], sh[5 + r]});
                    v[r] = o0.x; v[r + 1] = o0.y;
                    v[4 + r] = o1.x; v[5 + r] = o1.y;
                }
                store_stream16(y + ((size_t)rowid * a.OW + ow) * a.y_ld + 8 * g, pack2_bf16(v[0], v[1]),
                               pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
            }
        }
    };

    u3 pa[4][3], pb[4][3];
    if (first >= last) return;
    issue(first, pa);
    for (int chunk = first; chunk < last; chunk += 2) {
        issue(chunk + 1 < last ? chunk + 1 : nchunk, pb);
        compute(chunk, pa);
        if (chunk + 1 >= last) break;
        issue(chunk + 2 < last ? chunk + 2 : nchunk, pa);
        compute(chunk + 1, pb);
    }
}

static bool first_layer_shape(const ConvArgs &a) {
    return a.k == 3 && a.Cin == 3 && a.Cout == 32 && a.stride == 1 && a.dtype == YMX_BF16 &&
           a.out_dtype == YMX_BF16 && a.res == nullptr && (a.y_ld & 7) == 0;
}

// The network's first layer on a dense fp32 image: the row-run kernel above.
bool conv_first_rows_eligible(const ConvArgs &a) {
    return first_layer_shape(a) && a.x_dtype == YMX_F32 && a.x_ld == 3 && a.x_bytes != 0;
}

int launch_conv_direct(const ConvArgs &a, hipStream_t s) {
    if (conv_first_rows_eligible(a)) {
        const int cpr = (a.OW + 63) / 64;
        const int nchunk = a.B * a.OH * cpr;
        const int blocks = (nchunk + 3) / 4 < 256 * 8 ? (nchunk + 3) / 4 : 256 * 8;
        if (a.act == YMX_ACT_MISH)
            hipLaunchKernelGGL((conv_first_rows_kernel<YMX_ACT_MISH>), dim3(blocks), dim3(256), 0, s, a, cpr, nchunk);
        else if (a.act == YMX_ACT_LEAKY)
            hipLaunchKernelGGL((conv_first_rows_kernel<YMX_ACT_LEAKY>), dim3(blocks), dim3(256), 0, s, a, cpr, nchunk);
        else
            hipLaunchKernelGGL((conv_first_rows_kernel<YMX_ACT_NONE>), dim3(blocks), dim3(256), 0, s, a, cpr, nchunk);
        return check_launch("conv_first_rows");
    }
    if (first_layer_shape(a)) {
        const int nchunk = (a.P + 63) / 64;
        const int blocks = (nchunk + 3) / 4 < 256 * 8 ? (nchunk + 3) / 4 : 256 * 8;
        if (a.x_dtype == YMX_F32)
            hipLaunchKernelGGL((conv_first_mfma_kernel<float>), dim3(blocks), dim3(256), 0, s, a, nchunk);
        else
            hipLaunchKernelGGL((conv_first_mfma_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, a, nchunk);
        return check_launch("conv_first_mfma");
    }
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
// src_cout > 0 packs the kernel of the input-gradient convolution instead: w is the forward
// kernel [taps][Cout (its input channels)][src_cout (its output channels, <= Cin)], and the
// packed kernel is that one flipped over the taps with the two channel axes swapped (rows of
// zeros where Cin pads src_cout).
struct PackGeom {
    int K, Cin, Cout, Kw, rows, CB, kb, src_cout, ntaps;
    int s2_c;       // != 0: the 2x2 depth-to-space form of a stride-2 layer's input gradient; its input channels
};
__device__ __forceinline__ float pack_value(const float *__restrict__ w, const PackGeom &g, size_t i) {
    const int co = (int)(i / g.Kw), kk = (int)(i - (size_t)co * g.Kw);
    // packed column kk = blk*kb + tap*CB + c  <-  HWIO row tap*Cin + blk*CB + c
    const int blk = kk / g.kb, rem = kk - blk * g.kb, tap = rem / g.CB, c = rem - tap * g.CB;
    float v = 0.0f;
    if (g.s2_c) {
        // row co = (q, ci): output parity q = 2 ph + pw and input channel ci of the forward layer; column = (tap
        // (dr, dc) of the 2x2 window over dz, dz channel cz).  Input row 2 oh' + ph meets output row oh' + dr
        // through kernel row r: ph 0: (dr 0, r 1); ph 1: (dr 0, r 2), (dr 1, r 0); columns likewise.
        if (co < g.Cout && kk < g.K) {
            const int q = co / g.s2_c, ci = co - q * g.s2_c, cz = blk * g.CB + c;
            const int dr = tap >> 1, dc = tap & 1, ph = q >> 1, pw = q & 1;
            const int r = ph ? (dr ? 0 : 2) : (dr ? -1 : 1), s = pw ? (dc ? 0 : 2) : (dc ? -1 : 1);
            if (r >= 0 && s >= 0 && cz < g.src_cout) v = w[((size_t)(r * 3 + s) * g.s2_c + ci) * g.src_cout + cz];
        }
        return v;
    }
    if (co < g.Cout && kk < g.K) {
        const int ci = blk * g.CB + c;
        if (g.src_cout == 0) v = w[(size_t)(tap * g.Cin + ci) * g.Cout + co];
        else if (ci < g.src_cout) v = w[((size_t)(g.ntaps - 1 - tap) * g.Cout + co) * g.src_cout + ci];
    }
    return v;
}
template <typename T>
__global__ void __launch_bounds__(256)
pack_weights_kernel(const float *__restrict__ w, T *__restrict__ out, PackGeom g) {
    const size_t n = (size_t)g.rows * g.Kw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        elem<T>::store(out + i, pack_value(w, g, i));
}
// The kernel of a layer packed both ways in one launch (what a training step needs after every
// update of the weights): `fwd` for the layer itself, `bwd` for its input-gradient convolution.
template <typename T>
__global__ void __launch_bounds__(256)
pack_weights_pair_kernel(const float *__restrict__ w, T *__restrict__ fwd, T *__restrict__ bwd, PackGeom gf, PackGeom gb) {
    const size_t nf = (size_t)gf.rows * gf.Kw, n = nf + (size_t)gb.rows * gb.Kw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (i < nf) elem<T>::store(fwd + i, pack_value(w, gf, i));
        else elem<T>::store(bwd + (i - nf), pack_value(w, gb, i - nf));
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

static PackGeom pack_geom(int k, int Cin, int Cout, int src_cout) {
    PackGeom g;
    g.K = k * k * Cin; g.Cin = Cin; g.Cout = Cout; g.Kw = packed_kw(k, Cin); g.rows = packed_rows(Cout);
    g.CB = conv_channel_block(Cin); g.kb = k * k * g.CB; g.src_cout = src_cout; g.ntaps = k * k; g.s2_c = 0;
    return g;
}
// input gradient of a 3x3 stride-2 layer (C input, Cout output channels) as a 2x2 convolution over dz
// (dz_channels >= Cout) with 4 C depth-to-space output channels
static PackGeom pack_geom_s2(int C, int Cout, int dz_channels) {
    PackGeom g = pack_geom(2, dz_channels, 4 * C, Cout);
    g.s2_c = C;
    return g;
}

int launch_pack_weights(const float *w_hwio, void *w_packed, int k, int Cin, int Cout, int dtype,
                        hipStream_t s, int src_cout) {
    const int K = k * k * Cin;
    if (!conv_uses_mfma(Cin, dtype)) {
        if (src_cout) return fail(YMX_EUNSUPPORTED, "pack_weights_dgrad: %d channels in dtype %d take the direct kernel", Cin, dtype);
        const size_t n = (size_t)K * Cout;
        const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(copy_weights_kernel, dim3(grid), dim3(256), 0, s, w_hwio, (float *)w_packed, n,
                           dtype == YMX_BF16 ? 1 : 0);
        return check_launch("copy_weights");
    }
    const PackGeom g = pack_geom(k, Cin, Cout, src_cout);
    const size_t n = (size_t)g.rows * g.Kw;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == YMX_BF16)
        hipLaunchKernelGGL((pack_weights_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, w_hwio, (bf16_t *)w_packed, g);
    else
        hipLaunchKernelGGL((pack_weights_kernel<float>), dim3(grid), dim3(256), 0, s, w_hwio, (float *)w_packed, g);
    return check_launch("pack_weights");
}

// forward pack of a k x k x Cin x Cout kernel and the pack of its input-gradient convolution
// (dz_channels >= Cout input channels, Cin output channels) in one launch; MFMA layouts only
int launch_pack_weights_dgrad_s2(const float *w_hwio, void *w_packed, int Cin, int Cout, int dz_channels, int dtype, hipStream_t s) {
    if (!conv_uses_mfma(dz_channels, dtype)) return fail(YMX_EUNSUPPORTED, "pack_weights_dgrad_s2: %d channels in dtype %d", dz_channels, dtype);
    const PackGeom g = pack_geom_s2(Cin, Cout, dz_channels);
    const size_t n = (size_t)g.rows * g.Kw;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == YMX_BF16)
        hipLaunchKernelGGL((pack_weights_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, w_hwio, (bf16_t *)w_packed, g);
    else
        hipLaunchKernelGGL((pack_weights_kernel<float>), dim3(grid), dim3(256), 0, s, w_hwio, (float *)w_packed, g);
    return check_launch("pack_weights_dgrad_s2");
}

int launch_pack_weights_pair(const float *w_hwio, void *w_fwd, void *w_bwd, int k, int Cin, int Cout, int dz_channels, int stride,
                             int dtype, hipStream_t s) {
    if (!conv_uses_mfma(Cin, dtype) || !conv_uses_mfma(dz_channels, dtype))
        return fail(YMX_EUNSUPPORTED, "pack_weights_pair: %d / %d channels in dtype %d take the direct kernel", Cin, dz_channels, dtype);
    const PackGeom gf = pack_geom(k, Cin, Cout, 0);
    const PackGeom gb = stride == 2 ? pack_geom_s2(Cin, Cout, dz_channels) : pack_geom(k, dz_channels, Cin, Cout);
    const size_t n = (size_t)gf.rows * gf.Kw + (size_t)gb.rows * gb.Kw;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (dtype == YMX_BF16)
        hipLaunchKernelGGL((pack_weights_pair_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, w_hwio, (bf16_t *)w_fwd, (bf16_t *)w_bwd, gf, gb);
    else
        hipLaunchKernelGGL((pack_weights_pair_kernel<float>), dim3(grid), dim3(256), 0, s, w_hwio, (float *)w_fwd, (float *)w_bwd, gf, gb);
    return check_launch("pack_weights_pair");
}

}  // namespace ymx
