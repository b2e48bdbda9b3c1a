// Kernels of the rows beyond plain detector inference:
//   - SPP block of CSPDarknet53 (three stride-1 SAME max-pools of 13, 9 and 5 fused with the
//     channel concat that follows them, /root/reference/backbones.py:254-259);
//   - the YOLO loss of one scale, forward (/root/reference/losses.py:10-46 with the decode of
//     utils.py:107-131 fused in), without materialising the [B,S,S,3,128] IoU tensor;
//   - label assignment for a batch (/root/reference/dataset.py:387-482).
#include <algorithm>
#include <cstring>

#include "common.h"

namespace ymx {

// ---- SPP: y = concat(maxpool13(x), maxpool9(x), maxpool5(x), x) on the channel axis -------
// max over a (2r+1)^2 window = max over rows of max over columns, and a 9-window is a 5-window
// of 5-windows (13 likewise), so all three pools are three rounds of one horizontal and one
// vertical 5-max over an LDS copy of the map.  A workgroup holds the whole H x W map of one
// image for CH 16-byte channel chunks; threads map chunk-fastest, so the CH chunks of a pixel
// are stored as one contiguous run.  Pixels outside the map do not take part (TensorFlow pads
// SAME max-pools with -inf).
template <typename T>
__device__ __forceinline__ uint4 max16(const uint4 a, const uint4 b) {
    uint4 o;
    if constexpr (sizeof(T) == 4) {
        o.x = __float_as_uint(fmaxf(__uint_as_float(a.x), __uint_as_float(b.x)));
        o.y = __float_as_uint(fmaxf(__uint_as_float(a.y), __uint_as_float(b.y)));
        o.z = __float_as_uint(fmaxf(__uint_as_float(a.z), __uint_as_float(b.z)));
        o.w = __float_as_uint(fmaxf(__uint_as_float(a.w), __uint_as_float(b.w)));
    } else {
        auto m2 = [](uint32_t p, uint32_t r) {
            const float lo = fmaxf(__uint_as_float(p << 16), __uint_as_float(r << 16));
            const float hi = fmaxf(__uint_as_float(p & 0xffff0000u), __uint_as_float(r & 0xffff0000u));
            return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
        };
        o.x = m2(a.x, b.x); o.y = m2(a.y, b.y); o.z = m2(a.z, b.z); o.w = m2(a.w, b.w);
    }
    return o;
}

template <typename T, int CH>
__global__ void __launch_bounds__(256)
spp_concat_kernel(const T *__restrict__ x, T *__restrict__ y, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int KCH = 16 / (int)sizeof(T);
    const int npix = H * W;
    uint4 *bufA = (uint4 *)smem_raw;                 // [npix][CH]
    uint4 *bufB = bufA + (size_t)npix * CH;          // [npix][CH]
    const int t = threadIdx.x;
    const int ch = t % CH, p0 = t / CH;
    constexpr int PSTEP = 256 / CH;
    const int b = blockIdx.y;
    const int c0 = (blockIdx.x * CH + ch) * KCH;     // first channel of this thread's chunk
    const bool chunk_ok = c0 < C;
    const T *xb = x + (size_t)b * npix * C + c0;
    T *yb = y + (size_t)b * npix * 4 * C + c0;

    for (int p = p0; p < npix; p += PSTEP) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (chunk_ok) {
            v = *(const uint4 *)(xb + (size_t)p * C);
            *(uint4 *)(yb + (size_t)p * 4 * C + 3 * C) = v;
        }
        bufA[p * CH + ch] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int round = 0; round < 3; ++round) {
        for (int p = p0; p < npix; p += PSTEP) {        // horizontal 5-max: A -> B
            const int r = p / W, c = p - r * W;
            uint4 m = bufA[p * CH + ch];
#pragma unroll
            for (int d = -2; d <= 2; ++d)
                if (d != 0 && (unsigned)(c + d) < (unsigned)W) m = max16<T>(m, bufA[(p + d) * CH + ch]);
            bufB[p * CH + ch] = m;
        }
        __syncthreads();
        for (int p = p0; p < npix; p += PSTEP) {        // vertical 5-max: B -> A, and out
            const int r = p / W;
            uint4 m = bufB[p * CH + ch];
#pragma unroll
            for (int d = -2; d <= 2; ++d)
                if (d != 0 && (unsigned)(r + d) < (unsigned)H) m = max16<T>(m, bufB[(p + d * W) * CH + ch]);
            bufA[p * CH + ch] = m;
            if (chunk_ok) *(uint4 *)(yb + (size_t)p * 4 * C + (2 - round) * C) = m;   // 5 -> slot 2, 9 -> 1, 13 -> 0
        }
        __syncthreads();
    }
}

int launch_spp_concat(const void *x, void *y, int B, int H, int W, int C, int dtype, hipStream_t s) {
    const int kch = dtype == YMX_BF16 ? 8 : 4;
    const int nchunk = C / kch;
    const size_t per_chunk = (size_t)H * W * 16 * 2;          // both LDS copies of one chunk
    int ch = 8;
    while (ch > 1 && (per_chunk * ch > 64 * 1024 || ch > nchunk)) ch >>= 1;
    if (per_chunk * ch > 64 * 1024) return fail(YMX_EUNSUPPORTED, "spp_concat: map of %d x %d pixels too large (at most 2048 pixels)", H, W);
    const size_t lds = per_chunk * ch;
    dim3 grid((nchunk + ch - 1) / ch, B), block(256);
#define YMX_SPP(T, CHV) hipLaunchKernelGGL((spp_concat_kernel<T, CHV>), grid, block, lds, s, (const T *)x, (T *)y, H, W, C)
    if (dtype == YMX_BF16) {
        if (ch == 8) YMX_SPP(bf16_t, 8); else if (ch == 4) YMX_SPP(bf16_t, 4); else if (ch == 2) YMX_SPP(bf16_t, 2); else YMX_SPP(bf16_t, 1);
    } else {
        if (ch == 8) YMX_SPP(float, 8); else if (ch == 4) YMX_SPP(float, 4); else if (ch == 2) YMX_SPP(float, 2); else YMX_SPP(float, 1);
    }
#undef YMX_SPP
    return check_launch("spp_concat");
}

// ---- SPP block with its derivative ----------------------------------------------------------------
// Training form of spp_concat (backbones.py:254-259): the same outputs plus, for every pooled value,
// where in its window it came from (row-major position inside the k x k window, the first maximum in
// scan order), so that the backward pass hands each pooled gradient to exactly one input element:
//   dx[p] = dy_x[p] + sum over k in {13, 9, 5}, over the outputs o whose window holds p:
//           dy_k[o] if arg_k[o] == position of p in o's window.
// The backward kernel gathers (one thread per input element walks the outputs around it), so no
// atomics are needed and the sum has a fixed order.  One thread per (pixel, channel).
template <typename T>
__global__ void __launch_bounds__(256)
spp_argmax_kernel(const T *__restrict__ x, T *__restrict__ y, unsigned char *__restrict__ arg, int B, int H, int W, int C) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        size_t r = idx / C;
        const int j = (int)(r % W);
        r /= W;
        const int i = (int)(r % H), b = (int)(r / H);
        const size_t pix = ((size_t)b * H + i) * W + j;
        const T *xb = x + (size_t)b * H * W * C + c;
#pragma unroll
        for (int pool = 0; pool < 3; ++pool) {
            const int k = pool == 0 ? 13 : (pool == 1 ? 9 : 5), rad = k / 2;
            float best = -INFINITY;
            int where = 0;
            for (int di = -rad; di <= rad; ++di) {
                const int ii = i + di;
                if (ii < 0 || ii >= H) continue;
                for (int dj = -rad; dj <= rad; ++dj) {
                    const int jj = j + dj;
                    if (jj < 0 || jj >= W) continue;
                    const float v = elem<T>::load(xb + ((size_t)ii * W + jj) * C);
                    if (v > best) { best = v; where = (di + rad) * k + dj + rad; }
                }
            }
            elem<T>::store(y + pix * 4 * C + pool * C + c, best);
            arg[(pix * 3 + pool) * C + c] = (unsigned char)where;
        }
        elem<T>::store(y + pix * 4 * C + 3 * C + c, elem<T>::load(x + idx));
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
spp_bwd_kernel(const T *__restrict__ dy, const unsigned char *__restrict__ arg, T *__restrict__ dx, int B, int H, int W, int C) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        size_t r = idx / C;
        const int j = (int)(r % W);
        r /= W;
        const int i = (int)(r % H), b = (int)(r / H);
        const size_t pix = ((size_t)b * H + i) * W + j;
        float acc = elem<T>::load(dy + pix * 4 * C + 3 * C + c);
#pragma unroll
        for (int pool = 0; pool < 3; ++pool) {
            const int k = pool == 0 ? 13 : (pool == 1 ? 9 : 5), rad = k / 2;
            for (int oi = max(i - rad, 0); oi <= min(i + rad, H - 1); ++oi)
                for (int oj = max(j - rad, 0); oj <= min(j + rad, W - 1); ++oj) {
                    const size_t op = ((size_t)b * H + oi) * W + oj;
                    const int mine = (i - oi + rad) * k + (j - oj + rad);
                    if (arg[(op * 3 + pool) * C + c] == mine) acc += elem<T>::load(dy + op * 4 * C + pool * C + c);
                }
        }
        elem<T>::store(dx + idx, acc);
    }
}

int launch_spp_train(const void *x, void *y, unsigned char *arg, int B, int H, int W, int C, int dtype, hipStream_t s) {
    const size_t total = (size_t)B * H * W * C;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 1 << 16);
    if (dtype == YMX_BF16) hipLaunchKernelGGL(spp_argmax_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)x, (bf16_t *)y, arg, B, H, W, C);
    else hipLaunchKernelGGL(spp_argmax_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)x, (float *)y, arg, B, H, W, C);
    return check_launch("spp_concat_train");
}
int launch_spp_bwd(const void *dy, const unsigned char *arg, void *dx, int B, int H, int W, int C, int dtype, hipStream_t s) {
    const size_t total = (size_t)B * H * W * C;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 1 << 16);
    if (dtype == YMX_BF16) hipLaunchKernelGGL(spp_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)dy, arg, (bf16_t *)dx, B, H, W, C);
    else hipLaunchKernelGGL(spp_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)dy, arg, (float *)dx, B, H, W, C);
    return check_launch("spp_concat_backward");
}

// ---- YOLO loss of one scale, forward ---------------------------------------------------------
// centre-format IoU and GIoU exactly as /root/reference/utils.py:33-105 (divide_no_nan: 0 when
// the divisor is 0)
__device__ __forceinline__ float giou_cxcywh(float ax, float ay, float aw, float ah, float bx, float by, float bw, float bh) {
    const float area1 = aw * ah, area2 = bw * bh;
    const float ax0 = ax - aw * 0.5f, ay0 = ay - ah * 0.5f, ax1 = ax + aw * 0.5f, ay1 = ay + ah * 0.5f;
    const float bx0 = bx - bw * 0.5f, by0 = by - bh * 0.5f, bx1 = bx + bw * 0.5f, by1 = by + bh * 0.5f;
    const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.0f);
    const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.0f);
    const float inter = iw * ih;
    const float uni = area1 + area2 - inter;
    const float iou = uni != 0.0f ? inter / uni : 0.0f;
    const float ew = fmaxf(ax1, bx1) - fminf(ax0, bx0), eh = fmaxf(ay1, by1) - fminf(ay0, by0);
    const float earea = ew * eh;
    return iou - (earea != 0.0f ? (earea - uni) / earea : 0.0f);
}

// tf.nn.sigmoid_cross_entropy_with_logits: max(x, 0) - x z + log(1 + exp(-|x|))
__device__ __forceinline__ float sce_logits(float z, float x) {
    return fmaxf(x, 0.0f) - x * z + __logf(1.0f + __expf(-fabsf(x)));
}

// One wavefront per 64 candidates (row, col, anchor order) of one image, four per workgroup.
// A lane reads only the five leading values of its candidate's `conv` and `label` rows; the
// class parts of the two rows (80 of 85 values at 80 classes) are read -- by the whole wave,
// coalesced -- only for candidates whose `respond` label is non-zero, since the reference
// multiplies the class cross-entropy by it: a map with few objects is not read in full.
// Wave 0 first compacts the image's ground-truth boxes into LDS (all-zero rows of the
// 128-slot buffer have IoU 0 with everything and cannot change the maximum, which starts at
// 0 as IoU is never negative).  The background mask needs only "max IoU < threshold",
// evaluated as inter < threshold * union per box: no division.  Each wave reduces its three
// loss terms over its lanes into one partial (summed per image in a fixed order by the
// second kernel: no float atomics, the result is reproducible).
struct LossArgs {
    const float *conv, *pred, *label, *bboxes;
    float *partial;              // [B][nseg][3]
    int B, S, C, NB, nseg;
    int conv_ld;                 // elements from one grid cell to the next in conv (>= 3*(5+C))
    long long label_stride, bbox_stride;   // elements from one image to the next
    float stride, xyscale, anchor[3][2];
    float input_size, iou_thr;
};

__global__ void __launch_bounds__(256)
yolo_loss_kernel(const LossArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *sbox = (float *)smem_raw;                 // [NB][4] compacted
    int *scnt = (int *)(sbox + 4 * a.NB);
    const int D = 5 + a.C;
    const int t = threadIdx.x, lane = t & 63;
    const int q = __builtin_amdgcn_readfirstlane(t >> 6);
    const int seg = blockIdx.x * 4 + q, b = blockIdx.y;
    const int Ns = 3 * a.S * a.S;
    const int c0 = seg * 64;
    const int cme = c0 + lane, cellm = cme / 3, an = cme - cellm * 3;
    const bool live = cme < Ns;
    // the candidate's own values first: the loads are in flight while wave 0 compacts the boxes
    const float *row = a.conv + ((size_t)b * a.S * a.S + cellm) * a.conv_ld + an * D;
    const float *lab = a.label + (size_t)b * a.label_stride + (size_t)cme * D;
    float r[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, l[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, pr[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
        for (int i = 0; i < 5; ++i) { r[i] = row[i]; l[i] = lab[i]; }
        if (a.pred) {
            const float *pp = a.pred + ((size_t)b * Ns + cme) * D;
#pragma unroll
            for (int i = 0; i < 5; ++i) pr[i] = pp[i];
        }
    }
    if (q == 0) {
        int n = 0;
        for (int j0 = 0; j0 < a.NB; j0 += 64) {
            const int j = j0 + lane;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < a.NB) {
                const float *g = a.bboxes + (size_t)b * a.bbox_stride + (size_t)j * 4;
                v = make_float4(g[0], g[1], g[2], g[3]);
            }
            const bool nz = (v.x != 0.0f) | (v.y != 0.0f) | (v.z != 0.0f) | (v.w != 0.0f);
            const unsigned long long m = __ballot(nz);
            if (nz) *(float4 *)(sbox + 4 * (n + __popcll(m & ((1ull << lane) - 1ull)))) = v;
            n += __popcll(m);
        }
        if (lane == 0) *scnt = n;
    }
    __syncthreads();
    if (seg >= a.nseg) return;
    const int nbox = *scnt;

    float px, py, pw, ph, pconf;
    if (a.pred) {
        px = pr[0]; py = pr[1]; pw = pr[2]; ph = pr[3]; pconf = pr[4];
    } else {
        const int col = cellm % a.S, rw = cellm / a.S;
        const float half = 0.5f * (a.xyscale - 1.0f);
        px = ((1.0f / (1.0f + __expf(-r[0]))) * a.xyscale - half + (float)col) * a.stride;
        py = ((1.0f / (1.0f + __expf(-r[1]))) * a.xyscale - half + (float)rw) * a.stride;
        pw = __expf(r[2]) * a.anchor[an][0];
        ph = __expf(r[3]) * a.anchor[an][1];
        pconf = 1.0f / (1.0f + __expf(-r[4]));
    }
    // class cross-entropy of the responsible candidates, one at a time by the whole wave
    float prob_sum = 0.f;
    unsigned long long m = __ballot(live && l[4] != 0.0f);
    while (m) {
        const int who = __builtin_ctzll(m);
        m &= m - 1;
        const int cw = c0 + who, cellw = cw / 3, aw = cw - cellw * 3;
        const float *rr = a.conv + ((size_t)b * a.S * a.S + cellw) * a.conv_ld + aw * D + 5;
        const float *ll = a.label + (size_t)b * a.label_stride + (size_t)cw * D + 5;
        float s = 0.f;
        for (int c = lane; c < a.C; c += 64) s += sce_logits(ll[c], rr[c]);
#pragma unroll
        for (int dl = 32; dl > 0; dl >>= 1) s += __shfl_xor(s, dl);
        if (lane == who) prob_sum = s;
    }
    // background test against the ground-truth boxes
    bool hit = false;
    {
        const float area1 = pw * ph;
        const float ax0 = px - pw * 0.5f, ay0 = py - ph * 0.5f, ax1 = px + pw * 0.5f, ay1 = py + ph * 0.5f;
        for (int j = 0; j < nbox; ++j) {
            const float4 g = *(const float4 *)(sbox + 4 * j);           // same address in every lane: a broadcast
            const float bx0 = g.x - g.z * 0.5f, by0 = g.y - g.w * 0.5f, bx1 = g.x + g.z * 0.5f, by1 = g.y + g.w * 0.5f;
            const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.0f);
            const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.0f);
            const float inter = iw * ih;
            const float uni = area1 + g.z * g.w - inter;
            hit |= (uni != 0.0f) & !(inter < a.iou_thr * uni);
        }
    }
    float lg = 0.f, lc = 0.f, lp = 0.f;
    if (live) {
        const float respond = l[4];
        const float giou = giou_cxcywh(px, py, pw, ph, l[0], l[1], l[2], l[3]);
        const float scale = 2.0f - 1.0f * l[2] * l[3] / (a.input_size * a.input_size);
        lg = respond * scale * (1.0f - giou);
        const float bgd = (1.0f - respond) * (hit ? 0.0f : 1.0f);
        const float d = respond - pconf;
        const float bce = sce_logits(respond, r[4]);
        lc = (d * d) * (respond * bce + bgd * bce);
        lp = respond * prob_sum;
    }
#pragma unroll
    for (int dl = 32; dl > 0; dl >>= 1) {
        lg += __shfl_xor(lg, dl);
        lc += __shfl_xor(lc, dl);
        lp += __shfl_xor(lp, dl);
    }
    if (lane == 0) {
        float *o = a.partial + ((size_t)b * a.nseg + seg) * 3;
        o[0] = lg; o[1] = lc; o[2] = lp;
    }
}

// Sums the workgroup partials of every image (16 lanes per image, each a strided share in
// segment order, then a fixed shuffle tree), then averages over the batch in image order:
// per_image [B][3] (may be null) and out [3] = (giou, conf, prob).
__global__ void __launch_bounds__(256)
yolo_loss_reduce_kernel(const float *__restrict__ partial, float *__restrict__ per_image, float *__restrict__ out,
                        float *__restrict__ scratch, int B, int nseg) {
    const int t = threadIdx.x, sub = t & 15;
    for (int b0 = 0; b0 < B; b0 += 16) {
        const int b = b0 + (t >> 4);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (b < B) {
            const float *p = partial + (size_t)b * nseg * 3;
            for (int i = sub; i < nseg; i += 16) { s0 += p[3 * i]; s1 += p[3 * i + 1]; s2 += p[3 * i + 2]; }
        }
#pragma unroll
        for (int dl = 8; dl > 0; dl >>= 1) {
            s0 += __shfl_xor(s0, dl);
            s1 += __shfl_xor(s1, dl);
            s2 += __shfl_xor(s2, dl);
        }
        if (b < B && sub == 0) {
            scratch[3 * b] = s0; scratch[3 * b + 1] = s1; scratch[3 * b + 2] = s2;
            if (per_image) { per_image[3 * b] = s0; per_image[3 * b + 1] = s1; per_image[3 * b + 2] = s2; }
        }
    }
    __threadfence_block();
    __syncthreads();
    if (t < 3) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += scratch[3 * b + t];
        out[t] = s / (float)B;
    }
}

size_t yolo_loss_workspace_bytes(int B, int S) {
    if (B <= 0 || S <= 0) return 0;
    const size_t nseg = ((size_t)3 * S * S + 63) / 64;
    return ((size_t)B * nseg * 3 + (size_t)B * 3) * 4;
}

int launch_yolo_loss(const float *conv, int conv_ld, const float *pred, const float *label, long long label_stride,
                     const float *bboxes, long long bbox_stride, int B, int S, int C, int NB, float stride, float xyscale, const float *anchors, float iou_thr, float *out,
                     float *per_image, void *workspace, hipStream_t s) {
    LossArgs a;
    a.conv = conv; a.pred = pred; a.label = label; a.bboxes = bboxes;
    a.partial = (float *)workspace;
    a.B = B; a.S = S; a.C = C; a.NB = NB;
    a.nseg = (3 * S * S + 63) / 64;
    a.conv_ld = conv_ld > 0 ? conv_ld : 3 * (5 + C);
    a.label_stride = label_stride > 0 ? label_stride : (long long)3 * S * S * (5 + C);
    a.bbox_stride = bbox_stride > 0 ? bbox_stride : (long long)4 * NB;
    a.stride = stride; a.xyscale = xyscale;
    for (int i = 0; i < 6; ++i) (&a.anchor[0][0])[i] = anchors[i];
    a.input_size = (float)S * stride;
    a.iou_thr = iou_thr;
    const size_t lds = (size_t)NB * 16 + 16;
    if (lds > 64 * 1024) return fail(YMX_EUNSUPPORTED, "yolo_loss: %d ground-truth slots per image need %zu bytes of LDS", NB, lds);
    hipLaunchKernelGGL(yolo_loss_kernel, dim3((a.nseg + 3) / 4, B), dim3(256), lds, s, a);
    int rc = check_launch("yolo_loss");
    if (rc) return rc;
    float *scratch = a.partial + (size_t)B * a.nseg * 3;
    hipLaunchKernelGGL(yolo_loss_reduce_kernel, dim3(1), dim3(256), 0, s, a.partial, per_image, out, scratch, B, a.nseg);
    return check_launch("yolo_loss_reduce");
}


// ---- YOLO loss of one scale, gradient with respect to the raw head map ------------------------
// Value and its four partial derivatives with respect to the predicted box (cx, cy, w, h):
// forward-mode differentiation of the GIoU expression of /root/reference/utils.py:65-105,
// so that the gradient follows that text operation by operation.  A min / max passes the
// derivative of the operand it selects (the first one on a tie, as TensorFlow's gradient
// does); divide_no_nan is 0 with zero derivative where the divisor is 0.
struct Dual4 {
    float v, d[4];
};
__device__ __forceinline__ Dual4 dconst(float v) { return Dual4{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual4 dvar(float v, int k) {
    Dual4 r = dconst(v);
    r.d[k] = 1.0f;
    return r;
}
__device__ __forceinline__ Dual4 operator+(const Dual4 &a, const Dual4 &b) {
    return Dual4{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2], a.d[3] + b.d[3]}};
}
__device__ __forceinline__ Dual4 operator-(const Dual4 &a, const Dual4 &b) {
    return Dual4{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2], a.d[3] - b.d[3]}};
}
__device__ __forceinline__ Dual4 operator*(const Dual4 &a, const Dual4 &b) {
    return Dual4{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2],
                             a.d[3] * b.v + a.v * b.d[3]}};
}
__device__ __forceinline__ Dual4 operator*(const Dual4 &a, float s) {
    return Dual4{a.v * s, {a.d[0] * s, a.d[1] * s, a.d[2] * s, a.d[3] * s}};
}
__device__ __forceinline__ Dual4 dmax(const Dual4 &a, const Dual4 &b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ Dual4 dmin(const Dual4 &a, const Dual4 &b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ Dual4 ddiv_no_nan(const Dual4 &a, const Dual4 &b) {
    if (b.v == 0.0f) return dconst(0.0f);
    const float inv = 1.0f / b.v, q = a.v * inv;
    return Dual4{q, {(a.d[0] - q * b.d[0]) * inv, (a.d[1] - q * b.d[1]) * inv, (a.d[2] - q * b.d[2]) * inv,
                     (a.d[3] - q * b.d[3]) * inv}};
}

__device__ __forceinline__ Dual4 giou_dual(float px, float py, float pw, float ph, float lx, float ly, float lw, float lh) {
    const Dual4 X = dvar(px, 0), Y = dvar(py, 1), W = dvar(pw, 2), H = dvar(ph, 3);
    const Dual4 a1 = W * H, a2 = dconst(lw * lh);
    const Dual4 x0 = X - W * 0.5f, y0 = Y - H * 0.5f, x1 = X + W * 0.5f, y1 = Y + H * 0.5f;
    const Dual4 u0 = dconst(lx - lw * 0.5f), v0 = dconst(ly - lh * 0.5f), u1 = dconst(lx + lw * 0.5f), v1 = dconst(ly + lh * 0.5f);
    const Dual4 iw = dmax(dmin(x1, u1) - dmax(x0, u0), dconst(0.0f));
    const Dual4 ih = dmax(dmin(y1, v1) - dmax(y0, v0), dconst(0.0f));
    const Dual4 inter = iw * ih;
    const Dual4 uni = a1 + a2 - inter;
    const Dual4 iou = ddiv_no_nan(inter, uni);
    const Dual4 ew = dmax(x1, u1) - dmin(x0, u0), eh = dmax(y1, v1) - dmin(y0, v0);
    const Dual4 earea = ew * eh;
    return iou - ddiv_no_nan(earea - uni, earea);
}

// Same work split as the forward kernel (one wave per 64 candidates; ground-truth boxes
// compacted in LDS).  Each lane forms the five leading gradient values of its candidate
//   d giou_loss / d t_xywh = -respond * scale * d giou / d pred * d pred / d t
//   d conf_loss / d t_conf = (respond + background) * (2 (p - respond) p (1 - p) bce + (respond - p)^2 (p - respond))
// (the background mask is a comparison cast to float and carries no gradient), leaves them
// in LDS, and the wave then writes the 64 x (5+C) block of the gradient with coalesced
// stores: those five per row, respond * (sigmoid(t) - label) in the class columns of
// responsible rows, zero everywhere else.  Every value is scaled by the term's weight / B
// (the loss averages over the batch).
struct LossGradArgs {
    LossArgs f;
    float *dconv;                // [B][S][S][3][5+C] dense
    const float *w;              // device [3]: upstream gradient of (giou, conf, prob); null = all 1
};

__global__ void __launch_bounds__(256)
yolo_loss_grad_kernel(const LossGradArgs g) {
    const LossArgs &a = g.f;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *sbox = (float *)smem_raw;                 // [NB][4] compacted
    int *scnt = (int *)(sbox + 4 * a.NB);
    float *sg = (float *)(scnt + 4);                 // [4 waves][64][5]
    const int D = 5 + a.C;
    const int t = threadIdx.x, lane = t & 63;
    const int q = __builtin_amdgcn_readfirstlane(t >> 6);
    const int seg = blockIdx.x * 4 + q, b = blockIdx.y;
    const int Ns = 3 * a.S * a.S;
    const int c0 = seg * 64;
    const int cme = c0 + lane, cellm = cme / 3, an = cme - cellm * 3;
    const bool live = cme < Ns;
    const float *row = a.conv + ((size_t)b * a.S * a.S + cellm) * a.conv_ld + an * D;
    const float *lab = a.label + (size_t)b * a.label_stride + (size_t)cme * D;
    float r[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, l[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
        for (int i = 0; i < 5; ++i) { r[i] = row[i]; l[i] = lab[i]; }
    }
    if (q == 0) {
        int n = 0;
        for (int j0 = 0; j0 < a.NB; j0 += 64) {
            const int j = j0 + lane;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < a.NB) {
                const float *p = a.bboxes + (size_t)b * a.bbox_stride + (size_t)j * 4;
                v = make_float4(p[0], p[1], p[2], p[3]);
            }
            const bool nz = (v.x != 0.0f) | (v.y != 0.0f) | (v.z != 0.0f) | (v.w != 0.0f);
            const unsigned long long m = __ballot(nz);
            if (nz) *(float4 *)(sbox + 4 * (n + __popcll(m & ((1ull << lane) - 1ull)))) = v;
            n += __popcll(m);
        }
        if (lane == 0) *scnt = n;
    }
    __syncthreads();
    if (seg >= a.nseg) return;
    const int nbox = *scnt;
    const float invB = 1.0f / (float)a.B;
    const float w0 = g.w ? g.w[0] : 1.0f, w1 = g.w ? g.w[1] : 1.0f, w2 = g.w ? g.w[2] : 1.0f;

    const int col = cellm % a.S, rw = cellm / a.S;
    const float half = 0.5f * (a.xyscale - 1.0f);
    const float sx = 1.0f / (1.0f + __expf(-r[0])), sy = 1.0f / (1.0f + __expf(-r[1]));
    const float px = (sx * a.xyscale - half + (float)col) * a.stride;
    const float py = (sy * a.xyscale - half + (float)rw) * a.stride;
    const float pw = __expf(r[2]) * a.anchor[an][0];
    const float ph = __expf(r[3]) * a.anchor[an][1];
    const float pconf = 1.0f / (1.0f + __expf(-r[4]));
    const float respond = l[4];

    bool hit = false;
    {
        const float area1 = pw * ph;
        const float ax0 = px - pw * 0.5f, ay0 = py - ph * 0.5f, ax1 = px + pw * 0.5f, ay1 = py + ph * 0.5f;
        for (int j = 0; j < nbox; ++j) {
            const float4 gt = *(const float4 *)(sbox + 4 * j);
            const float bx0 = gt.x - gt.z * 0.5f, by0 = gt.y - gt.w * 0.5f, bx1 = gt.x + gt.z * 0.5f, by1 = gt.y + gt.w * 0.5f;
            const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.0f);
            const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.0f);
            const float inter = iw * ih;
            const float uni = area1 + gt.z * gt.w - inter;
            hit |= (uni != 0.0f) & !(inter < a.iou_thr * uni);
        }
    }
    float gout[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        if (respond != 0.0f) {
            const Dual4 gi = giou_dual(px, py, pw, ph, l[0], l[1], l[2], l[3]);
            const float scale = 2.0f - 1.0f * l[2] * l[3] / (a.input_size * a.input_size);
            const float k = -respond * scale * w0 * invB;
            gout[0] = k * gi.d[0] * (sx * (1.0f - sx) * a.xyscale * a.stride);
            gout[1] = k * gi.d[1] * (sy * (1.0f - sy) * a.xyscale * a.stride);
            gout[2] = k * gi.d[2] * pw;
            gout[3] = k * gi.d[3] * ph;
        }
        const float mask = respond + (1.0f - respond) * (hit ? 0.0f : 1.0f);
        const float bce = sce_logits(respond, r[4]);
        const float d = pconf - respond;
        gout[4] = w1 * invB * mask * (2.0f * d * pconf * (1.0f - pconf) * bce + d * d * d);
    }
    float *mine = sg + q * 320;
#pragma unroll
    for (int i = 0; i < 5; ++i) mine[lane * 5 + i] = gout[i];
    const unsigned long long resp = __ballot(live && respond != 0.0f);
    // the wave's LDS writes above are read by other lanes of the same wave only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int nvalid = Ns - c0;
    nvalid = nvalid > 64 ? 64 : nvalid;
    float *dst = g.dconv + ((size_t)b * Ns + c0) * D;
    const float wp = w2 * invB;
    for (int f = lane; f < nvalid * D; f += 64) {
        const int rr = f / D, cc = f - rr * D;
        float v = 0.0f;
        if (cc < 5) {
            v = mine[rr * 5 + cc];
        } else if ((resp >> rr) & 1ull) {
            const int cw = c0 + rr, cellw = cw / 3, aw = cw - cellw * 3;
            const float x = a.conv[((size_t)b * a.S * a.S + cellw) * a.conv_ld + aw * D + cc];
            const float z = a.label[(size_t)b * a.label_stride + (size_t)cw * D + cc];
            const float rs = a.label[(size_t)b * a.label_stride + (size_t)cw * D + 4];
            v = wp * rs * (1.0f / (1.0f + __expf(-x)) - z);
        }
        dst[f] = v;
    }
}

int launch_yolo_loss_grad(const float *conv, int conv_ld, const float *label, long long label_stride, const float *bboxes,
                          long long bbox_stride, int B, int S, int C, int NB, float stride, float xyscale, const float *anchors,
                          float iou_thr, const float *weights, float *dconv, hipStream_t s) {
    LossGradArgs g;
    LossArgs &a = g.f;
    a.conv = conv; a.pred = nullptr; a.label = label; a.bboxes = bboxes;
    a.partial = nullptr;
    a.B = B; a.S = S; a.C = C; a.NB = NB;
    a.nseg = (3 * S * S + 63) / 64;
    a.conv_ld = conv_ld > 0 ? conv_ld : 3 * (5 + C);
    a.label_stride = label_stride > 0 ? label_stride : (long long)3 * S * S * (5 + C);
    a.bbox_stride = bbox_stride > 0 ? bbox_stride : (long long)4 * NB;
    a.stride = stride; a.xyscale = xyscale;
    for (int i = 0; i < 6; ++i) (&a.anchor[0][0])[i] = anchors[i];
    a.input_size = (float)S * stride;
    a.iou_thr = iou_thr;
    g.dconv = dconv;
    g.w = weights;
    const size_t lds = (size_t)NB * 16 + 16 + 4 * 320 * 4;
    if (lds > 64 * 1024) return fail(YMX_EUNSUPPORTED, "yolo_loss_grad: %d ground-truth slots per image need %zu bytes of LDS", NB, lds);
    hipLaunchKernelGGL(yolo_loss_grad_kernel, dim3((a.nseg + 3) / 4, B), dim3(256), lds, s, g);
    return check_launch("yolo_loss_grad");
}


// ---- label assignment ------------------------------------------------------------------------
// WIDERDataset.preprocess_true_boxes (/root/reference/dataset.py:387-482) for a whole batch, on
// the device, so that the label tensors (3.6 MB per 416x416 image at 80 classes, nearly all
// zeros) are born in HBM instead of crossing PCIe.  Boxes of one image are applied in order
// (a later box overwrites a cell an earlier one claimed), so one wavefront walks an image's
// boxes serially; its lanes split the 5+C values of each label row.  Arithmetic in double,
// as the reference's numpy code; the stored values are rounded to fp32 once.
struct LabelArgs {
    const float *boxes;          // [B][nmax][5]: xmin, ymin, xmax, ymax, class
    const int *counts;           // [B]
    float *label[3];             // [B][S_i][S_i][3][5+C], zeroed by the caller of the kernel
    float *gt[3];                // [B][128][4], zeroed likewise
    int B, nmax, C, S[3];
    double stride[3], anchor[3][3][2];       // anchors in grid units
};

__global__ void __launch_bounds__(64)
assign_labels_kernel(const LabelArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int D = 5 + a.C;
    int n = a.counts ? a.counts[b] : a.nmax;
    n = n > a.nmax ? a.nmax : n;
    int cnt[3] = {0, 0, 0};
    for (int k = 0; k < n; ++k) {
        const float *bx = a.boxes + ((size_t)b * a.nmax + k) * 5;
        const double x0 = bx[0], y0 = bx[1], x1 = bx[2], y1 = bx[3];
        const int cls = (int)bx[4];
        if (cls < 0 || cls >= a.C) continue;
        const double xywh[4] = {(x1 + x0) * 0.5, (y1 + y0) * 0.5, x1 - x0, y1 - y0};
        const double lo = 0.01 * (1.0 / a.C), hi = 1.0 * (1 - 0.01) + lo;
        double iou[3][3];
        int gx[3], gy[3];
        bool any_pos = false;
        // a box whose centre lies outside the image has no responsible cell on any scale (the
        // reference's numpy indexing raises, or wraps a negative index): it is skipped whole --
        // no label row and no slot in the ground-truth ring
        bool inside = true;
        for (int i = 0; i < 3; ++i) {
            gx[i] = (int)floor(xywh[0] / a.stride[i]);
            gy[i] = (int)floor(xywh[1] / a.stride[i]);
            inside = inside && (unsigned)gx[i] < (unsigned)a.S[i] && (unsigned)gy[i] < (unsigned)a.S[i];
        }
        if (!inside) continue;
        auto write = [&](int i, int an) {
            float *row = a.label[i] + ((((size_t)b * a.S[i] + gy[i]) * a.S[i] + gx[i]) * 3 + an) * D;
            for (int c = lane; c < D; c += 64)
                row[c] = (float)(c < 4 ? xywh[c] : (c == 4 ? 1.0 : (c - 5 == cls ? hi : lo)));
        };
        auto push = [&](int i) {
            const int ind = cnt[i] % 128;
            if (lane < 4) a.gt[i][((size_t)b * 128 + ind) * 4 + lane] = (float)xywh[lane];
            ++cnt[i];
        };
        for (int i = 0; i < 3; ++i) {
            const double sx = 1.0 * xywh[0] / a.stride[i], sy = 1.0 * xywh[1] / a.stride[i];
            const double sw = 1.0 * xywh[2] / a.stride[i], sh = 1.0 * xywh[3] / a.stride[i];
            const double cx = gx[i] + 0.5, cy = gy[i] + 0.5;
            bool pos = false;
            for (int an = 0; an < 3; ++an) {
                const double aw = a.anchor[i][an][0], ah = a.anchor[i][an][1];
                const double a1 = sw * sh, a2 = aw * ah;
                const double lx = fmax(sx - sw * 0.5, cx - aw * 0.5), ly = fmax(sy - sh * 0.5, cy - ah * 0.5);
                const double rx = fmin(sx + sw * 0.5, cx + aw * 0.5), ry = fmin(sy + sh * 0.5, cy + ah * 0.5);
                const double inter = fmax(rx - lx, 0.0) * fmax(ry - ly, 0.0);
                const double uni = a1 + a2 - inter;
                iou[i][an] = uni != 0.0 ? inter / uni : 0.0;
                if (iou[i][an] > 0.3) { write(i, an); pos = true; }
            }
            if (pos) { push(i); any_pos = true; }
        }
        if (!any_pos) {
            int best = 0;
            for (int j = 1; j < 9; ++j)
                if (iou[j / 3][j % 3] > iou[best / 3][best % 3]) best = j;
            write(best / 3, best % 3);
            push(best / 3);
        }
    }
}

int launch_assign_labels(const LabelArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(assign_labels_kernel, dim3(a.B), dim3(64), 0, s, a);
    return check_launch("assign_labels");
}

}  // namespace ymx

extern "C" int ymx_assign_labels(const float *boxes, const int *counts, int B, int max_boxes, int input_size, int C,
                                 const float *strides, const float *anchors_grid, float *const *labels, float *const *gt_boxes,
                                 void *stream) {
    using namespace ymx;
    if (!boxes || !strides || !anchors_grid || !labels || !gt_boxes) return fail(YMX_EARG, "assign_labels: null pointer");
    if (B <= 0 || max_boxes <= 0 || input_size <= 0 || C <= 0) return fail(YMX_EARG, "assign_labels: non-positive size");
    LabelArgs a;
    memset(&a, 0, sizeof(a));
    a.boxes = boxes; a.counts = counts; a.B = B; a.nmax = max_boxes; a.C = C;
    for (int i = 0; i < 3; ++i) {
        if (!labels[i] || !gt_boxes[i]) return fail(YMX_EARG, "assign_labels: output %d missing", i);
        if (!(strides[i] >= 1.0f) || strides[i] != (float)(int)strides[i])
            return fail(YMX_EARG, "assign_labels: stride %d must be a whole number >= 1, got %g", i, (double)strides[i]);
        a.label[i] = labels[i]; a.gt[i] = gt_boxes[i];
        a.stride[i] = strides[i];
        a.S[i] = (int)(input_size / (int)strides[i]);
        for (int j = 0; j < 6; ++j) (&a.anchor[i][0][0])[j] = anchors_grid[i * 6 + j];
        const size_t lbytes = (size_t)B * a.S[i] * a.S[i] * 3 * (5 + C) * 4;
        if (hipMemsetAsync(labels[i], 0, lbytes, (hipStream_t)stream) != hipSuccess ||
            hipMemsetAsync(gt_boxes[i], 0, (size_t)B * 128 * 4 * 4, (hipStream_t)stream) != hipSuccess) {
            (void)hipGetLastError();
            return fail(YMX_ELAUNCH, "assign_labels: hipMemsetAsync failed");
        }
    }
    return launch_assign_labels(a, (hipStream_t)stream);
}
