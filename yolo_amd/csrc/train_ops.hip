// Backward pass of the convolution block (/root/reference/layers.py:41-106 differentiated;
// the reference leaves this to TensorFlow's automatic differentiation under model.fit):
//   act_apply        y = act(z) + residual            (training forward keeps z for the derivative)
//   act_backward     dz = dy * act'(z)
//   bias_grad        db[c] = sum over pixels of dz[p][c]
//   conv_wgrad       dW[tap][ci][co] = sum over pixels of x[p + tap][ci] * dz[p][co]   (MFMA)
//   upsample2x_bwd   adjoint of the bilinear x2 resize of layers.py:33-39
// The gradient with respect to the input of a convolution is itself a convolution (flipped,
// transposed kernel) and runs on the forward kernels.
#include <algorithm>

#include "lds_dma.h"

namespace ymx {

// ---- activation: forward with z kept, and derivative ----------------------------------------
__device__ __forceinline__ float mish_grad_f(float z) {
    // d/dz [z * tanh(softplus(z))] with e = exp(z):
    //   tanh(softplus(z)) = n / d,  n = e (e + 2),  d = n + 2
    //   derivative = e * w / d^2,   w = 4 (z + 1) + e (4 z + 6 + e (4 + e))
    // one exponential and one reciprocal (1 ulp each), the rest multiply-adds; -> 1 for large z
    // (taken as 1 beyond 20, where e^4z would leave the fp32 range), -> 0 for very negative z
#pragma clang fp contract(fast)
    const float zc = fminf(z, 20.0f);
    const float e = __builtin_amdgcn_exp2f(zc * 1.44269504088896f);
    const float w = 4.0f * zc + 4.0f + e * (4.0f * zc + 6.0f + e * (4.0f + e));
    const float d = e * (e + 2.0f) + 2.0f;
    const float g = e * w * __builtin_amdgcn_rcpf(d * d);
    return z > 20.0f ? 1.0f : g;
}

template <typename T> __device__ __forceinline__ float ld_f(const T *p, size_t i);
template <> __device__ __forceinline__ float ld_f<float>(const float *p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_f<bf16_t>(const bf16_t *p, size_t i) { return bf16_to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void st_f(T *p, size_t i, float v);
template <> __device__ __forceinline__ void st_f<float>(float *p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_f<bf16_t>(bf16_t *p, size_t i, float v) { p[i] = f32_to_bf16(v); }

// mode 0: y = act(z) + res (res may be null);  mode 1: y = dy(res argument) * act'(z)
template <typename T, int MODE>
__global__ void __launch_bounds__(256)
act_kernel(const T *__restrict__ z, const T *__restrict__ other, T *__restrict__ y, size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = ld_f(z, i);
        float o;
        if (MODE == 0) {
            o = act_f(v, act);
            if (other) o += ld_f(other, i);
        } else {
            const float g = act == 2 ? mish_grad_f(v) : (act == 1 ? (v > 0.0f ? 1.0f : 0.1f) : 1.0f);
            o = ld_f(other, i) * g;
        }
        st_f(y, i, o);
    }
}

// bf16, eight values per thread (16-byte loads and stores); n is a multiple of 8 here
template <int MODE>
__global__ void __launch_bounds__(256)
act_kernel_bf16x8(const uint4 *__restrict__ z, const uint4 *__restrict__ other, uint4 *__restrict__ y, size_t n8, int act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 zv = z[i];
        uint4 ov = make_uint4(0u, 0u, 0u, 0u);
        if (MODE == 1 || other) ov = other[i];
        const uint32_t zw[4] = {zv.x, zv.y, zv.z, zv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float z0 = __uint_as_float(zw[k] << 16), z1 = __uint_as_float(zw[k] & 0xffff0000u);
            const float o0 = __uint_as_float(ow[k] << 16), o1 = __uint_as_float(ow[k] & 0xffff0000u);
            float a0, a1;
            if (MODE == 0) {
                a0 = act_f(z0, act) + o0;
                a1 = act_f(z1, act) + o1;
            } else {
                a0 = o0 * (act == 2 ? mish_grad_f(z0) : (act == 1 ? (z0 > 0.0f ? 1.0f : 0.1f) : 1.0f));
                a1 = o1 * (act == 2 ? mish_grad_f(z1) : (act == 1 ? (z1 > 0.0f ? 1.0f : 0.1f) : 1.0f));
            }
            r[k] = pack2_bf16(a0, a1);
        }
        y[i] = make_uint4(r[0], r[1], r[2], r[3]);
    }
}

int launch_act(const void *z, const void *other, void *y, size_t n, int act, int dtype, int mode, hipStream_t s) {
    if (dtype == YMX_BF16 && n % 8 == 0 && aligned16(z) && aligned16(y) && aligned16(other)) {
        const size_t n8 = n / 8;
        const int grid8 = (int)std::min<size_t>((n8 + 255) / 256, 1 << 16);
        if (mode == 0) hipLaunchKernelGGL(act_kernel_bf16x8<0>, dim3(grid8), dim3(256), 0, s, (const uint4 *)z, (const uint4 *)other, (uint4 *)y, n8, act);
        else hipLaunchKernelGGL(act_kernel_bf16x8<1>, dim3(grid8), dim3(256), 0, s, (const uint4 *)z, (const uint4 *)other, (uint4 *)y, n8, act);
        return check_launch("act");
    }
    const int grid = (int)std::min<size_t>((n + 255) / 256, 1 << 16);
    if (dtype == YMX_BF16) {
        if (mode == 0) hipLaunchKernelGGL((act_kernel<bf16_t, 0>), dim3(grid), dim3(256), 0, s, (const bf16_t *)z, (const bf16_t *)other, (bf16_t *)y, n, act);
        else hipLaunchKernelGGL((act_kernel<bf16_t, 1>), dim3(grid), dim3(256), 0, s, (const bf16_t *)z, (const bf16_t *)other, (bf16_t *)y, n, act);
    } else {
        if (mode == 0) hipLaunchKernelGGL((act_kernel<float, 0>), dim3(grid), dim3(256), 0, s, (const float *)z, (const float *)other, (float *)y, n, act);
        else hipLaunchKernelGGL((act_kernel<float, 1>), dim3(grid), dim3(256), 0, s, (const float *)z, (const float *)other, (float *)y, n, act);
    }
    return check_launch("act");
}

// ---- bias gradient: column sums, two deterministic stages ---------------------------------------
// stage 1: workgroup s sums rows [s * rows_per, ...) for every channel into partial[s][C];
// stage 2: one thread per channel adds the partials in order.
template <typename T>
__global__ void __launch_bounds__(256)
colsum_kernel(const T *__restrict__ x, float *__restrict__ partial, long long P, int C, int ld, long long rows_per) {
    const long long r0 = (long long)blockIdx.x * rows_per;
    const long long r1 = r0 + rows_per < P ? r0 + rows_per : P;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.0f;
        for (long long r = r0; r < r1; ++r) acc += ld_f(x, (size_t)r * ld + c);
        partial[(size_t)blockIdx.x * C + c] = acc;
    }
}

__global__ void __launch_bounds__(256)
sum_partials_kernel(const float *__restrict__ partial, float *__restrict__ out, size_t n, int parts) {
    // four running sums over parts s = 0, 1, 2, 3 (mod 4), added at the end: a fixed order, and
    // the loads of a thread do not wait for one another
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int s = 0;
        for (; s + 4 <= parts; s += 4) {
            const float v0 = partial[(size_t)s * n + i], v1 = partial[(size_t)(s + 1) * n + i];
            const float v2 = partial[(size_t)(s + 2) * n + i], v3 = partial[(size_t)(s + 3) * n + i];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; s < parts; ++s) a0 += partial[(size_t)s * n + i];
        out[i] = (a0 + a1) + (a2 + a3);
    }
}

// The same for short vectors (bias gradients: n = channels) where a thread per element would
// leave the GPU to a handful of serial loops: 32 elements x 32 part lanes per workgroup, lane l
// adds parts l, l + 32, ... and the 32 lane sums are added in order.
__global__ void __launch_bounds__(1024)
sum_partials_small_kernel(const float *__restrict__ partial, float *__restrict__ out, int n, int parts) {
    __shared__ float red[32][32];
    const int t = threadIdx.x, e = t & 31, pl = t >> 5;
    const int i = blockIdx.x * 32 + e;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (i < n) {
        int s = pl;
        for (; s + 96 < parts; s += 128) {
            const float v0 = partial[(size_t)s * n + i], v1 = partial[(size_t)(s + 32) * n + i];
            const float v2 = partial[(size_t)(s + 64) * n + i], v3 = partial[(size_t)(s + 96) * n + i];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; s < parts; s += 32) a0 += partial[(size_t)s * n + i];
    }
    red[pl][e] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (t < 32 && i < n) {
        float sum = 0.0f;
#pragma unroll
        for (int l = 0; l < 32; ++l) sum += red[l][t];
        out[i] = sum;
    }
}

// Partial weight gradients: short tensors with many slices (1x1 layers: 8 K - 256 K elements,
// up to 256 slices) go through the part-lane form too, where a thread per element would walk
// the slices one load round trip after the other.
static void launch_sum_partials(const float *partial, float *out, size_t n, int parts, hipStream_t s) {
    if (n <= (1u << 18) && parts >= 8)
        hipLaunchKernelGGL(sum_partials_small_kernel, dim3((unsigned)((n + 31) / 32)), dim3(1024), 0, s, partial, out, (int)n, parts);
    else
        hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65536)), dim3(256), 0, s, partial, out, n, parts);
}

// bf16 rows whose 16-byte chunks divide the workgroup (C / 8 a power of two up to 256, dense
// or padded rows): thread = (row lane, chunk); every thread sums its chunk's eight channels
// over the slab's rows in steps of 256 / chunks rows, the row lanes are then added in LDS in
// a fixed order.
__global__ void __launch_bounds__(256)
colsum_bf16_kernel(const bf16_t *__restrict__ x, float *__restrict__ partial, long long P, int C, int ld, long long rows_per) {
    __shared__ float red[256 * 8];
    const int nch = C >> 3, t = threadIdx.x;
    const int ch = t % nch, rl = t / nch, nrl = 256 / nch;
    const long long r0 = (long long)blockIdx.x * rows_per;
    const long long r1 = r0 + rows_per < P ? r0 + rows_per : P;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long r = r0 + rl; r < r1; r += nrl) {
        const uint4 v = *(const uint4 *)(x + (size_t)r * ld + ch * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[2 * k] += __uint_as_float(w[k] << 16);
            acc[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[t * 8 + k] = acc[k];
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        const int cch = c >> 3, k = c & 7;
        float sum = 0.0f;
        for (int l = 0; l < nrl; ++l) sum += red[(l * nch + cch) * 8 + k];
        partial[(size_t)blockIdx.x * C + c] = sum;
    }
}

static int colsum_parts(long long P) {
    long long parts = (P + 511) / 512;
    return (int)(parts > 512 ? 512 : (parts < 1 ? 1 : parts));
}
size_t bias_grad_workspace_bytes(long long P, int C) { return (size_t)colsum_parts(P) * C * 4; }

int launch_bias_grad(const void *dz, long long P, int C, int ld, int dtype, float *db, void *ws, hipStream_t s) {
    const int parts = colsum_parts(P);
    const long long rows_per = (P + parts - 1) / parts;
    float *partial = (float *)ws;
    const int nch = C / 8;
    const bool fast = dtype == YMX_BF16 && C % 8 == 0 && nch <= 256 && (nch & (nch - 1)) == 0 && ld % 8 == 0 && aligned16(dz);
    if (fast) hipLaunchKernelGGL(colsum_bf16_kernel, dim3(parts), dim3(256), 0, s, (const bf16_t *)dz, partial, P, C, ld, rows_per);
    else if (dtype == YMX_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(parts), dim3(256), 0, s, (const bf16_t *)dz, partial, P, C, ld, rows_per);
    else hipLaunchKernelGGL(colsum_kernel<float>, dim3(parts), dim3(256), 0, s, (const float *)dz, partial, P, C, ld, rows_per);
    int rc = check_launch("bias_grad");
    if (rc) return rc;
    hipLaunchKernelGGL(sum_partials_small_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, partial, db, C, parts);
    return check_launch("bias_grad_reduce");
}

// dz = dy * act'(z) and the column sums of dz in one pass (bf16, C / 8 a power of two up to
// 256, dense rows): a thread keeps the same 16-byte channel chunk for its whole strided walk
// (256 is a multiple of the chunks per row), so it carries eight running sums; the workgroup
// then adds its row lanes in LDS in a fixed order and writes one partial row, and
// sum_partials_small_kernel adds the workgroups' rows in order.  The sums are taken over the
// bf16-rounded dz, the values the weight gradient reads.
__global__ void __launch_bounds__(256)
act_bwd_bias_kernel(const uint4 *__restrict__ z, const uint4 *__restrict__ dy, uint4 *__restrict__ dz, float *__restrict__ partial,
                    size_t n8, int C, int act) {
    __shared__ float red[256 * 8];
    const int nch = C >> 3, t = threadIdx.x;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto one = [&](const uint4 zv, const uint4 ov) {
        const uint32_t zw[4] = {zv.x, zv.y, zv.z, zv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float z0 = __uint_as_float(zw[k] << 16), z1 = __uint_as_float(zw[k] & 0xffff0000u);
            const float o0 = __uint_as_float(ow[k] << 16), o1 = __uint_as_float(ow[k] & 0xffff0000u);
            const float a0 = o0 * (act == 2 ? mish_grad_f(z0) : (z0 > 0.0f ? 1.0f : 0.1f));
            const float a1 = o1 * (act == 2 ? mish_grad_f(z1) : (z1 > 0.0f ? 1.0f : 0.1f));
            r[k] = pack2_bf16(a0, a1);
            acc[2 * k] += __uint_as_float(r[k] << 16);
            acc[2 * k + 1] += __uint_as_float(r[k] & 0xffff0000u);
        }
        return make_uint4(r[0], r[1], r[2], r[3]);
    };
    // two chunks of the strided walk per iteration: four 16-byte loads in flight per thread
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + t;
    for (; i + step < n8; i += 2 * step) {
        const uint4 za = z[i], oa = dy[i], zb = z[i + step], ob = dy[i + step];
        dz[i] = one(za, oa);
        dz[i + step] = one(zb, ob);
    }
    if (i < n8) dz[i] = one(z[i], dy[i]);
#pragma unroll
    for (int k = 0; k < 8; ++k) red[t * 8 + k] = acc[k];
    __syncthreads();
    const int nrl = 256 / nch;
    for (int c = t; c < C; c += 256) {
        const int cch = c >> 3, k = c & 7;
        float sum = 0.0f;
        for (int l = 0; l < nrl; ++l) sum += red[(l * nch + cch) * 8 + k];
        partial[(size_t)blockIdx.x * C + c] = sum;
    }
}

static int act_bwd_bias_parts(size_t n8) {
    // two workgroups per CU: measured over the detector's activation shapes at batch 64 (tools/actbwd_time.py),
    // 256 / 384 / 512 / 768 / 1024 / 2048 workgroups take 4.51 / 4.01 / 3.56 / 3.59 / 3.81 / 4.06 ms per step
    const size_t g = (n8 + 255) / 256;
    return (int)(g > 512 ? 512 : (g < 1 ? 1 : g));
}
bool act_bwd_bias_supported(long long P, int C, int act, int dtype) {
    const int nch = C / 8;
    return dtype == YMX_BF16 && (act == 1 || act == 2) && C % 8 == 0 && nch >= 1 && nch <= 256 && (nch & (nch - 1)) == 0 && P > 0;
}
size_t act_bwd_bias_workspace_bytes(long long P, int C) { return (size_t)act_bwd_bias_parts((size_t)P * C / 8) * C * 4; }

int launch_act_bwd_bias(const void *z, const void *dy, void *dz, long long P, int C, int act, float *db, void *ws, hipStream_t s) {
    const size_t n8 = (size_t)P * C / 8;
    const int parts = act_bwd_bias_parts(n8);
    hipLaunchKernelGGL(act_bwd_bias_kernel, dim3(parts), dim3(256), 0, s, (const uint4 *)z, (const uint4 *)dy, (uint4 *)dz, (float *)ws, n8, C, act);
    int rc = check_launch("act_backward_bias");
    if (rc) return rc;
    hipLaunchKernelGGL(sum_partials_small_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, (const float *)ws, db, C, parts);
    return check_launch("act_backward_bias_reduce");
}


// ---- batch normalisation in training mode -------------------------------------------------------
// /root/reference/layers.py:4-15 under model.fit: Keras normalises with the statistics of the batch,
//   y = gamma * (u - mean_B) / sqrt(var_B + eps) + beta   (var_B biased, over all pixels of the batch),
// and keeps moving averages for inference.  The pieces here (dense [n_pix][C] rows, bf16 or fp32):
//   bn_stats       sums[0][c] = sum_p u,  sums[1][c] = sum_p u^2            (forward)
//   bn_act_bwd     dz = dy * act'(u * scale[c] + shift[c]) stored, and
//                  sums[0][c] = sum_p dz,  sums[1][c] = sum_p dz * u        (backward, pass 1)
//   bn_bwd_apply   du = a[c] * dz + b[c] * u + c[c]                         (backward, pass 2)
// from which the host forms mean / variance and, with s = 1 / sqrt(var + eps), N pixels,
//   dbeta = sum dz,  dgamma = s * (sum dz u - mean * sum dz),
//   du = gamma s (dz - dbeta / N - xhat dgamma / N),  xhat = (u - mean) s.
// Sums are taken in two ordered stages (slabs of rows per workgroup, then the slabs in order), so
// results are reproducible.  One thread walks one channel of its slab; consecutive threads read
// consecutive channels.
template <typename T, int MODE>          // MODE 0: (u, u^2);  1: (dz, dz * u) with dz computed and stored
__global__ void __launch_bounds__(256)
bn_reduce_kernel(const T *__restrict__ u, const T *__restrict__ dy, T *__restrict__ dz, const float *__restrict__ scale,
                 const float *__restrict__ shift, float *__restrict__ partial, long long P, int C, long long rows_per, int act) {
    const long long r0 = (long long)blockIdx.x * rows_per;
    const long long r1 = r0 + rows_per < P ? r0 + rows_per : P;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s0 = 0.0f, s1 = 0.0f;
        const float sc = MODE == 1 ? scale[c] : 1.0f, sh = MODE == 1 ? shift[c] : 0.0f;
        for (long long r = r0; r < r1; ++r) {
            const size_t i = (size_t)r * C + c;
            const float v = ld_f(u, i);
            if (MODE == 0) {
                s0 += v;
                s1 += v * v;
            } else {
                const float z = v * sc + sh;
                const float g = act == 2 ? mish_grad_f(z) : (act == 1 ? (z > 0.0f ? 1.0f : 0.1f) : 1.0f);
                st_f(dz, i, ld_f(dy, i) * g);
                const float d = ld_f(dz, i);                         // the rounded value, what pass 2 reads
                s0 += d;
                s1 += d * v;
            }
        }
        partial[((size_t)blockIdx.x * 2) * C + c] = s0;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = s1;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const T *__restrict__ dz, const T *__restrict__ u, const float *__restrict__ a, const float *__restrict__ b,
                    const float *__restrict__ c0, T *__restrict__ du, size_t n, int C) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        st_f(du, i, a[c] * ld_f(dz, i) + b[c] * ld_f(u, i) + c0[c]);
    }
}

size_t bn_workspace_bytes(long long P, int C) { return (size_t)colsum_parts(P) * 2 * C * 4; }

int launch_bn_reduce(const void *u, const void *dy, void *dz, const float *scale, const float *shift, long long P, int C, int act,
                     int dtype, float *sums, void *ws, hipStream_t s) {
    const int parts = colsum_parts(P);
    const long long rows_per = (P + parts - 1) / parts;
    float *partial = (float *)ws;
    if (dy == nullptr) {
        if (dtype == YMX_BF16) hipLaunchKernelGGL((bn_reduce_kernel<bf16_t, 0>), dim3(parts), dim3(256), 0, s, (const bf16_t *)u, (const bf16_t *)nullptr, (bf16_t *)nullptr, scale, shift, partial, P, C, rows_per, act);
        else hipLaunchKernelGGL((bn_reduce_kernel<float, 0>), dim3(parts), dim3(256), 0, s, (const float *)u, (const float *)nullptr, (float *)nullptr, scale, shift, partial, P, C, rows_per, act);
    } else {
        if (dtype == YMX_BF16) hipLaunchKernelGGL((bn_reduce_kernel<bf16_t, 1>), dim3(parts), dim3(256), 0, s, (const bf16_t *)u, (const bf16_t *)dy, (bf16_t *)dz, scale, shift, partial, P, C, rows_per, act);
        else hipLaunchKernelGGL((bn_reduce_kernel<float, 1>), dim3(parts), dim3(256), 0, s, (const float *)u, (const float *)dy, (float *)dz, scale, shift, partial, P, C, rows_per, act);
    }
    int rc = check_launch("bn_reduce");
    if (rc) return rc;
    // partial is [parts][2 C]: the two sums of a slab lie side by side, so one ordered pass adds both
    hipLaunchKernelGGL(sum_partials_small_kernel, dim3((2 * C + 31) / 32), dim3(1024), 0, s, partial, sums, 2 * C, parts);
    return check_launch("bn_reduce_sum");
}

int launch_bn_bwd_apply(const void *dz, const void *u, const float *a, const float *b, const float *c, void *du, long long P, int C,
                        int dtype, hipStream_t s) {
    const size_t n = (size_t)P * C;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 1 << 16);
    if (dtype == YMX_BF16) hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)dz, (const bf16_t *)u, a, b, c, (bf16_t *)du, n, C);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)dz, (const float *)u, a, b, c, (float *)du, n, C);
    return check_launch("bn_backward_apply");
}

// ---- weight gradient --------------------------------------------------------------------------
// GEMM  dW[ci][co] (per tap) = sum_p X[p + tap][ci] * dZ[p][co]: both operands have the
// reduction index (pixels) as their SLOW axis in memory, the opposite of what an MFMA lane
// wants (8 consecutive reduction elements).  Tiles are therefore staged in LDS as they lie in
// memory, [pixel][channel] rows, and read with the transposing LDS read of gfx950
// (ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns arrives column-major),
// two reads per operand fragment.
// Workgroup = 4 waves = tile of TCI (64 or 128) input channels x 128 output channels of one
// tap and one slice of the pixels; the waves split it 1 x 4 (64 x 32 each) or 2 x 2 (64 x 64
// each).  Per step 64 pixels: the next step's rows are loaded into registers while the MFMAs
// of the current one run.  Slices (grid.z) write separate partial results that a second
// kernel adds in order, so the result is reproducible.
struct WgradArgs {
    const bf16_t *x, *dz;
    float *out;                 // [splits][k*k][Cin][Cout]
    int B, H, W, Cin, OH, OW, Cout, k, stride, pad;
    int x_ld, dz_ld;
    int P;                      // B * OH * OW
    int chunks_per_split;       // 64-pixel chunks per slice
    int n_ci_tiles;
    unsigned ow_m, ow_s, ohw_m, ohw_s;     // fastdiv constants of OW and OH * OW
};

constexpr int WG_PB = 128 * 2 + 16;      // LDS row pitch of the dZ tile: 128 channels + 16 bytes

// Eight consecutive reduction rows of one column per lane for N fragments 32 bytes (16
// columns) apart: two transposing reads each (rows 0-3 and 4-7 of the lane group's block,
// 4 * PITCH bytes apart), one wait for all of them.
template <int PITCH>
__device__ __forceinline__ void lds_read_tr8x4(const char *p, uint4 (&f)[4]) {
    uint2 r[8];
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
                 "ds_read_b64_tr_b16 %1, %8 offset:%9\n\t"
                 "ds_read_b64_tr_b16 %2, %8 offset:32\n\t"
                 "ds_read_b64_tr_b16 %3, %8 offset:%10\n\t"
                 "ds_read_b64_tr_b16 %4, %8 offset:64\n\t"
                 "ds_read_b64_tr_b16 %5, %8 offset:%11\n\t"
                 "ds_read_b64_tr_b16 %6, %8 offset:96\n\t"
                 "ds_read_b64_tr_b16 %7, %8 offset:%12\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(addr), "n"(4 * PITCH), "n"(4 * PITCH + 32), "n"(4 * PITCH + 64), "n"(4 * PITCH + 96)
                 : "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = make_uint4(r[2 * i].x, r[2 * i].y, r[2 * i + 1].x, r[2 * i + 1].y);
}
template <int PITCH>
__device__ __forceinline__ void lds_read_tr8x2(const char *p, uint4 (&f)[2]) {
    uint2 r[4];
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                 "ds_read_b64_tr_b16 %1, %4 offset:%5\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:32\n\t"
                 "ds_read_b64_tr_b16 %3, %4 offset:%6\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                 : "v"(addr), "n"(4 * PITCH), "n"(4 * PITCH + 32)
                 : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) f[i] = make_uint4(r[2 * i].x, r[2 * i].y, r[2 * i + 1].x, r[2 * i + 1].y);
}

template <int TCI>
__global__ void __launch_bounds__(256)
conv_wgrad_kernel(const WgradArgs a) {
    constexpr int PA = TCI * 2 + 16;                 // LDS row pitch of the X tile
    constexpr int WCI = TCI / 64;                    // waves along the input channels
    constexpr int NJ = 2 * WCI;                      // 16-column blocks of output channels per wave
    constexpr int NA = TCI / 8 * 64 / 256;           // 16-byte items of the X tile per thread (2 or 4)
    __shared__ __attribute__((aligned(16))) char sA[64 * PA];
    __shared__ __attribute__((aligned(16))) char sB[64 * WG_PB];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wci = wave % WCI, wco = wave / WCI;
    const int tile_ci = blockIdx.x % a.n_ci_tiles, tile_co = blockIdx.x / a.n_ci_tiles;
    const int tap = blockIdx.y, tr = tap / a.k, tc = tap - tr * a.k;
    const int ci0 = tile_ci * TCI, co0 = tile_co * 128;
    const int c_begin = blockIdx.z * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    const int nchunks = (a.P + 63) / 64;
    if (c_end > nchunks) c_end = nchunks;

    f32x4_t acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    uint4 ra[NA], rb[4];
    auto load_chunk = [&](int chunk) {
        const int p0 = chunk * 64;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int e = t + u * 256, row = e / (TCI / 8), it = e % (TCI / 8);
            const int p = p0 + row;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int ci = ci0 + it * 8;
            if (p < a.P && ci < a.Cin) {
                const int b = fastdiv(p, a.ohw_m, a.ohw_s), rem = p - b * (a.OH * a.OW);
                const int oh = fastdiv(rem, a.ow_m, a.ow_s), ow = rem - oh * a.OW;
                const int ih = oh * a.stride + tr - a.pad, iw = ow * a.stride + tc - a.pad;
                if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                    v = *(const uint4 *)(a.x + (((size_t)b * a.H + ih) * a.W + iw) * a.x_ld + ci);
            }
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + u * 256, row = e >> 4, it = e & 15;
            const int p = p0 + row;
            const int co = co0 + it * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (p < a.P && co < a.Cout) v = *(const uint4 *)(a.dz + (size_t)p * a.dz_ld + co);
            rb[u] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int e = t + u * 256, row = e / (TCI / 8), it = e % (TCI / 8);
            *(uint4 *)(sA + row * PA + it * 16) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + u * 256, row = e >> 4, it = e & 15;
            *(uint4 *)(sB + row * WG_PB + it * 16) = rb[u];
        }
    };

    // transposed-read geometry: 16-lane group g covers reduction rows 8 g .. 8 g + 7 of a
    // 32-pixel step; lane 4 q + p of the group addresses row q, columns 4 p .. 4 p + 3
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int a_off = (8 * g + q4) * PA + wci * 128 + p4 * 8;
    const int b_off = (8 * g + q4) * WG_PB + wco * (NJ * 32) + p4 * 8;

    if (c_begin < c_end) load_chunk(c_begin);
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        __syncthreads();                       // the previous step's reads are done
        store_chunk();
        __syncthreads();
        if (chunk + 1 < c_end) load_chunk(chunk + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {       // two 32-pixel reduction steps per chunk
            uint4 fa[4], fb[NJ];
            lds_read_tr8x4<PA>(sA + ks * 32 * PA + a_off, fa);
            if constexpr (NJ == 4) {
                lds_read_tr8x4<WG_PB>(sB + ks * 32 * WG_PB + b_off, fb);
            } else {
                lds_read_tr8x2<WG_PB>(sB + ks * 32 * WG_PB + b_off, fb);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[i]),
                                                                         __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
        }
    }
    // D[row = input channel][col = output channel]: lane holds rows 4 (lane >> 4) + r of column lane & 15
    float *out = a.out + ((size_t)blockIdx.z * a.k * a.k + tap) * a.Cin * a.Cout;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wci * 64 + i * 16 + 4 * (lane >> 4) + r;
                const int co = co0 + wco * (NJ * 16) + j * 16 + (lane & 15);
                if (ci < a.Cin && co < a.Cout) out[(size_t)ci * a.Cout + co] = acc[i][j][r];
            }
}

// ---- weight gradient, LDS-DMA form ----------------------------------------------------------
// The same GEMM with the tiles brought into LDS by LDS-DMA (no staging registers, no ds_write)
// and a larger tile for wide layers: TCI = 256 input channels x 128 output channels, waves 2 x 2
// of 128 x 64 (three LDS bytes read per four MFMA results instead of four).  A step is 32
// pixels; a ring of three or four buffers, the transfers of the next two or three steps are in
// flight under the current step's MFMAs (counted vmcnt: they complete in order), one barrier per
// step.  LDS rows are unpadded ([pixel][channel], as the DMA writes them: lane l
// of an instruction fills bytes 16 l .. 16 l + 15); bank conflicts of the transposing reads are
// avoided by permuting the 32-byte column groups of a row with an XOR of row bits, applied on
// the SOURCE side of the DMA and undone in the read addresses.  Rows outside the image (a
// shifted tap at the border, the tail of the last step, channels beyond Cin / Cout) are
// out-of-range buffer offsets, which the hardware fills with zeros.
template <int TCI> struct WgradGeom {
    static constexpr int WCI = TCI >= 128 ? 2 : 1;     // waves along the input channels
    static constexpr int WCO = 4 / WCI;
    static constexpr int NI = TCI / (16 * WCI);        // 16-row blocks of input channels per wave
    static constexpr int NJ = 128 / (16 * WCO);        // 16-column blocks of output channels per wave
    static constexpr int RA = TCI * 2, RB = 256;       // LDS row bytes
    static constexpr int A_BYTES = 32 * RA, B_BYTES = 32 * RB;
    static constexpr int A_INSTR = A_BYTES / 1024, B_INSTR = B_BYTES / 1024;
    static constexpr int BUF = A_BYTES + B_BYTES;
    // ring stages: what one CU must keep in flight to cover the latency of its loads is about
    // 100 KB, so the ring is as deep as LDS allows at two (TCI = 256) or more workgroups per CU
    static constexpr int NS = TCI == 256 ? 3 : 4;
    static constexpr int PER_WAVE = (A_INSTR + B_INSTR) / 4;   // DMA instructions per wave and step
    static constexpr int LDS_BYTES = NS * BUF;
};

__device__ __forceinline__ int wg_rowmix(int r) { return (r & 3) | ((r >> 1) & 4); }

// two transposing reads per address (rows r and r + 4 of the lane group's block), four addresses
template <int ROWB>
__device__ __forceinline__ void lds_read_tr8_addr4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, uint4 *f) {
    uint2 r[8];
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
                 "ds_read_b64_tr_b16 %1, %8 offset:%12\n\t"
                 "ds_read_b64_tr_b16 %2, %9\n\t"
                 "ds_read_b64_tr_b16 %3, %9 offset:%12\n\t"
                 "ds_read_b64_tr_b16 %4, %10\n\t"
                 "ds_read_b64_tr_b16 %5, %10 offset:%12\n\t"
                 "ds_read_b64_tr_b16 %6, %11\n\t"
                 "ds_read_b64_tr_b16 %7, %11 offset:%12\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(4 * ROWB)
                 : "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = make_uint4(r[2 * i].x, r[2 * i].y, r[2 * i + 1].x, r[2 * i + 1].y);
}
template <int ROWB>
__device__ __forceinline__ void lds_read_tr8_addr2(unsigned a0, unsigned a1, uint4 *f) {
    uint2 r[4];
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                 "ds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
                 "ds_read_b64_tr_b16 %2, %5\n\t"
                 "ds_read_b64_tr_b16 %3, %5 offset:%6\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                 : "v"(a0), "v"(a1), "n"(4 * ROWB)
                 : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) f[i] = make_uint4(r[2 * i].x, r[2 * i].y, r[2 * i + 1].x, r[2 * i + 1].y);
}

template <int TCI>
__global__ void __launch_bounds__(256)
conv_wgrad_dma_kernel(const WgradArgs a, const unsigned x_bytes, const unsigned dz_bytes) {
    typedef WgradGeom<TCI> G;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wci = wave % G::WCI, wco = wave / G::WCI;
    const int tile_ci = blockIdx.x % a.n_ci_tiles, tile_co = blockIdx.x / a.n_ci_tiles;
    const int tap = blockIdx.y, tr = tap / a.k, tc = tap - tr * a.k;
    const int ci0 = tile_ci * TCI, co0 = tile_co * 128;
    const int c_begin = blockIdx.z * a.chunks_per_split;        // steps of 32 pixels
    int c_end = c_begin + a.chunks_per_split;
    const int nsteps = (a.P + 31) / 32;
    if (c_end > nsteps) c_end = nsteps;
    const i32x4_t xrsrc = make_rsrc(a.x, x_bytes);
    const i32x4_t zrsrc = make_rsrc(a.dz, dz_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

    // DMA geometry.  X tile: instruction j covers rows [j * RPA, (j + 1) * RPA) of the step,
    // lane l fills slot l % SA of row l / SA with the data of the un-permuted slot.
    constexpr int SA = G::RA / 16, RPA = 1024 / G::RA;           // 16-byte slots per row, rows per instruction
    constexpr int AJ = G::A_INSTR / 4 > 0 ? G::A_INSTR / 4 : 1;  // X instructions per wave (TCI = 64: waves 0-3 take one each)
    const int a_rl = lane / SA, a_sl = lane % SA;
    constexpr int BJ = G::B_INSTR / 4;                           // dZ instructions per wave (2)
    const int b_rl = lane >> 4, b_sl = lane & 15;

    auto issue = [&](int step, int buf) {
        const int p0 = step * 32;
        const unsigned base = lds0 + (unsigned)(buf * G::BUF);
#pragma unroll
        for (int jj = 0; jj < AJ; ++jj) {
            const int j = wave + 4 * jj;
            if (G::A_INSTR < 4 && j >= G::A_INSTR) break;
            const int row = j * RPA + a_rl;
            const int p = p0 + row;
            const int pair = (a_sl >> 1) ^ (wg_rowmix(row) & (SA / 2 - 1));
            const int ci = ci0 + (pair * 2 + (a_sl & 1)) * 8;
            unsigned off = OOB;
            if (p < a.P && ci < a.Cin && step < c_end) {
                const int b = fastdiv(p, a.ohw_m, a.ohw_s), rem = p - b * (a.OH * a.OW);
                const int oh = fastdiv(rem, a.ow_m, a.ow_s), ow = rem - oh * a.OW;
                const int ih = oh * a.stride + tr - a.pad, iw = ow * a.stride + tc - a.pad;
                if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                    off = (unsigned)((((b * a.H + ih) * a.W + iw) * a.x_ld + ci) * 2);
            }
            lds_dma16(xrsrc, base + (unsigned)(j * 1024), off);
        }
#pragma unroll
        for (int jj = 0; jj < BJ; ++jj) {
            const int j = wave + 4 * jj;
            const int row = j * 4 + b_rl;
            const int p = p0 + row;
            const int pair = (b_sl >> 1) ^ wg_rowmix(row);
            const int co = co0 + (pair * 2 + (b_sl & 1)) * 8;
            unsigned off = OOB;
            if (p < a.P && co < a.Cout && step < c_end) off = (unsigned)((p * a.dz_ld + co) * 2);
            lds_dma16(zrsrc, base + (unsigned)(G::A_BYTES + j * 1024), off);
        }
    };

    f32x4_t acc[G::NI][G::NJ];
#pragma unroll
    for (int i = 0; i < G::NI; ++i)
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // transposed-read addresses: 16-lane group g covers rows 8 g .. 8 g + 7 of the step; lane
    // 4 q + p of the group addresses row 8 g + q (and + 4), bytes 8 p .. 8 p + 7 of a block's 32
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int r0 = 8 * g + q4;
    const int mix = wg_rowmix(r0);                               // the same for row r0 + 4
    unsigned aoff[G::NI], boff[G::NJ];
#pragma unroll
    for (int i = 0; i < G::NI; ++i)
        aoff[i] = (unsigned)(r0 * G::RA + (((wci * G::NI + i) ^ (mix & (SA / 2 - 1))) * 32) + p4 * 8);
#pragma unroll
    for (int j = 0; j < G::NJ; ++j)
        boff[j] = (unsigned)(G::A_BYTES + r0 * G::RB + (((wco * G::NJ + j) ^ mix) * 32) + p4 * 8);

    // prologue: NS - 1 steps in flight (past the end of the slice the transfers move zeros, which
    // keeps every wave's instruction count per step the same: the counted waits rely on it)
#pragma unroll
    for (int i = 0; i < G::NS - 1; ++i) issue(c_begin + i, i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G::NS - 2) * G::PER_WAVE) : "memory");
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (int step = c_begin; step < c_end; ++step) {
        {
            int nb = buf + G::NS - 1;
            nb = nb >= G::NS ? nb - G::NS : nb;
            issue(step + G::NS - 1, nb);                         // into the buffer read in the previous iteration
        }
        const unsigned base = lds0 + (unsigned)(buf * G::BUF);
        uint4 fa[G::NI], fb[G::NJ];
#pragma unroll
        for (int i = 0; i < G::NI; i += 4)
            lds_read_tr8_addr4<G::RA>(base + aoff[i], base + aoff[i + 1], base + aoff[i + 2], base + aoff[i + 3], fa + i);
        if constexpr (G::NJ == 4) lds_read_tr8_addr4<G::RB>(base + boff[0], base + boff[1], base + boff[2], base + boff[3], fb);
        else lds_read_tr8_addr2<G::RB>(base + boff[0], base + boff[1], fb);
#pragma unroll
        for (int i = 0; i < G::NI; ++i)
#pragma unroll
            for (int j = 0; j < G::NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[i]),
                                                                     __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G::NS - 2) * G::PER_WAVE) : "memory");   // this wave's part of the next step has landed
        __builtin_amdgcn_s_barrier();                            // everyone's has, and everyone is done reading this buffer
        asm volatile("" ::: "memory");
        buf = buf + 1 == G::NS ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float *out = a.out + ((size_t)blockIdx.z * a.k * a.k + tap) * a.Cin * a.Cout;
#pragma unroll
    for (int i = 0; i < G::NI; ++i)
#pragma unroll
        for (int j = 0; j < G::NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wci * (G::NI * 16) + i * 16 + 4 * (lane >> 4) + r;
                const int co = co0 + wco * (G::NJ * 16) + j * 16 + (lane & 15);
                if (ci < a.Cin && co < a.Cout) out[(size_t)ci * a.Cout + co] = acc[i][j][r];
            }
}

thread_local int g_wgrad_variant = 0;        // 0: automatic; 1: register-staged kernel; 2: per-tap LDS-DMA kernel (tuning option "wgrad_variant")

// ---- weight gradient of 3x3 stride-1 layers: all nine taps from one resident slab ------------
// The kernels above give every tap its own workgroups, so a layer's dz tiles and (shifted) x
// tiles cross the L2 nine times: 85 FLOP per byte, and the step time is what L2 delivers.
// For a 3x3 stride-1 convolution output pixel p and tap (r, s) meet input pixel
// p + (r - 1) W + (s - 1) of the flattened [B*H*W] pixel axis, so one contiguous run of x rows
// serves all nine taps of a run of output pixels (the idea of conv_igemm_halo.hip).  Here a
// workgroup owns 64 input x 64 output channels and ALL taps (wave = 32 x 32 channels x 9
// taps: 144 accumulator registers); per 32-pixel step it brings in 32 new x rows and 32 dz
// rows (8 KB for 2.4 MFLOP: 295 FLOP per byte) by LDS-DMA:
//   x rows   a ring of RX rows (a power of two, >= the 2 W + 34 rows a step touches plus the
//            steps in flight), row of input pixel i at slot i mod RX, fetched in aligned groups
//            of eight rows (exactly four groups per step); pixels before the tensor or after it
//            are out-of-range offsets and arrive as zeros
//   dz rows  a ring of NS stages of 32 rows
//   borders  a tap that leaves the image (or wraps into the neighbouring row or image) must
//            contribute nothing: the lane that supplies that pixel's row address to the
//            transposing read points it at a row of zeros instead
// Same swizzle, transposed reads, counted waits and ordered pixel slices as above.
struct WgradSlabArgs {
    WgradArgs w;
    int RX;                      // rows of the x ring
    int GW;                      // ceil((33 + W) / 8): row groups between a step's first pixel and its frontier
    unsigned x_bytes, dz_bytes;
};
constexpr int WS_NS = 3;         // dz stages = steps in flight + 1 (a fourth stage gains 2 % at 52 x 52 and costs the second workgroup per CU at 104 x 104)

// Transposing reads issued without waiting for them: the caller overlaps them with the MFMAs of
// the fragments read before and calls lds_wait_frags (LDS operations complete in order, so
// "at most N outstanding" says which fragments have arrived).
__device__ __forceinline__ void lds_issue_tr4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, uint2 *r) {
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                 "ds_read_b64_tr_b16 %1, %5\n\t"
                 "ds_read_b64_tr_b16 %2, %6\n\t"
                 "ds_read_b64_tr_b16 %3, %7"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                 : "memory");
}
__device__ __forceinline__ void lds_issue_tr2(unsigned a0, unsigned a1, uint2 *r) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\t"
                 "ds_read_b64_tr_b16 %1, %3"
                 : "=&v"(r[0]), "=&v"(r[1])
                 : "v"(a0), "v"(a1)
                 : "memory");
}
template <int OUTSTANDING>
__device__ __forceinline__ void lds_wait_frags2(uint2 *r) {     // r[0..1] are valid after this
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(OUTSTANDING) : "memory");
}
template <int OUTSTANDING>
__device__ __forceinline__ void lds_wait_frags(uint2 *r) {      // r[0..3] are valid after this
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(OUTSTANDING) : "memory");
}

// Chunk swizzle of the 128-byte rows: a transposing read is served in two groups of 32 lanes,
// i.e. rows s + {0..3} and s + {8..11} (s arbitrary: a tap shifts the rows), over 64 banks of
// 4 bytes.  Rows of equal parity share a half of the banks, so the four such rows of a group
// (s, s + 2, s + 8, s + 10 and the odd ones likewise) must put their 32-byte chunk in four
// different places: bit 1 and bit 3 of the row number tell them apart for every s.
__device__ __forceinline__ int ws_key(int row) { return ((row >> 1) & 1) | ((row >> 2) & 2); }

__global__ void __launch_bounds__(256, 2)
conv_wgrad_slab_kernel(const WgradSlabArgs sa) {
    const WgradArgs &a = sa.w;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int D = WS_NS - 1;                                 // steps in flight ahead of the one computed
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tile_ci = blockIdx.x % a.n_ci_tiles, tile_co = blockIdx.x / a.n_ci_tiles;
    const int ci0 = tile_ci * 64, co0 = tile_co * 64;
    const int c_begin = blockIdx.z * a.chunks_per_split;        // steps of 32 pixels
    int c_end = c_begin + a.chunks_per_split;
    const int nsteps = (a.P + 31) / 32;
    if (c_end > nsteps) c_end = nsteps;
    const i32x4_t xrsrc = make_rsrc(a.x, sa.x_bytes);
    const i32x4_t zrsrc = make_rsrc(a.dz, sa.dz_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const int RX = sa.RX, W = a.W;
    const unsigned dz0 = lds0 + (unsigned)(RX * 128);
    const unsigned zrow = dz0 + (unsigned)(WS_NS * 4096);        // 128 bytes of zeros

    // DMA geometry: an instruction moves eight rows of 128 bytes; lane l fills slot l & 7 of row l >> 3
    const int d_rl = lane >> 3, d_sl = lane & 7;
    auto issue_x_group = [&](int gi) {                           // input pixels 8 gi .. 8 gi + 7
        const int pi = gi * 8 + d_rl;
        const int pair = (d_sl >> 1) ^ ws_key(pi);
        const int ci = ci0 + (pair * 2 + (d_sl & 1)) * 8;
        unsigned off = OOB;
        if (pi >= 0 && pi < a.P && ci < a.Cin) off = (unsigned)((pi * a.x_ld + ci) * 2);
        lds_dma16(xrsrc, lds0 + (unsigned)(((gi * 8) & (RX - 1)) * 128), off);
    };
    auto issue_dz = [&](int step, int stage) {                   // this wave's eight rows of the step's 32
        const int row = wave * 8 + d_rl;
        const int p = step * 32 + row;
        const int pair = (d_sl >> 1) ^ ws_key(row);
        const int co = co0 + (pair * 2 + (d_sl & 1)) * 8;
        unsigned off = OOB;
        if (step < c_end && p < a.P && co < a.Cout) off = (unsigned)((p * a.dz_ld + co) * 2);
        lds_dma16(zrsrc, dz0 + (unsigned)(stage * 4096 + wave * 1024), off);
    };

    // wave w = input channels 16 w .. 16 w + 15 of the tile x all 64 output channels x 9 taps: the
    // x fragment, which is read once per tap, is a single 16-channel block per wave, and the dz
    // fragments (four blocks), read once per step, serve all nine taps
    f32x4_t acc[9][4];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tp][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // transposed-read geometry: this lane supplies the addresses of pixel rows r0 and r0 + 4 of the step
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int r0 = 8 * g + q4;
    unsigned ablk[9][2];                                         // LDS address of (tap, pixel row r0 / r0 + 4, this wave's block) inside slab row 0
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const int shift = (tp / 3 - 1) * W + (tp % 3 - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h)                              // steps start at multiples of 32: the key does not move
            ablk[tp][h] = lds0 + (unsigned)(((wave ^ ws_key(r0 + 4 * h + shift)) * 32) + p4 * 8);
    }
    unsigned boff[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) boff[j][h] = (unsigned)((r0 + 4 * h) * 128 + ((j ^ ws_key(r0 + 4 * h)) * 32) + p4 * 8);

    // a wave whose 16 input channels all lie beyond the layer's (the first layer has 8) only moves data
    const bool live = __builtin_amdgcn_readfirstlane((ci0 + wave * 16 < a.Cin) ? 1 : 0) != 0;

    // ---- prologue: the slab up to the frontier of step c_begin + D - 1, the dz rows of those steps ----
    if (t < 32) ((float *)(smem + RX * 128 + WS_NS * 4096))[t] = 0.0f;
    if (c_begin < c_end) {
        const int pb = c_begin * 32 - W - 1;
        const int g_first = pb >= 0 ? pb / 8 : -((-pb + 7) / 8);
        const int g_end = 4 * (c_begin + D - 1) + sa.GW;
        for (int gi = g_first + wave; gi < g_end; gi += 4) issue_x_group(gi);
#pragma unroll
        for (int i = 0; i < D; ++i) issue_dz(c_begin + i, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int stage = 0;
    for (int step = c_begin; step < c_end; ++step) {
        // next transfers: the four row groups of step + D (one per wave) and its dz rows
        issue_x_group(4 * (step + D - 1) + sa.GW + wave);
        {
            int ns = stage + D;
            ns = ns >= WS_NS ? ns - WS_NS : ns;
            issue_dz(step + D, ns);
        }
        if (live) {
        // which taps of this lane's two pixels stay inside the image
        // (kept as lane masks: the per-tap tests are then scalar ANDs and one select per address)
        bool up[2], down[2], left[2], right[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = step * 32 + r0 + 4 * h;
            const int b = fastdiv(p, a.ohw_m, a.ohw_s), rem = p - b * (a.H * W);
            const int oh = fastdiv(rem, a.ow_m, a.ow_s), ow = rem - oh * W;
            up[h] = oh > 0; down[h] = oh < a.H - 1; left[h] = ow > 0; right[h] = ow < W - 1;
        }
        const unsigned zb = dz0 + (unsigned)(stage * 4096);
        // ring positions in bytes: (row & (RX - 1)) * 128 == (row * 128) & ((RX - 1) * 128)
        const unsigned ring_mask = (unsigned)((RX - 1) * 128);
        const unsigned pb0 = (unsigned)((step * 32 + r0) * 128), pb1 = pb0 + 512u;
        auto issue_tap = [&](int tp, uint2 *r) {
            const int tr = tp / 3, tc = tp % 3;
            const unsigned shift_b = (unsigned)(((tr - 1) * W + (tc - 1)) * 128);       // wave-uniform
            bool ok0 = true, ok1 = true;
            if (tr == 0) { ok0 = ok0 && up[0]; ok1 = ok1 && up[1]; }
            if (tr == 2) { ok0 = ok0 && down[0]; ok1 = ok1 && down[1]; }
            if (tc == 0) { ok0 = ok0 && left[0]; ok1 = ok1 && left[1]; }
            if (tc == 2) { ok0 = ok0 && right[0]; ok1 = ok1 && right[1]; }
            const unsigned ra = ok0 ? ((pb0 + shift_b) & ring_mask) : zrow - lds0;     // ablk carries lds0
            const unsigned rb = ok1 ? ((pb1 + shift_b) & ring_mask) : zrow - lds0;
            lds_issue_tr2(ra + ablk[tp][0], rb + ablk[tp][1], r);
        };
        uint2 rb8[8], ra2[3][2];
        lds_issue_tr4(zb + boff[0][0], zb + boff[0][1], zb + boff[1][0], zb + boff[1][1], rb8);
        lds_issue_tr4(zb + boff[2][0], zb + boff[2][1], zb + boff[3][0], zb + boff[3][1], rb8 + 4);
        issue_tap(0, ra2[0]);
        issue_tap(1, ra2[1]);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            // the reads of the next two taps travel while this tap's products are formed
            if (tp < 7) {
                issue_tap(tp + 2, ra2[(tp + 2) % 3]);
                if (tp == 0) { lds_wait_frags<4>(rb8); lds_wait_frags<4>(rb8 + 4); }
                lds_wait_frags2<4>(ra2[tp % 3]);
            } else if (tp == 7) {
                lds_wait_frags2<2>(ra2[tp % 3]);
            } else {
                lds_wait_frags2<0>(ra2[tp % 3]);
            }
            const uint2 *r = ra2[tp % 3];
            const uint4 fa = make_uint4(r[0].x, r[0].y, r[1].x, r[1].y);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 fb = make_uint4(rb8[2 * j].x, rb8[2 * j].y, rb8[2 * j + 1].x, rb8[2 * j + 1].y);
                acc[tp][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, fb),
                                                                     acc[tp][j], 0, 0, 0);
            }
        }
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 2) : "memory");   // the next step's rows (this wave's part) have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage = stage + 1 == WS_NS ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        float *out = a.out + ((size_t)blockIdx.z * 9 + tp) * a.Cin * a.Cout;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wave * 16 + 4 * (lane >> 4) + r;
                const int co = co0 + j * 16 + (lane & 15);
                if (ci < a.Cin && co < a.Cout) out[(size_t)ci * a.Cout + co] = acc[tp][j][r];
            }
    }
}

static int wgrad_slab_rx(int W) {
    const int need = 32 * (WS_NS - 1) + 2 * W + 41;
    int rx = 128;
    while (rx < need) rx <<= 1;
    return rx;
}
static bool wgrad_use_slab(int k, int stride, int W, long long x_bytes, long long dz_bytes) {
    return g_wgrad_variant == 0 && k == 3 && stride == 1 && x_bytes < 0x7fffffffLL && dz_bytes < 0x7fffffffLL && wgrad_slab_rx(W) <= 1024;
}
static int wgrad_slab_splits(int Cin, int Cout, long long P, int W) {
    const long long base = (long long)((Cin + 63) / 64) * ((Cout + 63) / 64);
    // one full round of workgroups: two fit a CU by registers, one where the ring takes most of the LDS
    const long long slots = wgrad_slab_rx(W) > 512 ? 256 : 512;
    long long splits = (slots + base - 1) / base;
    const long long nsteps = (P + 31) / 32;
    if (splits > nsteps / 8) splits = nsteps / 8;                // at least 256 pixels per slice
    if (splits > 256) splits = 256;
    return (int)(splits < 1 ? 1 : splits);
}


// ---- weight gradient of 3x3 stride-2 layers: all nine taps from one staging of the input rows ------
// The per-tap kernels give every tap its own workgroups, so x and dz cross the L2 nine times, and
// with stride 2 each tap uses half of every 128-byte line it fetches: 32->64 @416 -> 208 moves
// 6 GB through the L2 for 1.06 GB of operands.  Here a workgroup owns 32 input x 64 output
// channels and all nine taps, and walks units of 64 consecutive output pixels of one output row.
// Per unit it stages, PLANAR (one run of pixels per channel, as in conv_wgrad_first_kernel), the
// three input rows the unit touches -- once per horizontal tap s, taking every second pixel, so
// that XS[r][s][ci][j] = x[2 oh + r - 1][2 (ow0 + j) + s - 1][ci] -- and the dz row segment; both MFMA
// operands are then aligned 16-byte LDS reads.  Two neighbouring output pixels are packed into one
// 32-bit LDS store; the 16-byte chunks of a plane row are XOR-swizzled by the channel group so that
// the lanes of a store instruction (four channel groups x consecutive pixel pairs) hit distinct banks.
// Wave w keeps the accumulators of (tap, channel half) blocks w, w + 4, ... (4-5 of the 18) x four
// output-channel blocks.  Units beyond a workgroup's share and pixels beyond the row are zeros.
struct WgradS2Args {
    const bf16_t *x, *dz;
    float *partial;             // [gridDim.x][9][Cin][Cout]
    int B, H, W, OH, OW, Cin, Cout, x_ld, dz_ld;
    int nseg, units;            // 64-pixel segments per output row; B * OH * nseg
};
constexpr int WS2_SEG = 64, WS2_TCI = 32, WS2_TCO = 64;

__global__ void __launch_bounds__(256)
conv_wgrad_s2_kernel(const WgradS2Args a) {
    __shared__ __attribute__((aligned(16))) char smem[(9 * WS2_TCI + WS2_TCO) * WS2_SEG * 2];
    char *XS = smem;                                              // [3 r][3 s][32 ci][64 j] bf16, rows of 128 bytes
    char *DZ = smem + 9 * WS2_TCI * WS2_SEG * 2;                  // [64 co][64 j]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ci0 = blockIdx.y * WS2_TCI, co0 = blockIdx.z * WS2_TCO;
    f32x4_t acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // fragment addresses: operand row = lane & 15 of the block, pixels 8 * (lane >> 4) .. + 7 of a 32-pixel step
    const int frow = lane & 15, kgrp = lane >> 4;
    // The global loads of unit u + 1 are issued (into registers) before unit u is computed, so that their
    // latency runs under the MFMAs and the LDS traffic of the current unit.
    constexpr int NTASK = 3 * 3 * 4 * 32, NQ = (NTASK + 255) / 256;    // x tasks = (input row r, tap column s, channel group c8, pixel pair jp)
    uint4 xa[NQ], xb[NQ], za, zb;
    auto fetch = [&](int u) {
        const int seg = u % a.nseg, rowid = u / a.nseg;
        const int oh = rowid % a.OH, b = rowid / a.OH;
        const int ow0 = seg * WS2_SEG;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = q * 256 + t;
            xa[q] = make_uint4(0u, 0u, 0u, 0u);
            xb[q] = xa[q];
            if (i < NTASK) {
                const int c8 = i & 3, jp = (i >> 2) & 31, rs = i >> 7, r = rs / 3, sx = rs - r * 3;
                const int ih = 2 * oh + r - 1, iw = 2 * (ow0 + 2 * jp) + sx - 1, ci = ci0 + c8 * 8;
                if (ih >= 0 && ih < a.H && ci < a.Cin) {
                    const bf16_t *px = a.x + (((size_t)b * a.H + ih) * a.W) * a.x_ld + ci;
                    if (iw >= 0 && iw < a.W && ow0 + 2 * jp < a.OW) xa[q] = *(const uint4 *)(px + (size_t)iw * a.x_ld);
                    if (iw + 2 < a.W && ow0 + 2 * jp + 1 < a.OW) xb[q] = *(const uint4 *)(px + (size_t)(iw + 2) * a.x_ld);
                }
            }
        }
        {
            const int c8 = t & 7, jp = t >> 3;                     // dz: 8 channel groups x 32 pixel pairs = one task per thread
            za = make_uint4(0u, 0u, 0u, 0u);
            zb = za;
            const int co = co0 + c8 * 8, ow = ow0 + 2 * jp;
            if (co < a.Cout) {
                const bf16_t *pz = a.dz + (((size_t)b * a.OH + oh) * a.OW) * a.dz_ld + co;
                if (ow < a.OW) za = *(const uint4 *)(pz + (size_t)ow * a.dz_ld);
                if (ow + 1 < a.OW) zb = *(const uint4 *)(pz + (size_t)(ow + 1) * a.dz_ld);
            }
        }
    };
    // two neighbouring pixels of eight channels -> eight 32-bit words, one per channel plane
    auto put = [&](char *base, const uint4 &va, const uint4 &vb) {
        const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *(uint32_t *)(base + (2 * e) * 128) = (wa[e] & 0xffffu) | (wb[e] << 16);
            *(uint32_t *)(base + (2 * e + 1) * 128) = (wa[e] >> 16) | (wb[e] & 0xffff0000u);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = q * 256 + t;
            if (i < NTASK) {
                const int c8 = i & 3, jp = (i >> 2) & 31, rs = i >> 7;
                // dword jp of the plane row, its 16-byte chunk (jp >> 2) swizzled by the channel group
                put(XS + (size_t)(rs * WS2_TCI + c8 * 8) * 128 + ((((jp >> 2) ^ c8) << 4) | ((jp & 3) << 2)), xa[q], xb[q]);
            }
        }
        const int c8 = t & 7, jp = t >> 3;
        put(DZ + (size_t)(c8 * 8) * 128 + ((((jp >> 2) ^ c8) << 4) | ((jp & 3) << 2)), za, zb);
    };
    if ((int)blockIdx.x < a.units) fetch(blockIdx.x);
    for (int u = blockIdx.x; u < a.units; u += gridDim.x) {
        __syncthreads();                                           // the previous unit's fragment reads are done
        stash();
        __syncthreads();
        if (u + (int)gridDim.x < a.units) fetch(u + gridDim.x);
#pragma unroll
        for (int step = 0; step < WS2_SEG / 32; ++step) {
            const int chunk = step * 4 + kgrp;
            uint4 fb[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int co = nb * 16 + frow;
                fb[nb] = *(const uint4 *)(DZ + co * 128 + ((chunk ^ (co >> 3)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int mb = wave + 4 * i;                       // (tap, channel half) block
                if (mb < 18) {
                    const int plane = mb * 16 + frow;              // = (tap * 32 + half * 16 + row)
                    const uint4 fa = *(const uint4 *)(XS + plane * 128 + ((chunk ^ ((plane >> 3) & 3)) << 4));
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, fb[nb]),
                                                                             acc[i][nb], 0, 0, 0);
                }
            }
        }
    }
    float *out = a.partial + (size_t)blockIdx.x * 9 * a.Cin * a.Cout;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int mb = wave + 4 * i;
        if (mb < 18) {
            const int tap = mb >> 1, half = mb & 1;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + half * 16 + 4 * (lane >> 4) + r, co = co0 + nb * 16 + (lane & 15);
                    if (ci < a.Cin && co < a.Cout) out[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[i][nb][r];
                }
        }
    }
}

static bool wgrad_use_s2(int k, int stride, int Cin, int Cout) {
    // the first stride-2 layer only (32 input channels, one channel tile): measured at batch 64, 32->64 @416->208
    // 0.94 -> 0.54 ms, but 64->128 @208->104 0.32 -> 0.56 (two channel tiles restage the rows, and the planar
    // staging's LDS stores, not memory, bound this kernel) and worse beyond; those stay on the per-tap kernels,
    // whose tensors the L2 / Infinity Cache hold
    return g_wgrad_variant == 0 && k == 3 && stride == 2 && Cin % 8 == 0 && Cout % 8 == 0 && Cin <= 32;
}
static int wgrad_s2_parts(int B, int OH, int OW, int Cin, int Cout) {
    const long long tiles = (long long)((Cin + WS2_TCI - 1) / WS2_TCI) * ((Cout + WS2_TCO - 1) / WS2_TCO);
    const long long units = (long long)B * OH * ((OW + WS2_SEG - 1) / WS2_SEG);
    long long gx = (768 + tiles - 1) / tiles;                       // three workgroups per CU in all
    if (gx > units) gx = units;
    return (int)(gx < 1 ? 1 : gx);
}

static int wgrad_tci(int Cin, bool dma) { return dma ? (Cin > 128 ? 256 : (Cin > 64 ? 128 : 64)) : (Cin > 64 ? 128 : 64); }
static bool wgrad_use_dma(long long x_bytes, long long dz_bytes) {
    return g_wgrad_variant != 1 && x_bytes < 0x7fffffffLL && dz_bytes < 0x7fffffffLL;
}
static int wgrad_splits(int k, int Cin, int Cout, long long P, bool dma) {
    const int tci = wgrad_tci(Cin, dma);
    const long long base = (long long)((Cin + tci - 1) / tci) * ((Cout + 127) / 128) * k * k;
    long long splits = ((dma ? 1024 : 1536) + base - 1) / base;  // four / six workgroups per CU in all
    const long long nchunks = (P + 63) / 64;
    if (splits > nchunks / 4) splits = nchunks / 4;              // at least 256 pixels per slice
    if (splits > 256) splits = 256;
    return (int)(splits < 1 ? 1 : splits);
}
static long long wgrad_bytes(long long pixels, int ld) { return pixels * ld * 2; }
size_t conv_wgrad_workspace_bytes(int k, int Cin, int Cout, long long P) {
    // the larger of the two kernels' needs (which one runs depends on the tensors' sizes and the option)
    const int s0 = wgrad_splits(k, Cin, Cout, P, false), s1 = wgrad_splits(k, Cin, Cout, P, true);
    int splits = s0 > s1 ? s0 : s1;
    if (k == 3) {
        const int s2 = wgrad_slab_splits(Cin, Cout, P, 1);          // the narrow-ring case has the most slices
        splits = s2 > splits ? s2 : splits;
    }
    return splits > 1 ? (size_t)splits * k * k * Cin * Cout * 4 : 0;
}
// the stride-2 all-taps kernel's partial results (it is chosen by shape, so the caller's query must cover it)
size_t conv_wgrad_s2_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    const int OH = (H + 1 - 3) / 2 + 1, OW = (W + 1 - 3) / 2 + 1;
    return (size_t)wgrad_s2_parts(B, OH, OW, Cin, Cout) * 9 * Cin * Cout * 4;
}

// Which of the three weight-gradient kernels launch_conv_wgrad takes for dense tensors of this
// shape under the calling thread's "wgrad_variant".
const char *conv_wgrad_label(int B, int H, int W, int Cin, int Cout, int k, int stride) {
    const int OH = stride == 1 ? H : (H + 1 - k) / 2 + 1, OW = stride == 1 ? W : (W + 1 - k) / 2 + 1;
    const long long xb = wgrad_bytes((long long)B * H * W, Cin), zb = wgrad_bytes((long long)B * OH * OW, Cout);
    if (wgrad_use_slab(k, stride, W, xb, zb)) return "wgrad_slab";
    if (wgrad_use_s2(k, stride, Cin, Cout)) return "wgrad_s2";
    return wgrad_use_dma(xb, zb) ? "wgrad_dma" : "wgrad_reg";
}

int launch_conv_wgrad(const void *x, int x_ld, const void *dz, int dz_ld, float *dw, int B, int H, int W, int Cin, int Cout, int k,
                      int stride, void *ws, hipStream_t s) {
    WgradArgs a;
    a.x = (const bf16_t *)x; a.dz = (const bf16_t *)dz;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.k = k; a.stride = stride;
    a.OH = stride == 1 ? H : (H + 1 - k) / 2 + 1;
    a.OW = stride == 1 ? W : (W + 1 - k) / 2 + 1;
    a.pad = stride == 1 ? (k - 1) / 2 : 1;                       // stride 2: one zero row / column on the top / left only
    a.x_ld = x_ld > 0 ? x_ld : Cin;
    a.dz_ld = dz_ld > 0 ? dz_ld : Cout;
    const long long P = (long long)B * a.OH * a.OW;
    if (P >= (1LL << 31) - 64) return fail(YMX_EUNSUPPORTED, "conv_wgrad: more than 2^31 output pixels");
    a.P = (int)P;
    set_fastdiv((unsigned)a.OW, a.ow_m, a.ow_s);
    set_fastdiv((unsigned)(a.OH * a.OW), a.ohw_m, a.ohw_s);
    const long long xb = wgrad_bytes((long long)B * H * W, a.x_ld), zb = wgrad_bytes(P, a.dz_ld);
    if (wgrad_use_slab(k, stride, W, xb, zb)) {
        WgradSlabArgs sa;
        const int splits = wgrad_slab_splits(Cin, Cout, P, W);
        const int nsteps = (int)((P + 31) / 32);
        a.chunks_per_split = (nsteps + splits - 1) / splits;
        a.n_ci_tiles = (Cin + 63) / 64;
        a.out = splits > 1 ? (float *)ws : dw;
        sa.w = a;
        sa.RX = wgrad_slab_rx(W);
        sa.GW = (33 + W + 7) / 8;
        sa.x_bytes = (unsigned)xb; sa.dz_bytes = (unsigned)zb;
        const int lds = sa.RX * 128 + WS_NS * 4096 + 128;
        static LdsReserve reserved;
        if (const int rc = reserve_lds(reserved, (const void *)conv_wgrad_slab_kernel, 144 * 1024, "conv_wgrad_slab")) return rc;
        hipLaunchKernelGGL(conv_wgrad_slab_kernel, dim3(a.n_ci_tiles * ((Cout + 63) / 64), 1, splits), dim3(256), lds, s, sa);
        int rc = check_launch("conv_wgrad_slab");
        if (rc || splits == 1) return rc;
        const size_t n = (size_t)9 * Cin * Cout;
        launch_sum_partials((const float *)ws, dw, n, splits, s);
        return check_launch("conv_wgrad_reduce");
    }
    if (wgrad_use_s2(k, stride, Cin, Cout)) {
        WgradS2Args sa;
        sa.x = a.x; sa.dz = a.dz; sa.partial = (float *)ws;
        sa.B = B; sa.H = H; sa.W = W; sa.OH = a.OH; sa.OW = a.OW; sa.Cin = Cin; sa.Cout = Cout; sa.x_ld = a.x_ld; sa.dz_ld = a.dz_ld;
        sa.nseg = (a.OW + WS2_SEG - 1) / WS2_SEG;
        sa.units = B * a.OH * sa.nseg;
        const int parts = wgrad_s2_parts(B, a.OH, a.OW, Cin, Cout);
        hipLaunchKernelGGL(conv_wgrad_s2_kernel, dim3(parts, (Cin + WS2_TCI - 1) / WS2_TCI, (Cout + WS2_TCO - 1) / WS2_TCO), dim3(256), 0, s, sa);
        int rc = check_launch("conv_wgrad_s2");
        if (rc) return rc;
        launch_sum_partials((const float *)ws, dw, (size_t)9 * Cin * Cout, parts, s);
        return check_launch("conv_wgrad_reduce");
    }
    const bool dma = wgrad_use_dma(xb, zb);
    const int splits = wgrad_splits(k, Cin, Cout, P, dma);
    const int tci = wgrad_tci(Cin, dma);
    a.n_ci_tiles = (Cin + tci - 1) / tci;
    a.out = splits > 1 ? (float *)ws : dw;
    const int gx = a.n_ci_tiles * ((Cout + 127) / 128);
    if (dma) {
        const int nsteps = (int)((P + 31) / 32);
        a.chunks_per_split = (nsteps + splits - 1) / splits;
#define YMX_WGRAD_DMA(T)                                                                                                  \
        {                                                                                                                     \
            auto kern = conv_wgrad_dma_kernel<T>;                                                                             \
            static LdsReserve reserved;                                                                                       \
            if (const int rc = reserve_lds(reserved, (const void *)kern, WgradGeom<T>::LDS_BYTES, "conv_wgrad")) return rc;    \
            hipLaunchKernelGGL(kern, dim3(gx, k * k, splits), dim3(256), WgradGeom<T>::LDS_BYTES, s, a, (unsigned)xb, (unsigned)zb); \
        }
        if (tci == 256) YMX_WGRAD_DMA(256)
        else if (tci == 128) YMX_WGRAD_DMA(128)
        else YMX_WGRAD_DMA(64)
#undef YMX_WGRAD_DMA
    } else {
        const int nchunks = (int)((P + 63) / 64);
        a.chunks_per_split = (nchunks + splits - 1) / splits;
        if (tci == 128) hipLaunchKernelGGL(conv_wgrad_kernel<128>, dim3(gx, k * k, splits), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(conv_wgrad_kernel<64>, dim3(gx, k * k, splits), dim3(256), 0, s, a);
    }
    int rc = check_launch("conv_wgrad");
    if (rc || splits == 1) return rc;
    const size_t n = (size_t)k * k * Cin * Cout;
    launch_sum_partials((const float *)ws, dw, n, splits, s);
    return check_launch("conv_wgrad_reduce");
}

// ---- weight gradient of the first layer (3 input channels) ------------------------------------------
// dW[tap][ci][co] = sum_p x[p + tap][ci] * dz[p][co] with 9 * 3 = 27 (tap, ci) rows and up to 64
// output channels: 864-1728 sums over ~10^7 pixels, i.e. a reduction that should cost the time of
// reading dz once.  The general kernels pad the image to 8 channels (a copy) and leave all but one
// wave of a workgroup without arithmetic.  Here a workgroup walks image rows; per row it stages,
// in LDS and PLANAR (one run of pixels per channel), the three input rows it touches -- once per
// horizontal tap offset, so that every fragment read is an aligned 16 bytes -- and the dz row
// transposed the same way; then both MFMA operands are plain 16-byte LDS reads:
//   A[m = (tap, ci)][k = pixel]   from XA[s][r][ci][col],  holding x[row + r - 1][col + s - 1][ci]
//   B[k = pixel][n = co]          from DZ[co][col]
// Each wave takes every fourth 32-pixel step of the row and keeps its own accumulators; every
// wave of every workgroup writes one partial result, added in order by sum_partials.
struct WgradFirstArgs {
    const void *x;              // [B][H][W][Cin] fp32 or bf16, dense
    const bf16_t *dz;           // [B][H][W][dz_ld] bf16
    float *partial;             // [gridDim.x * 4][9 * Cin * Cout]
    int B, H, W, Cin, Cout, dz_ld, Wp, x_f32;
};

template <int NB>               // 16-channel blocks of dz handled (Cout <= 16 * NB)
__global__ void __launch_bounds__(256)
conv_wgrad_first_kernel(const WgradFirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t *XA = (bf16_t *)smem;                                  // [3 s][3 r][3 ci][Wp]
    bf16_t *DZ = XA + 27 * a.Wp;                                  // [16 NB][Wp]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int Wp = a.Wp, W = a.W, Cin = a.Cin;
    const int M = 9 * Cin;
    f32x4_t acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // this lane's operand rows: A row m = mb * 16 + (lane & 15) -> (tap, ci) -> plane of XA
    unsigned aoff[2];
    bool alive[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int m = mb * 16 + (lane & 15);
        alive[mb] = m < M;
        const int tap = alive[mb] ? m / Cin : 0, ci = alive[mb] ? m - tap * Cin : 0;
        const int r = tap / 3, sx = tap - r * 3;
        aoff[mb] = (unsigned)((((sx * 3 + r) * 3 + ci) * Wp + 8 * (lane >> 4)) * 2);
    }
    const unsigned boff = (unsigned)(((lane & 15) * Wp + 8 * (lane >> 4)) * 2);
    // XA and DZ start as zeros: the columns [W, Wp) of every plane are never written below (the MFMAs
    // run over whole 32-pixel steps, so they must hold finite values, and dz's must be zero), nor are
    // the two border columns that stand for x[-1] and x[W]
    for (int i = t; i < (27 + 16 * NB) * Wp; i += 256) XA[i] = 0;
    const int nrows = a.B * a.H;
    const int row_elems = W * Cin;                                  // elements of one image row
    const int nch8 = a.Cout >> 3;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int b = row / a.H, ih = row - b * a.H;
        __syncthreads();                                           // the previous row's fragment reads (or the zero fill) are done
        // x planes, source driven: the three image rows are read once, coalesced, eight elements per thread in
        // flight; element (r, col, ci) goes to XA[s][r][ci][col - s + 1] for the three horizontal tap offsets s
        for (int i0 = 0; i0 < 3 * row_elems; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256 + t;
                v[u] = 0.0f;
                if (i < 3 * row_elems) {
                    const int r = i / row_elems, e = i - r * row_elems, yy = ih + r - 1;
                    if (yy >= 0 && yy < a.H) {
                        const size_t g = ((size_t)b * a.H + yy) * row_elems + e;
                        v[u] = a.x_f32 ? ((const float *)a.x)[g] : bf16_to_f32(((const bf16_t *)a.x)[g]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256 + t;
                if (i < 3 * row_elems) {
                    const int r = i / row_elems, e = i - r * row_elems;
                    const int col = e / Cin, ci = e - col * Cin;
                    const bf16_t q = f32_to_bf16(v[u]);
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) {
                        const int c = col - sx + 1;
                        if (c >= 0 && c < W) XA[((sx * 3 + r) * 3 + ci) * Wp + c] = q;
                    }
                }
            }
        }
        // dz row: 16-byte chunks (8 channels of one pixel), four per thread in flight, scattered into the channel
        // planes; consecutive lanes take consecutive pixels of one channel group, so the 2-byte stores of a wave
        // fill consecutive LDS words
        const bf16_t *zrow = a.dz + ((size_t)b * a.H + ih) * W * a.dz_ld;
        for (int i0 = 0; i0 < W * nch8; i0 += 256 * 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 + t;
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (i < W * nch8) {
                    const int c8 = i / W, px = i - c8 * W;
                    v[u] = *(const uint4 *)(zrow + (size_t)px * a.dz_ld + c8 * 8);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 + t;
                if (i < W * nch8) {
                    const int c8 = i / W, px = i - c8 * W;
                    const uint32_t w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        DZ[(c8 * 8 + 2 * e) * Wp + px] = (bf16_t)(w4[e] & 0xffffu);
                        DZ[(c8 * 8 + 2 * e + 1) * Wp + px] = (bf16_t)(w4[e] >> 16);
                    }
                }
            }
        }
        __syncthreads();
        for (int col0 = wave * 32; col0 < Wp; col0 += 128) {
            uint4 fa[2], fb[NB];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                fa[mb] = alive[mb] ? *(const uint4 *)(smem + aoff[mb] + col0 * 2) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) fb[nb] = *(const uint4 *)((const char *)DZ + boff + (nb * 16 * Wp + col0) * 2);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[mb]), __builtin_bit_cast(bf16x8_t, fb[nb]),
                                                                          acc[mb][nb], 0, 0, 0);
        }
    }
    float *out = a.partial + ((size_t)blockIdx.x * 4 + wave) * M * a.Cout;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb * 16 + 4 * (lane >> 4) + r, n = nb * 16 + (lane & 15);
                if (m < M && n < a.Cout) out[(size_t)m * a.Cout + n] = acc[mb][nb][r];
            }
}

static int wgrad_first_grid(int B, int H) { return std::min(B * H, 768); }        // three workgroups per CU
bool conv_wgrad_first_supported(int B, int H, int W, int Cin, int Cout) {
    const int Wp = (W + 31) / 32 * 32, nb = (Cout + 15) / 16;
    return Cin >= 1 && Cin <= 3 && Cout % 8 == 0 && Cout <= 64 && (long long)B * H < (1LL << 30) && (27 + 16 * nb) * Wp * 2 <= 52 * 1024;
}
size_t conv_wgrad_first_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (!conv_wgrad_first_supported(B, H, W, Cin, Cout)) return 0;
    return (size_t)wgrad_first_grid(B, H) * 4 * 9 * Cin * Cout * 4;
}
int launch_conv_wgrad_first(const void *x, int x_f32, const void *dz, int dz_ld, float *dw, int B, int H, int W, int Cin, int Cout, void *ws,
                            hipStream_t s) {
    WgradFirstArgs a;
    a.x = x; a.dz = (const bf16_t *)dz; a.partial = (float *)ws;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.dz_ld = dz_ld > 0 ? dz_ld : Cout; a.x_f32 = x_f32;
    a.Wp = (W + 31) / 32 * 32;
    const int nb = (Cout + 15) / 16, grid = wgrad_first_grid(B, H);
    const int lds = (27 + 16 * nb) * a.Wp * 2;
    if (nb <= 2) hipLaunchKernelGGL(conv_wgrad_first_kernel<2>, dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(conv_wgrad_first_kernel<4>, dim3(grid), dim3(256), lds, s, a);
    int rc = check_launch("conv_wgrad_first");
    if (rc) return rc;
    launch_sum_partials((const float *)ws, dw, (size_t)9 * Cin * Cout, grid * 4, s);
    return check_launch("conv_wgrad_first_reduce");
}

// ---- adjoint of the bilinear x2 resize --------------------------------------------------------
// Forward (elementwise.hip): src = (o + 0.5) / 2 - 0.5, lo = max(floor(src), 0), hi = min(ceil(src), n - 1),
// y[o] = x[lo] * (1 - f) + x[hi] * f, f = src - floor(src).  Input i therefore receives from outputs
// 2 i - 1 .. 2 i + 2 the weights this function returns (both roles added: at the borders lo == hi).
__device__ __forceinline__ float up_weight(int o, int i, int n) {
    const float src = ((float)o + 0.5f) * 0.5f - 0.5f;
    const float fl = floorf(src);
    const int lo = max((int)fl, 0), hi = min((int)ceilf(src), n - 1);
    const float f = src - fl;
    return (lo == i ? 1.0f - f : 0.0f) + (hi == i ? f : 0.0f);
}

template <typename T>
__global__ void __launch_bounds__(256)
upsample2x_bwd_kernel(const T *__restrict__ dy, T *__restrict__ dx, int B, int H, int W, int C, int dy_ld, int c_off) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        size_t r = idx / C;
        const int j = (int)(r % W);
        r /= W;
        const int i = (int)(r % H), b = (int)(r / H);
        float acc = 0.0f;
        for (int oy = 2 * i - 1; oy <= 2 * i + 2; ++oy) {
            if (oy < 0 || oy >= 2 * H) continue;
            const float wy = up_weight(oy, i, H);
            if (wy == 0.0f) continue;
            for (int ox = 2 * j - 1; ox <= 2 * j + 2; ++ox) {
                if (ox < 0 || ox >= 2 * W) continue;
                const float wx = up_weight(ox, j, W);
                if (wx == 0.0f) continue;
                acc += wy * wx * ld_f(dy, (((size_t)b * 2 * H + oy) * 2 * W + ox) * dy_ld + c_off + c);
            }
        }
        st_f(dx, idx, acc);
    }
}

// bf16, eight channels per thread: the same sum in the same order (oy outer, ox inner) with 16-byte loads
__global__ void __launch_bounds__(256)
upsample2x_bwd_bf16x8_kernel(const bf16_t *__restrict__ dy, bf16_t *__restrict__ dx, int B, int H, int W, int C8, int dy_ld, int c_off) {
    const size_t total = (size_t)B * H * W * C8;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c8 = (int)(idx % C8);
        size_t r = idx / C8;
        const int j = (int)(r % W);
        r /= W;
        const int i = (int)(r % H), b = (int)(r / H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int oy = 2 * i - 1; oy <= 2 * i + 2; ++oy) {
            if (oy < 0 || oy >= 2 * H) continue;
            const float wy = up_weight(oy, i, H);
            if (wy == 0.0f) continue;
            for (int ox = 2 * j - 1; ox <= 2 * j + 2; ++ox) {
                if (ox < 0 || ox >= 2 * W) continue;
                const float wx = up_weight(ox, j, W);
                if (wx == 0.0f) continue;
                const uint4 v = *(const uint4 *)(dy + (((size_t)b * 2 * H + oy) * 2 * W + ox) * dy_ld + c_off + c8 * 8);
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                const float wgt = wy * wx;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += wgt * __uint_as_float(w4[e] << 16);
                    acc[2 * e + 1] += wgt * __uint_as_float(w4[e] & 0xffff0000u);
                }
            }
        }
        *(uint4 *)(dx + idx * 8) = make_uint4(pack2_bf16(acc[0], acc[1]), pack2_bf16(acc[2], acc[3]), pack2_bf16(acc[4], acc[5]), pack2_bf16(acc[6], acc[7]));
    }
}

int launch_upsample2x_bwd(const void *dy, void *dx, int B, int H, int W, int C, int dy_ld, int c_off, int dtype, hipStream_t s) {
    const size_t total = (size_t)B * H * W * C;
    if (dtype == YMX_BF16 && C % 8 == 0 && dy_ld % 8 == 0 && c_off % 8 == 0 && aligned16(dy) && aligned16(dx)) {
        const int grid8 = (int)std::min<size_t>((total / 8 + 255) / 256, 1 << 16);
        hipLaunchKernelGGL(upsample2x_bwd_bf16x8_kernel, dim3(grid8), dim3(256), 0, s, (const bf16_t *)dy, (bf16_t *)dx, B, H, W, C / 8, dy_ld, c_off);
        return check_launch("upsample2x_bwd");
    }
    const int grid = (int)std::min<size_t>((total + 255) / 256, 1 << 16);
    if (dtype == YMX_BF16) hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)dy, (bf16_t *)dx, B, H, W, C, dy_ld, c_off);
    else hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)dy, (float *)dx, B, H, W, C, dy_ld, c_off);
    return check_launch("upsample2x_bwd");
}

}  // namespace ymx
