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
    // d/dz [z * tanh(softplus(z))] = t + z * s * (1 - t^2),  t = tanh(softplus(z)) = n / (n + 2),
    // n = e (e + 2), e = exp(z), s = sigmoid(z) = e / (1 + e); -> 1 for large z, -> 0 for very negative z
    if (z > 20.0f) return 1.0f;
    const float e = __expf(z);
    const float n = e * (e + 2.0f);
    const float t = n / (n + 2.0f);
    const float s = e / (1.0f + e);
    return t + z * s * (1.0f - t * t);
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

int launch_act(const void *z, const void *other, void *y, size_t n, int act, int dtype, int mode, hipStream_t s) {
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
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float acc = 0.0f;
        for (int s = 0; s < parts; ++s) acc += partial[(size_t)s * n + i];
        out[i] = acc;
    }
}

static int colsum_parts(long long P) {
    long long parts = (P + 511) / 512;
    return (int)(parts > 1024 ? 1024 : (parts < 1 ? 1 : parts));
}
size_t bias_grad_workspace_bytes(long long P, int C) { return (size_t)colsum_parts(P) * C * 4; }

int launch_bias_grad(const void *dz, long long P, int C, int ld, int dtype, float *db, void *ws, hipStream_t s) {
    const int parts = colsum_parts(P);
    const long long rows_per = (P + parts - 1) / parts;
    float *partial = (float *)ws;
    if (dtype == YMX_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(parts), dim3(256), 0, s, (const bf16_t *)dz, partial, P, C, ld, rows_per);
    else hipLaunchKernelGGL(colsum_kernel<float>, dim3(parts), dim3(256), 0, s, (const float *)dz, partial, P, C, ld, rows_per);
    int rc = check_launch("bias_grad");
    if (rc) return rc;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partial, db, (size_t)C, parts);
    return check_launch("bias_grad_reduce");
}

// ---- weight gradient --------------------------------------------------------------------------
// GEMM  dW[ci][co] (per tap) = sum_p X[p + tap][ci] * dZ[p][co]: both operands have the
// reduction index (pixels) as their SLOW axis in memory, the opposite of what an MFMA lane
// wants (8 consecutive reduction elements).  Tiles are therefore staged in LDS as they lie in
// memory, [pixel][channel] rows, and read with the transposing LDS read of gfx950
// (ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns arrives column-major),
// two reads per operand fragment.
// Workgroup = 4 waves = tile of 64 input channels x 128 output channels of one tap and one
// slice of the pixels; wave w owns output channels [32 w, 32 w + 32).  Per step 64 pixels:
// the next step's rows are loaded into registers while the MFMAs of the current one run.
// Slices (grid.z) write separate partial results that a second kernel adds in order, so the
// result is reproducible.
struct WgradArgs {
    const bf16_t *x, *dz;
    float *out;                 // [splits][k*k][Cin][Cout]
    int B, H, W, Cin, OH, OW, Cout, k, stride, pad;
    int x_ld, dz_ld;
    long long P;                // B * OH * OW
    long long chunks_per_split; // 64-pixel chunks per slice
    int n_ci_tiles;
};

constexpr int WG_PA = 64 * 2 + 16;       // LDS row pitch of the X tile: 64 channels + 16 bytes
constexpr int WG_PB = 128 * 2 + 16;      // dZ tile: 128 channels + 16 bytes

// Eight consecutive reduction rows of one column per lane: two transposing reads (rows
// 0-3 and 4-7 of the lane group's block, PITCH bytes apart) and the wait for both.
template <int PITCH>
__device__ __forceinline__ uint4 lds_read_tr8(const char *p) {
    uint2 lo, hi;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\t"
                 "ds_read_b64_tr_b16 %1, %2 offset:%3\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(addr), "n"(4 * PITCH)
                 : "memory");
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

__global__ void __launch_bounds__(256)
conv_wgrad_kernel(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) char sA[64 * WG_PA];
    __shared__ __attribute__((aligned(16))) char sB[64 * WG_PB];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tile_ci = blockIdx.x % a.n_ci_tiles, tile_co = blockIdx.x / a.n_ci_tiles;
    const int tap = blockIdx.y, tr = tap / a.k, tc = tap - tr * a.k;
    const int ci0 = tile_ci * 64, co0 = tile_co * 128;
    const long long c_begin = (long long)blockIdx.z * a.chunks_per_split;
    long long c_end = c_begin + a.chunks_per_split;
    const long long nchunks = (a.P + 63) / 64;
    if (c_end > nchunks) c_end = nchunks;

    f32x4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // staging geometry: X tile = 64 rows x 8 sixteen-byte items (2 per thread), dZ tile = 64 x 16 (4 per thread)
    uint4 ra[2], rb[4];
    auto load_chunk = [&](long long chunk) {
        const long long p0 = chunk * 64;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = t + u * 256, row = e >> 3, it = e & 7;
            const long long p = p0 + row;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int ci = ci0 + it * 8;
            if (p < a.P && ci < a.Cin) {
                const int ow = (int)(p % a.OW);
                const long long q = p / a.OW;
                const int oh = (int)(q % a.OH), b = (int)(q / a.OH);
                const int ih = oh * a.stride + tr - a.pad, iw = ow * a.stride + tc - a.pad;
                if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                    v = *(const uint4 *)(a.x + (((size_t)b * a.H + ih) * a.W + iw) * a.x_ld + ci);
            }
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + u * 256, row = e >> 4, it = e & 15;
            const long long p = p0 + row;
            const int co = co0 + it * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (p < a.P && co < a.Cout) v = *(const uint4 *)(a.dz + (size_t)p * a.dz_ld + co);
            rb[u] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = t + u * 256, row = e >> 3, it = e & 7;
            *(uint4 *)(sA + row * WG_PA + it * 16) = ra[u];
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
    const int a_off = (8 * g + q4) * WG_PA + p4 * 8;
    const int b_off = (8 * g + q4) * WG_PB + (wave * 32) * 2 + p4 * 8;

    if (c_begin < c_end) load_chunk(c_begin);
    for (long long chunk = c_begin; chunk < c_end; ++chunk) {
        __syncthreads();                       // the previous step's reads are done
        store_chunk();
        __syncthreads();
        if (chunk + 1 < c_end) load_chunk(chunk + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {       // two 32-pixel reduction steps per chunk
            uint4 fa[4], fb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = lds_read_tr8<WG_PA>(sA + ks * 32 * WG_PA + a_off + i * 32);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb[j] = lds_read_tr8<WG_PB>(sB + ks * 32 * WG_PB + b_off + j * 32);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[i]),
                                                                         __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
        }
    }
    // D[row = input channel][col = output channel]: lane holds rows 4 (lane >> 4) + r of column lane & 15
    float *out = a.out + ((size_t)blockIdx.z * a.k * a.k + tap) * a.Cin * a.Cout;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + i * 16 + 4 * (lane >> 4) + r;
                const int co = co0 + wave * 32 + j * 16 + (lane & 15);
                if (ci < a.Cin && co < a.Cout) out[(size_t)ci * a.Cout + co] = acc[i][j][r];
            }
}

static int wgrad_splits(int k, int Cin, int Cout, long long P) {
    const long long base = (long long)((Cin + 63) / 64) * ((Cout + 127) / 128) * k * k;
    long long splits = (2048 + base - 1) / base;                 // about eight workgroups per CU in all
    const long long nchunks = (P + 63) / 64;
    if (splits > nchunks / 4) splits = nchunks / 4;              // at least four chunks per slice
    if (splits > 256) splits = 256;
    return (int)(splits < 1 ? 1 : splits);
}
size_t conv_wgrad_workspace_bytes(int k, int Cin, int Cout, long long P) {
    const int splits = wgrad_splits(k, Cin, Cout, P);
    return splits > 1 ? (size_t)splits * k * k * Cin * Cout * 4 : 0;
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
    a.P = (long long)B * a.OH * a.OW;
    const int splits = wgrad_splits(k, Cin, Cout, a.P);
    const long long nchunks = (a.P + 63) / 64;
    a.chunks_per_split = (nchunks + splits - 1) / splits;
    a.n_ci_tiles = (Cin + 63) / 64;
    a.out = splits > 1 ? (float *)ws : dw;
    const int gx = a.n_ci_tiles * ((Cout + 127) / 128);
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(gx, k * k, splits), dim3(256), 0, s, a);
    int rc = check_launch("conv_wgrad");
    if (rc || splits == 1) return rc;
    const size_t n = (size_t)k * k * Cin * Cout;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65536)), dim3(256), 0, s, (const float *)ws, dw, n, splits);
    return check_launch("conv_wgrad_reduce");
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

int launch_upsample2x_bwd(const void *dy, void *dx, int B, int H, int W, int C, int dy_ld, int c_off, int dtype, hipStream_t s) {
    const size_t total = (size_t)B * H * W * C;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 1 << 16);
    if (dtype == YMX_BF16) hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)dy, (bf16_t *)dx, B, H, W, C, dy_ld, c_off);
    else hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)dy, (float *)dx, B, H, W, C, dy_ld, c_off);
    return check_launch("upsample2x_bwd");
}

}  // namespace ymx
