// Fused implicit-GEMM convolution on MFMA for gfx950.
//
//   y[p, co] = act( sum_k X[p, k] * Wt[co, k] * scale[co] + shift[co] ) + res[p, co]
//
// p  = output pixel (b, oh, ow) flattened, NHWC activations
// k  = (r, c, ci) flattened, ci fastest; the packed weights are [Cout_pad][Kw] rows of k
// X[p, k] = x[b, oh*stride + r - pad, ow*stride + c - pad, ci]  (zero outside the image)
//
// The MFMA runs "weights as A, pixels as B": D[row = co][col = pixel].  With the
// 16x16 C/D register map (col = lane & 15, row = 4*(lane >> 4) + reg) every lane then
// owns 4 consecutive output channels of one pixel, so the NHWC epilogue loads bias /
// BN vectors and the residual, and stores the result, as 8- or 16-byte vectors with
// no shuffle and no LDS round trip.
//
// Tiles: a workgroup of 4 waves computes TP pixels x TQ channels; each wave 64 pixels
// x (TQ / WGQ) channels as 16x16 MFMA tiles.  K advances 128 bytes per row per step
// (64 bf16 or 32 f32).  Both operand tiles are staged through LDS in rows of 128 B,
// with the 16-byte chunk index XOR-swizzled by (row & 7) so that the ds_write_b128
// of the staging pass and the ds_read_b128 of the fragment reads are conflict-free.
//
// Pipeline: register staging, two LDS buffers, two register sets.  In step t the
// global loads of tile t+2 are issued first, then the MFMAs of tile t run from LDS,
// then tile t+1 (loaded during step t-1) is written to the other LDS buffer, then one
// barrier.  A load therefore has two full steps to land before anything waits for it.
// Activation loads are raw buffer loads: a lane whose tap falls outside the image
// (or beyond K) uses an out-of-range offset and the hardware returns zeros, so there
// is no branch and no select between the load and its use.
#include "common.h"

namespace ymx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <typename T, typename OT, int TP, int TQ, int WGP>
__global__ void __launch_bounds__(256, 2)
conv_igemm_kernel(const ConvArgs a, const int ntile_q, const int nk) {
    constexpr int KCH = 16 / (int)sizeof(T);    // elements per 16-byte chunk
    constexpr int BK = 8 * KCH;                 // K elements per step (128 B per row)
    constexpr int WGQ = 4 / WGP;
    constexpr int WP = TP / WGP, WQ = TQ / WGQ;
    constexpr int NI = WQ / 16, NJ = WP / 16;
    constexpr int NXL = TP / 32, NWL = TQ / 32; // 16-byte loads per thread per step
    constexpr int XBYTES = TP * 128, STAGE = (TP + TQ) * 128;
    constexpr unsigned OOB = 0x80000000u;       // beyond any tensor this kernel accepts
    static_assert(WP % 16 == 0 && WQ % 16 == 0 && NXL >= 1 && NWL >= 1, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wp = wave % WGP, wq = wave / WGP;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tq = id % ntile_q, tp = id / ntile_q;
    const int p0 = tp * TP, q0 = tq * TQ;
    const T *__restrict__ w = (const T *)a.w;
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);

    // ---- per-thread staging geometry: row (t>>3)+32*i of the tile, chunk t&7 ----
    const int lrow = t >> 3, chunk = t & 7;
    int xoff[NXL];          // element offset of (b, ih0, iw0, 0); may be negative
    unsigned xmask[NXL];    // bit (r*k+c): tap (r,c) of this pixel lies inside the image
#pragma unroll
    for (int i = 0; i < NXL; ++i) pixel_geometry(a, p0 + lrow + 32 * i, xoff[i], xmask[i]);
    const T *wsrc = w + (size_t)(q0 + lrow) * a.Kw + chunk * KCH;
    const int st_off = lrow * 128 + ((chunk ^ (lrow & 7)) << 4);

    struct Regs {
        u32x4_t x[NXL];
        uint4 w[NWL];
    };
    Regs ra, rb;

    auto gload = [&](Regs &r, int kt) {
        int tapoff;
        unsigned tapbit;
        k_decode(a, kt * BK + chunk * KCH, tapoff, tapbit);     // this lane's tap and channel in tile kt
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const bool ok = (xmask[i] & tapbit) != 0;
            const unsigned off = ok ? (unsigned)(xoff[i] + tapoff) * (unsigned)sizeof(T) : OOB;
            r.x[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
        }
        const int ktw = kt < nk ? kt : nk - 1;      // tiles past the end re-read the last column
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            r.w[i] = *(const uint4 *)(wsrc + (size_t)i * 32 * a.Kw + (size_t)ktw * BK);
    };
    auto lstore = [&](const Regs &r, int buf) {
        char *xs = smem + buf * STAGE + st_off;
#pragma unroll
        for (int i = 0; i < NXL; ++i) *(u32x4_t *)(xs + i * 4096) = r.x[i];
#pragma unroll
        for (int i = 0; i < NWL; ++i) *(uint4 *)(xs + XBYTES + i * 4096) = r.w[i];
    };

    f32x4_t acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int a_off = XBYTES + (wq * WQ + (lane & 15)) * 128;
    const int b_off = (wp * WP + (lane & 15)) * 128;
    const int swz0 = (((lane >> 4)) ^ (lane & 7)) << 4;
    const int swz1 = ((4 + (lane >> 4)) ^ (lane & 7)) << 4;

    auto compute = [&](int buf) {
        const char *base = smem + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int swz = kk ? swz1 : swz0;
            uint4 af[NI], bfr[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) af[i] = *(const uint4 *)(base + a_off + i * 2048 + swz);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = *(const uint4 *)(base + b_off + j * 2048 + swz);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]),
                            acc[i][j], 0, 0, 0);
                    } else {
                        // each lane holds k = 4*(lane>>4) .. +3 of this 16-deep slice; MFMA e
                        // consumes element e of both operands, so the four MFMAs cover the
                        // slice's 16 k exactly once.
                        const f32x4_t av = __builtin_bit_cast(f32x4_t, af[i]);
                        const f32x4_t bv = __builtin_bit_cast(f32x4_t, bfr[j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };
    // One pipeline step: tile kt is in LDS buffer kt&1, tile kt+1 in `next`, `far` is free.
    // Every step issues the same loads and stores, needed or not (tiles past the end
    // read zeros / a clamped weight column and land in an LDS buffer nobody reads), so
    // the number of loads in flight at each wait is a compile-time constant and the
    // compiler can wait for exactly the older register set.
    auto step = [&](Regs &far, const Regs &next, int kt) {
        gload(far, kt + 2);
        compute(kt & 1);
        lstore(next, (kt & 1) ^ 1);
        __syncthreads();
    };

    gload(ra, 0);
    gload(rb, 1);
    lstore(ra, 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        step(ra, rb, kt);
        step(rb, ra, kt + 1);
    }
    if (kt < nk) compute(kt & 1);

    // ---- epilogue: scale/shift, activation, shortcut, store ----------------------
    OT *__restrict__ y = (OT *)a.y;
    const T *__restrict__ res = (const T *)a.res;
    const bool vec_ok = ((a.Cout & 3) == 0) && ((a.y_ld & 3) == 0) && (res == nullptr || (a.res_ld & 3) == 0);
    const int co_base = q0 + wq * WQ + 4 * (lane >> 4);
    const int px_base = p0 + wp * WP + (lane & 15);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int co = co_base + i * 16;
        float sc[4], sh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = co + r;
            const bool in = c < a.Cout;
            sc[r] = in ? (a.scale ? a.scale[c] : 1.0f) : 0.0f;
            sh[r] = (in && a.shift) ? a.shift[c] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int p = px_base + j * 16;
            if (p >= a.P || co >= a.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_f(affine_f(acc[i][j][r], sc[r], sh[r]), a.act);
            if (vec_ok) {
                if (res) {
                    const T *rp = res + (size_t)p * a.res_ld + co;
                    if constexpr (sizeof(T) == 2) {
                        const uint2 rv = *(const uint2 *)rp;
                        v[0] += __uint_as_float(rv.x << 16);
                        v[1] += __uint_as_float(rv.x & 0xffff0000u);
                        v[2] += __uint_as_float(rv.y << 16);
                        v[3] += __uint_as_float(rv.y & 0xffff0000u);
                    } else {
                        const float4 rv = *(const float4 *)rp;
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    }
                }
                OT *yp = y + conv_out_offset(a, p, co);
                if constexpr (sizeof(OT) == 2) {
                    *(uint2 *)yp = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
                } else {
                    *(float4 *)yp = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (co + r < a.Cout) {
                        float o = v[r];
                        if (res) o += elem<T>::load(res + (size_t)p * a.res_ld + co + r);
                        elem<OT>::store(y + conv_out_offset(a, p, co + r), o);
                    }
                }
            }
        }
    }
}

template <typename T, typename OT, int TP, int TQ, int WGP>
static int launch_cfg(const ConvArgs &a, hipStream_t s) {
    constexpr int KCH = 16 / (int)sizeof(T), BK = 8 * KCH;
    constexpr int LDS = 2 * (TP + TQ) * 128;
    auto kern = conv_igemm_kernel<T, OT, TP, TQ, WGP>;
    static LdsReserve reserved;        // one per kernel instantiation; remembers the devices already set up
    if (const int rc = reserve_lds(reserved, (const void *)kern, LDS, "conv_igemm")) return rc;
    const int ntile_p = (a.P + TP - 1) / TP;
    const int ntile_q = (a.Cout + TQ - 1) / TQ;
    const int nk = (a.K + BK - 1) / BK;
    const long long grid = (long long)ntile_p * ntile_q;
    if (grid > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_igemm: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, s, a, ntile_q, nk);
    return check_launch("conv_igemm");
}

template <typename T, typename OT>
static int launch_shape(const ConvArgs &a, hipStream_t s) {
    if (a.Cout <= 32) return launch_cfg<T, OT, 256, 32, 4>(a, s);
    if (a.Cout <= 64) return launch_cfg<T, OT, 256, 64, 4>(a, s);
    return launch_cfg<T, OT, 128, 128, 2>(a, s);
}

int launch_conv_igemm(const ConvArgs &a, hipStream_t s) {
    if (a.dtype == YMX_BF16 && a.out_dtype == YMX_BF16) return launch_shape<bf16_t, bf16_t>(a, s);
    if (a.dtype == YMX_BF16 && a.out_dtype == YMX_F32) return launch_shape<bf16_t, float>(a, s);
    if (a.dtype == YMX_F32 && a.out_dtype == YMX_F32) return launch_shape<float, float>(a, s);
    return fail(YMX_EUNSUPPORTED, "conv_igemm: dtype %d -> out_dtype %d not supported", a.dtype, a.out_dtype);
}

}  // namespace ymx
