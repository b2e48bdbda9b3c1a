// bf16 implicit-GEMM convolution, LDS-DMA staging, several workgroups per CU (gfx950).
//
// Same mathematics, operand orientation (weights = MFMA A operand, pixels = B, D[row =
// channel][col = pixel]) and zero-padding-by-out-of-range-offset as conv_igemm_dma.hip.
// The geometry is chosen so that at least TWO independent workgroups fit on a CU: while
// one sits in its prologue, at its barrier or in its epilogue, the other's waves keep
// the CU busy.  That matters most for the narrow, short-K layers of the backbone's first
// stages, which are bound by memory latency and by their epilogue, not by the MFMAs.
//
//   workgroup      4 waves (256 threads); tile TQ channels x TP pixels with
//                  TQ = 256: 2 x 2 waves of 128 ch x 64 px, TP = 128
//                  TQ <= 128: 1 x 4 waves of TQ ch x 64 px, TP = 256
//   K step         32 (one mfma_f32_16x16x32_bf16 deep): LDS rows of 64 B; a 3x3 conv
//                  with 32 input channels is exactly one tap per step
//   LDS            3 stages x (TP + max(TQ, 64)) rows x 64 B  (60 .. 72 KB)
//   pipeline       step s: issue DMA of tile s+2, MFMAs of tile s, wait for tile s+1
//                  (counted vmcnt: the pieces just issued may stay in flight), one
//                  barrier.  A tile therefore has two full steps to land.
//   epilogue       per-wave transpose through LDS (fp32, XOR-swizzled), bias/BN,
//                  activation, shortcut, one rounding, 128 (or 64) contiguous bytes per
//                  pixel per store instruction
//
// LDS image of one DMA piece (one wave-instruction, 1 KB) = 16 rows x 64 B.  The 16-byte
// slot of K-chunk c in row r is  c ^ ((-(r >> 2)) & 3): with that XOR the ds_read_b128
// fragment reads (16 rows x one chunk per 16-lane group) touch 16 distinct slots of the
// 256-byte bank row.  The swizzle is applied to the DMA's source address (the lane filling
// slot s of row r fetches chunk s ^ f(r)), the LDS image itself stays linear.
#include "lds_dma.h"

namespace ymx {

template <int N> struct vmcnt_wait;
template <> struct vmcnt_wait<5> { __device__ static __forceinline__ void run() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); } };
template <> struct vmcnt_wait<6> { __device__ static __forceinline__ void run() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } };
template <> struct vmcnt_wait<8> { __device__ static __forceinline__ void run() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); } };

template <typename OT, int TQ>
__global__ void __launch_bounds__(256, 2)
conv_igemm_dma2_kernel(const ConvArgs a, const int ntile_q, const int nk) {
    constexpr int WGQ = TQ >= 256 ? 2 : 1, WGP = 4 / WGQ;       // wave grid: channels x pixels
    constexpr int WQ = TQ / WGQ, WP = 64, TP = WP * WGP;         // wave tile, pixel extent
    constexpr int NI = WQ / 16, NJ = WP / 16;
    constexpr int WROWS = TQ < 64 ? 64 : TQ;                     // weight rows staged (zero rows beyond Cout)
    constexpr int NXI = TP / 64, NWI = WROWS / 64;               // DMA pieces per wave per step
    constexpr int NPIECE = NXI + NWI;
    constexpr int XBYTES = TP * 64, STAGE = (TP + WROWS) * 64, NSTAGE = 3;
    constexpr int HQ = WQ < 64 ? WQ : 64, NH = WQ / HQ;          // epilogue: channel halves of <= 64
    constexpr int SLOTS = HQ / 4, PPI = 64 / SLOTS;              // 16-byte slots per row, pixels per instruction
    constexpr unsigned OOB = 0x80000000u;
    // (the launcher sizes the dynamic LDS as max(pipeline stages, 4 per-wave transpose images))

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wq = wave % WGQ, wp = wave / WGQ;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tq = id % ntile_q, tp = id / ntile_q;
    const int p0 = tp * TP, q0 = tq * TQ;
    const i32x4_t xrsrc = make_rsrc(a.x, a.x_bytes);
    const i32x4_t wrsrc = make_rsrc(a.w, a.w_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

    // ---- DMA geometry: a piece is 16 rows; lane -> row lane>>2, slot lane&3 ----------
    const int srow = lane >> 2;
    const int chunk = (lane & 3) ^ ((-(srow >> 2)) & 3);        // K-chunk (8 elements) fetched
    int xoff[NXI];
    unsigned xmask[NXI];
    {
        const int ohw = a.OH * a.OW;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int p = p0 + (wave * NXI + i) * 16 + srow;
            int off = 0;
            unsigned m = 0;
            if (p < a.P) {
                const int b = p / ohw, rem = p - b * ohw;
                const int oh = rem / a.OW, ow = rem - oh * a.OW;
                const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;
                off = ((b * a.H + ih0) * a.W + iw0) * a.x_ld;
                for (int r = 0; r < a.k; ++r)
                    for (int c = 0; c < a.k; ++c) {
                        const bool ok = (unsigned)(ih0 + r) < (unsigned)a.H &&
                                        (unsigned)(iw0 + c) < (unsigned)a.W;
                        m |= (ok ? 1u : 0u) << (r * a.k + c);
                    }
            }
            xoff[i] = off;
            xmask[i] = m;
        }
    }
    int ci = chunk * 8, tap = 0, tr = 0, tc = 0;
    while (ci >= a.Cin) {
        ci -= a.Cin;
        ++tap;
        if (++tc == a.k) { tc = 0; ++tr; }
    }
    const unsigned woff0 = (unsigned)(((q0 + wave * NWI * 16 + srow) * a.Kw + chunk * 8) * 2);
    const unsigned wpiece = (unsigned)(16 * a.Kw * 2);

    // issue this wave's DMA pieces of tile kt into `stage`, then advance its K position.
    // Tiles past the end use out-of-range offsets (zero fill, no traffic), so every
    // step issues the same number of pieces and the counted wait below is exact.
    auto dma = [&](int kt, int stage) {
        const unsigned sbase = lds0 + (unsigned)(stage * STAGE);
        const int tapoff = (tr * a.W + tc) * a.x_ld + ci;
        const unsigned tapbit = tap < 32 ? (1u << tap) : 0u;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const bool ok = (xmask[i] & tapbit) != 0;
            const unsigned off = ok ? (unsigned)(xoff[i] + tapoff) * 2u : OOB;
            lds_dma16(xrsrc, sbase + (unsigned)((wave * NXI + i) * 1024), off);
        }
        const unsigned wk = kt < nk ? woff0 + (unsigned)kt * 64u : OOB;
#pragma unroll
        for (int i = 0; i < NWI; ++i)
            lds_dma16(wrsrc, sbase + (unsigned)(XBYTES + (wave * NWI + i) * 1024), wk + (unsigned)i * wpiece);
        ci += 32;
        while (ci >= a.Cin) {
            ci -= a.Cin;
            ++tap;
            if (++tc == a.k) { tc = 0; ++tr; }
        }
    };

    f32x4_t acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment reads: row = tile row + (lane & 15), chunk = lane >> 4
    const int frow = lane & 15;
    const int fswz = (((lane >> 4) ^ ((-(frow >> 2)) & 3)) << 4);
    const int a_off = XBYTES + (wq * WQ + frow) * 64 + fswz;     // + i * 1024
    const int b_off = (wp * WP + frow) * 64 + fswz;              // + j * 1024

    auto compute = [&](int stage) {
        const char *base = smem + stage * STAGE;
        uint4 bfr[NJ], af[2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = *(const uint4 *)(base + b_off + j * 1024);
        af[0] = *(const uint4 *)(base + a_off);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 1 < NI) af[(i + 1) & 1] = *(const uint4 *)(base + a_off + (i + 1) * 1024);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, af[i & 1]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
        }
    };

    dma(0, 0);
    dma(1, 1);
    vmcnt_wait<NPIECE>::run();
    __syncthreads();
    int st = 0;                                          // stage of tile s
    for (int s = 0; s < nk; ++s) {
        int st2 = st + 2;
        if (st2 >= NSTAGE) st2 -= NSTAGE;
        dma(s + 2, st2);
        compute(st);
        vmcnt_wait<NPIECE>::run();
        __syncthreads();
        st = st + 1 == NSTAGE ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: per-wave transpose through LDS ------------------------------------
    OT *__restrict__ y = (OT *)a.y;
    const bf16_t *__restrict__ res = (const bf16_t *)a.res;
    const bool vec_ok = ((a.Cout & 3) == 0) && ((a.y_ld & 3) == 0) && (res == nullptr || (a.res_ld & 3) == 0);
    char *tbase = smem + wave * (WP * HQ * 4);
    const int act = a.act;
#pragma unroll
    for (int half = 0; half < NH; ++half) {
#pragma unroll
        for (int il = 0; il < HQ / 16; ++il) {
            const int i = half * (HQ / 16) + il;
            const int co = q0 + wq * WQ + i * 16 + 4 * (lane >> 4);
            float sc[4], sh[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = co + r;
                const bool in = c < a.Cout;
                sc[r] = in ? (a.scale ? a.scale[c] : 1.0f) : 0.0f;
                sh[r] = (in && a.shift) ? a.shift[c] : 0.0f;
            }
            const int slot = il * 4 + (lane >> 4);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float4 v;
                v.x = act_f(acc[i][j][0] * sc[0] + sh[0], act);
                v.y = act_f(acc[i][j][1] * sc[1] + sh[1], act);
                v.z = act_f(acc[i][j][2] * sc[2] + sh[2], act);
                v.w = act_f(acc[i][j][3] * sc[3] + sh[3], act);
                const int px = j * 16 + (lane & 15);
                *(float4 *)(tbase + px * (HQ * 4) + ((slot ^ (px & (SLOTS - 1))) << 4)) = v;
            }
        }
        // rows back out: lane -> pixel PPI*n + lane / SLOTS, channels 4*(lane % SLOTS) .. +3
        const int slot = lane % SLOTS;
        const int co = q0 + wq * WQ + half * HQ + slot * 4;
#pragma unroll 4
        for (int n = 0; n < WP / PPI; ++n) {
            const int px = n * PPI + lane / SLOTS;
            const int p = p0 + wp * WP + px;
            float4 v = *(const float4 *)(tbase + px * (HQ * 4) + ((slot ^ (px & (SLOTS - 1))) << 4));
            if (p >= a.P || co >= a.Cout) continue;
            if (vec_ok) {
                if (res) {
                    const uint2 rv = *(const uint2 *)(res + (size_t)p * a.res_ld + co);
                    v.x += __uint_as_float(rv.x << 16);
                    v.y += __uint_as_float(rv.x & 0xffff0000u);
                    v.z += __uint_as_float(rv.y << 16);
                    v.w += __uint_as_float(rv.y & 0xffff0000u);
                }
                OT *yp = y + (size_t)p * a.y_ld + co;
                if constexpr (sizeof(OT) == 2) {
                    *(uint2 *)yp = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
                } else {
                    *(float4 *)yp = v;
                }
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (co + r < a.Cout) {
                        float o = vv[r];
                        if (res) o += bf16_to_f32(res[(size_t)p * a.res_ld + co + r]);
                        elem<OT>::store(y + (size_t)p * a.y_ld + co + r, o);
                    }
                }
            }
        }
    }
}

template <typename OT, int TQ>
static int launch_dma2_cfg(const ConvArgs &a, hipStream_t s) {
    constexpr int TP = TQ >= 256 ? 128 : 256;
    constexpr int WROWS = TQ < 64 ? 64 : TQ;
    constexpr int HQ = (TQ >= 256 ? 128 : TQ) < 64 ? TQ : 64;
    constexpr int PIPE = 3 * (TP + WROWS) * 64, EPI = 4 * 64 * HQ * 4;
    constexpr int LDS = PIPE > EPI ? PIPE : EPI;
    auto kern = conv_igemm_dma2_kernel<OT, TQ>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            (void)hipGetLastError();
            return fail(YMX_ELAUNCH, "conv_igemm_dma2: cannot reserve %d bytes of LDS", LDS);
        }
        configured = true;
    }
    const int ntile_p = (a.P + TP - 1) / TP;
    const int ntile_q = (a.Cout + TQ - 1) / TQ;
    const int nk = (a.K + 31) / 32;
    const long long grid = (long long)ntile_p * ntile_q;
    if (grid > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_igemm_dma2: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, s, a, ntile_q, nk);
    return check_launch("conv_igemm_dma2");
}

template <typename OT>
static int launch_dma2_shape(const ConvArgs &a, hipStream_t s) {
    if (a.Cout <= 32) return launch_dma2_cfg<OT, 32>(a, s);
    if (a.Cout <= 64) return launch_dma2_cfg<OT, 64>(a, s);
    if (a.Cout <= 128) return launch_dma2_cfg<OT, 128>(a, s);
    return launch_dma2_cfg<OT, 256>(a, s);
}

// bf16 operands only; out_dtype bf16 or f32.
int launch_conv_igemm_dma2(const ConvArgs &a, hipStream_t s) {
    if (a.dtype != YMX_BF16) return fail(YMX_EUNSUPPORTED, "conv_igemm_dma2: bf16 operands only");
    if (a.out_dtype == YMX_BF16) return launch_dma2_shape<bf16_t>(a, s);
    return launch_dma2_shape<float>(a, s);
}

}  // namespace ymx
