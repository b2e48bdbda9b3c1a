// Large-tile bf16 implicit-GEMM convolution with LDS-DMA staging (gfx950).
//
// Same mathematics and operand orientation as conv_igemm.hip (weights as the MFMA A
// operand, pixels as B, D[row = channel][col = pixel]); what differs is the data path:
//
//  - a workgroup of 8 waves (512 threads) owns a 256-pixel x TQ-channel tile (TQ = 256 or
//    128), so every byte staged from L2 feeds twice as many MFMAs as in the 128x128
//    kernel, whose staging traffic (64 B/clk/CU at full MFMA rate) sits on the CU's
//    L1 bandwidth;
//  - tiles go global -> LDS directly with `buffer_load_dwordx4 ... lds` (no VGPR
//    staging, no ds_write pass).  The LDS image of one wave-instruction is linear
//    (M0 base + lane*16 B = 8 rows x 128 B); the XOR swizzle that keeps the ds_read_b128
//    fragment reads conflict-free is applied on the SOURCE side: the lane that fills
//    slot s of row r fetches K-chunk s ^ (r & 7).  A lane whose tap is outside the image
//    (zero padding) or beyond K uses an out-of-range buffer offset, for which the
//    hardware writes zeros into LDS: padding costs no branch and no memory traffic;
//  - two LDS stages of (256 + TQ) x 128 B; the DMA of tile t+1 is issued before the
//    MFMAs of tile t and waited for (vmcnt(0)) after them, one barrier per K step.
#include "conv_epilogue.h"

namespace ymx {

template <typename OT, int TQ>
__global__ void __launch_bounds__(512, 2)
conv_igemm_dma_kernel(const ConvArgs a, const int ntile_q, const int nk) {
    constexpr int TP = 256;
    constexpr int WGQ = TQ / 128, WGP = 8 / WGQ;        // wave grid: channels x pixels
    constexpr int WQ = 128, WP = TP / WGP;              // wave tile
    constexpr int NI = WQ / 16, NJ = WP / 16;
    constexpr int NXI = TP / 64, NWI = TQ / 64;         // DMA instructions per wave per step
    constexpr int XBYTES = TP * 128, STAGE = (TP + TQ) * 128;
    constexpr unsigned OOB = 0x80000000u;

    typedef EpiGeom<NI> Epi;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    YMX_STAMP(0);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wq = wave % WGQ, wp = wave / WGQ;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tq = id % ntile_q, tp = id / ntile_q;
    const int p0 = tp * TP, q0 = tq * TQ;
    const i32x4_t xrsrc = make_rsrc(a.x, a.x_bytes);
    const i32x4_t wrsrc = make_rsrc(a.w, a.w_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

    // ---- DMA geometry: instruction i of this wave fills rows 8*(wave*N + i) .. +7 ----
    const int srow = lane >> 3;                         // row inside an 8-row piece
    const int chunk = (lane & 7) ^ (srow & 7);          // K-chunk this lane fetches (swizzle)
    int xoff[NXI];
    unsigned xmask[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) pixel_geometry(a, p0 + (wave * NXI + i) * 8 + srow, xoff[i], xmask[i]);
    // byte offset of this lane's weight chunk in tile 0, piece 0 of this wave
    const unsigned woff0 = (unsigned)(((q0 + wave * NWI * 8 + srow) * a.Kw + chunk * 8) * 2);
    const unsigned wpiece = (unsigned)(8 * a.Kw * 2);

    auto dma = [&](int kt, int stage) {
        const unsigned sbase = lds0 + (unsigned)(stage * STAGE);
        int tapoff;
        unsigned tapbit;
        k_decode(a, kt * 64 + chunk * 8, tapoff, tapbit);
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const bool ok = (xmask[i] & tapbit) != 0;
            const unsigned off = ok ? (unsigned)(xoff[i] + tapoff) * 2u : OOB;
            lds_dma16(xrsrc, sbase + (unsigned)((wave * NXI + i) * 1024), off);
        }
#pragma unroll
        for (int i = 0; i < NWI; ++i)
            lds_dma16(wrsrc, sbase + (unsigned)(XBYTES + (wave * NWI + i) * 1024),
                      woff0 + (unsigned)i * wpiece + (unsigned)kt * 128u);
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

    // One K step.  The wave's MFMA work is cut into NI segments (one 16-channel row of
    // its tile each: 2*NJ MFMAs); the DMA pieces of the next tile are issued one X and
    // one W piece per segment in the first segments (so the last piece still has half a
    // step to land), and the A fragments of segment i+1 are read from LDS before the
    // MFMAs of segment i.  With two waves per SIMD the issue work of one wave then runs
    // under the MFMAs of the other instead of both queueing for the matrix pipe at once.
    auto step = [&](int kt) {
        const char *base = smem + (kt & 1) * STAGE;
        const bool more = kt + 1 < nk;
        const unsigned sbase = lds0 + (unsigned)(((kt + 1) & 1) * STAGE);
        int tapoff;
        unsigned tapbit;
        k_decode(a, (kt + 1) * 64 + chunk * 8, tapoff, tapbit);
        const unsigned wstep = woff0 + (unsigned)(kt + 1) * 128u;

        uint4 bfr[2][NJ], af[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[kk][j] = *(const uint4 *)(base + b_off + j * 2048 + (kk ? swz1 : swz0));
        af[0][0] = *(const uint4 *)(base + a_off + swz0);
        af[0][1] = *(const uint4 *)(base + a_off + swz1);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 1 < NI) {
                af[(i + 1) & 1][0] = *(const uint4 *)(base + a_off + (i + 1) * 2048 + swz0);
                af[(i + 1) & 1][1] = *(const uint4 *)(base + a_off + (i + 1) * 2048 + swz1);
            }
            if (more) {
                if (i < NXI) {
                    const bool ok = (xmask[i] & tapbit) != 0;
                    const unsigned off = ok ? (unsigned)(xoff[i] + tapoff) * 2u : OOB;
                    lds_dma16(xrsrc, sbase + (unsigned)((wave * NXI + i) * 1024), off);
                }
                if (i < NWI)
                    lds_dma16(wrsrc, sbase + (unsigned)(XBYTES + (wave * NWI + i) * 1024), wstep + (unsigned)i * wpiece);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, af[i & 1][kk]), __builtin_bit_cast(bf16x8_t, bfr[kk][j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    YMX_STAMP(1);
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    YMX_STAMP(2);
    for (int kt = 0; kt < nk - 1; ++kt) step(kt);
    // first half of the shortcut input: in flight during the last K step (which issues no DMA)
    const int cbase = q0 + wq * WQ, pbase = p0 + wp * WP;
    epi_u32x4_t rpre[Epi::NH][Epi::NPASS];
    if (epi_res_pre(a)) {
        epi_load_res<NI>(a, 0, cbase, pbase, lane, rpre[0]);
    } else {
#pragma unroll
        for (int n = 0; n < Epi::NPASS; ++n) rpre[0][n] = epi_u32x4_t{0u, 0u, 0u, 0u};
    }
    step(nk - 1);
    YMX_STAMP(3);

    // ---- epilogue: per-wave transpose through LDS (conv_epilogue.h); LDS is free now ----
    static_assert(8 * Epi::LDS_BYTES <= 2 * STAGE, "per-wave transpose images must fit in LDS");
    epi_run<OT, NI, NJ>(a, acc, smem + wave * Epi::LDS_BYTES, cbase, pbase, lane, rpre);
    YMX_STAMP_END(5);
}

template <typename OT, int TQ>
static int launch_dma_cfg(const ConvArgs &a, hipStream_t s) {
    constexpr int LDS = 2 * (256 + TQ) * 128;
    auto kern = conv_igemm_dma_kernel<OT, TQ>;
    static LdsReserve reserved;        // one per kernel instantiation; remembers the devices already set up
    if (const int rc = reserve_lds(reserved, (const void *)kern, LDS, "conv_igemm_dma")) return rc;
    const int ntile_p = (a.P + 255) / 256;
    const int ntile_q = (a.Cout + TQ - 1) / TQ;
    const int nk = (a.K + 63) / 64;
    const long long grid = (long long)ntile_p * ntile_q;
    if (grid > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_igemm_dma: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, a, ntile_q, nk);
    return check_launch("conv_igemm_dma");
}

// bf16 operands only; out_dtype bf16 or f32.
int launch_conv_igemm_dma(const ConvArgs &a, hipStream_t s) {
    if (a.dtype != YMX_BF16) return fail(YMX_EUNSUPPORTED, "conv_igemm_dma: bf16 operands only");
    if (a.z) return launch_dma_cfg<bf16_zout_t, 256>(a, s);
    if (a.d2s_c) return launch_dma_cfg<bf16_d2s_t, 256>(a, s);
    if (a.out_dtype == YMX_BF16) return launch_dma_cfg<bf16_t, 256>(a, s);
    return launch_dma_cfg<float, 256>(a, s);
}

}  // namespace ymx

YMX_STAMP_EXPORT(ymx_debug_stamps_dma1)
