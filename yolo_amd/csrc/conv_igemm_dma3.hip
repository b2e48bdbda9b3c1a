// bf16 implicit-GEMM convolution, 256 x 256 tile, LDS-DMA ring with fragment prefetch
// across the barrier (gfx950).
//
// Same mathematics, operand orientation and zero padding as conv_igemm_dma.hip (weights =
// MFMA A operand, pixels = B; out-of-range buffer offsets write zeros into LDS), same
// 8-wave workgroup (2 x 4 waves of 128 channels x 64 pixels).  What differs is the
// pipeline around the per-step barrier:
//
//   K step         32 (LDS rows of 64 B), four ring stages of (256 + 256) rows = 128 KB
//   DMA            tile k+3 is issued during step k and has two full steps to land; the
//                  wait at the end of step k is a counted one (tile k+2 must be in, tile
//                  k+3 may still fly)
//   fragments      the pixel fragments of tile k+1 and its first two weight fragments are
//                  read from LDS DURING step k (tile k+1 became visible at the barrier
//                  that ended step k-1), so the first MFMAs of step k+1 issue right after
//                  the barrier instead of waiting for an LDS round trip; the other weight
//                  fragments stream two segments ahead of their MFMAs, as before
//
// The order in which K terms are added to an accumulator is the same as in every other
// convolution kernel here, so results are bit-identical to theirs.
#include "conv_epilogue.h"

namespace ymx {

template <typename OT>
__global__ void __launch_bounds__(512, 2)
conv_igemm_dma3_kernel(const ConvArgs a, const int ntile_q, const int nk) {
    constexpr int TP = 256, TQ = 256;
    constexpr int WGQ = 2;                                  // wave grid: 2 (channels) x 4 (pixels)
    constexpr int WQ = 128, WP = 64;
    constexpr int NI = WQ / 16, NJ = WP / 16;
    constexpr int NXI = 2, NWI = 2;                         // DMA pieces (16 rows x 64 B) per wave per step
    constexpr int XBYTES = TP * 64, STAGE = (TP + TQ) * 64, NSTAGE = 4;
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

    // ---- DMA geometry: a piece is 16 rows; lane -> row lane>>2, slot lane&3 ----------
    const int srow = lane >> 2;
    const int chunk = (lane & 3) ^ ((-(srow >> 2)) & 3);        // K-chunk (8 elements) fetched
    int xoff[NXI];
    unsigned xmask[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) pixel_geometry(a, p0 + (wave * NXI + i) * 16 + srow, xoff[i], xmask[i]);
    const unsigned woff0 = (unsigned)(((q0 + wave * NWI * 16 + srow) * a.Kw + chunk * 8) * 2);
    const unsigned wpiece = (unsigned)(16 * a.Kw * 2);

    // Offsets of the next tile to issue, branch-free: K index -> (tap, channel) -> (row, col).
    // Tiles past the end of K (and taps past the filter) get out-of-range offsets.
    int issue_kt = 0;
    unsigned xo[NXI], wo;
    auto dma_prepare = [&]() {
        int tapoff;
        unsigned tapbit;
        k_decode(a, issue_kt * 32 + chunk * 8, tapoff, tapbit);
#pragma unroll
        for (int i = 0; i < NXI; ++i) xo[i] = (xmask[i] & tapbit) ? (unsigned)(xoff[i] + tapoff) * 2u : OOB;
        wo = issue_kt < nk ? woff0 + (unsigned)issue_kt * 64u : OOB;
        ++issue_kt;
    };
    auto dma_x = [&](unsigned sbase, int i) { lds_dma16(xrsrc, sbase + (unsigned)((wave * NXI + i) * 1024), xo[i]); };
    auto dma_w = [&](unsigned sbase, int i) {
        lds_dma16(wrsrc, sbase + (unsigned)(XBYTES + (wave * NWI + i) * 1024), wo + (unsigned)i * wpiece);
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

    uint4 bfr[2][NJ];       // pixel fragments of the current and the next tile
    uint4 af[4];            // ring of weight fragments: segment i lives in af[i & 3]

    // ---- prologue: three tiles in flight, fragments of tile 0 in registers -------------
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
        dma_prepare();
        const unsigned sbase = lds0 + (unsigned)(s * STAGE);
#pragma unroll
        for (int i = 0; i < NXI; ++i) dma_x(sbase, i);
#pragma unroll
        for (int i = 0; i < NWI; ++i) dma_w(sbase, i);
    }
    YMX_STAMP(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // tile 0 is in
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) bfr[0][j] = *(const uint4 *)(smem + b_off + j * 1024);
    af[0] = *(const uint4 *)(smem + a_off);
    af[1] = *(const uint4 *)(smem + a_off + 1024);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // tile 1 is in
    __syncthreads();
    YMX_STAMP(2);

    // One K step: MFMAs of tile kt (stage st) from registers; LDS reads of the rest of its
    // weight fragments two segments ahead; in the later segments the first fragments of
    // tile kt+1 (stage st+1); DMA of tile kt+3 into the stage tile kt-1 has left.
    auto step = [&](int st, auto parity_tag) {
        constexpr int PAR = decltype(parity_tag)::value;    // which bfr set holds tile kt
        const char *cur = smem + st * STAGE;
        const int stn = st + 1 == NSTAGE ? 0 : st + 1;
        const char *nxt = smem + stn * STAGE;
        const int std_ = st == 0 ? NSTAGE - 1 : st - 1;
        const unsigned sbase = lds0 + (unsigned)(std_ * STAGE);
        dma_prepare();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 2 < NI) af[(i + 2) & 3] = *(const uint4 *)(cur + a_off + (i + 2) * 1024);
            else af[(i + 2) & 3] = *(const uint4 *)(nxt + a_off + (i + 2 - NI) * 1024);
            if (i >= NI - NJ) bfr[PAR ^ 1][i - (NI - NJ)] = *(const uint4 *)(nxt + b_off + (i - (NI - NJ)) * 1024);
            if (i < NXI) dma_x(sbase, i);
            else if (i < NXI + NWI) dma_w(sbase, i - NXI);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, af[i & 3]), __builtin_bit_cast(bf16x8_t, bfr[PAR][j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile kt+2 is in; tile kt+3 may still fly
        __syncthreads();
    };

    const int cbase = q0 + wq * WQ, pbase = p0 + wp * WP;
    epi_u32x4_t rpre[Epi::NH][Epi::NPASS];
    const bool res_pre = epi_res_pre(a);
#pragma unroll
    for (int n = 0; n < Epi::NPASS; ++n) rpre[0][n] = epi_u32x4_t{0u, 0u, 0u, 0u};

    // Steps come in pairs (the two fragment register sets alternate); an odd step count is
    // rounded up with one step on an all-zero tile.  The shortcut input's first half is
    // fetched before the last step: the pieces issued there are all out of range and
    // younger, so the counted wait still holds.
    int st = 0;
    for (int kt = 0; kt < nk; kt += 2) {
        step(st, std::integral_constant<int, 0>{});
        st = st + 1 == NSTAGE ? 0 : st + 1;
        if (kt + 2 >= nk && res_pre) epi_load_res<NI>(a, 0, cbase, pbase, lane, rpre[0]);
        step(st, std::integral_constant<int, 1>{});
        st = st + 1 == NSTAGE ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    YMX_STAMP(3);

    // ---- epilogue: per-wave transpose through LDS (conv_epilogue.h); LDS is free now ----
    static_assert(8 * Epi::LDS_BYTES <= NSTAGE * STAGE, "per-wave transpose images must fit in LDS");
    epi_run<OT, NI, NJ>(a, acc, smem + wave * Epi::LDS_BYTES, cbase, pbase, lane, rpre);
    YMX_STAMP_END(5);
}

template <typename OT>
static int launch_dma3_cfg(const ConvArgs &a, hipStream_t s) {
    constexpr int LDS = 4 * (256 + 256) * 64;
    auto kern = conv_igemm_dma3_kernel<OT>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            (void)hipGetLastError();
            return fail(YMX_ELAUNCH, "conv_igemm_dma3: cannot reserve %d bytes of LDS", LDS);
        }
        configured = true;
    }
    const int ntile_p = (a.P + 255) / 256;
    const int ntile_q = (a.Cout + 255) / 256;
    const int nk = (a.K + 31) / 32;
    const long long grid = (long long)ntile_p * ntile_q;
    if (grid > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_igemm_dma3: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, a, ntile_q, nk);
    return check_launch("conv_igemm_dma3");
}

// bf16 operands only; out_dtype bf16 or f32.
int launch_conv_igemm_dma3(const ConvArgs &a, hipStream_t s) {
    if (a.dtype != YMX_BF16) return fail(YMX_EUNSUPPORTED, "conv_igemm_dma3: bf16 operands only");
    if (a.out_dtype == YMX_BF16) return launch_dma3_cfg<bf16_t>(a, s);
    return launch_dma3_cfg<float>(a, s);
}

}  // namespace ymx

YMX_STAMP_EXPORT(ymx_debug_stamps_dma3)
