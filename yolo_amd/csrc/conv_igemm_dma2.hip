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
//   LDS            NSTAGE (2, 3 or 4) stages x (TP + max(TQ, 64)) rows x 64 B  (72 .. 80 KB;
//                  40 KB = four workgroups per CU for the two-stage 32-channel tile that
//                  takes layers of one or two K steps)
//   pipeline       step s: issue DMA of tile s+NSTAGE-1, MFMAs of tile s, wait for tile
//                  s+1 (counted vmcnt: the younger pieces may stay in flight), one
//                  barrier.  A tile therefore has NSTAGE-1 full steps to land.
//   shortcut       the residual input is fetched into registers before the K loop
//   epilogue       per-wave transpose through LDS (fp32, XOR-swizzled), bias/BN,
//                  activation, shortcut, one rounding, 16-byte stores, 128 (or 64)
//                  contiguous bytes per pixel per store instruction
//
// LDS image of one DMA piece (one wave-instruction, 1 KB) = 16 rows x 64 B.  The 16-byte
// slot of K-chunk c in row r is  c ^ ((-(r >> 2)) & 3): with that XOR the ds_read_b128
// fragment reads (16 rows x one chunk per 16-lane group) touch 16 distinct slots of the
// 256-byte bank row.  The swizzle is applied to the DMA's source address (the lane filling
// slot s of row r fetches chunk s ^ f(r)), the LDS image itself stays linear.
#include "conv_epilogue.h"

namespace ymx {

template <int N> __device__ __forceinline__ void vmcnt_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename OT, int TQ, int NSTAGE, int EHQ = 64>
__global__ void __launch_bounds__(256, 2)
conv_igemm_dma2_kernel(const ConvArgs a, const int ntile_q, const int nk) {
    constexpr int WGQ = TQ >= 256 ? 2 : 1, WGP = 4 / WGQ;       // wave grid: channels x pixels
    constexpr int WQ = TQ / WGQ, WP = 64, TP = WP * WGP;         // wave tile, pixel extent
    constexpr int NI = WQ / 16, NJ = WP / 16;
    constexpr int WROWS = TQ < 64 ? 64 : TQ;                     // weight rows staged (zero rows beyond Cout)
    constexpr int NXI = TP / 64, NWI = WROWS / 64;               // DMA pieces per wave per step
    constexpr int NPIECE = NXI + NWI;
    constexpr int XBYTES = TP * 64, STAGE = (TP + WROWS) * 64;
    constexpr int INFLIGHT = NPIECE * (NSTAGE - 2);             // pieces that may still be in flight at a step's end
    typedef EpiGeom<NI, EHQ> Epi;
    constexpr unsigned OOB = 0x80000000u;
    // (the launcher sizes the dynamic LDS as max(pipeline stages, 4 per-wave transpose images))

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

    // issue this wave's DMA pieces of tile kt into `stage`.
    // Tiles past the end use out-of-range offsets (zero fill, no traffic), so every
    // step issues the same number of pieces and the counted wait below is exact.
    auto dma = [&](int kt, int stage) {
        const unsigned sbase = lds0 + (unsigned)(stage * STAGE);
        int tapoff;
        unsigned tapbit;
        k_decode(a, kt * 32 + chunk * 8, tapoff, tapbit);
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
    };

    // ---- shortcut input: 64-channel wave tiles fetch it now, so that its latency hides
    // behind the whole K loop (it is older than every DMA piece: the counted waits below
    // cover it); 128-channel ones, short of registers, fetch the first half before the
    // last K step and the second half during the epilogue
    const int cbase = q0 + wq * WQ, pbase = p0 + wp * WP;
    const bool res_pre = epi_res_pre(a);
    epi_u32x4_t rpre[Epi::NH][Epi::NPASS];
    if (res_pre && Epi::NH == 1) {
        epi_load_res<NI, EHQ>(a, 0, cbase, pbase, lane, rpre[0]);
    } else {
#pragma unroll
        for (int n = 0; n < Epi::NPASS; ++n) rpre[0][n] = epi_u32x4_t{0u, 0u, 0u, 0u};
    }

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

    YMX_STAMP(1);
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i) dma(i, i);
    if constexpr (NI <= 4 && NSTAGE >= 4) {
        // 32- and 64-channel tiles have registers to spare: the fragments of step s+1 are read
        // from LDS while the MFMAs of step s run (two register sets).  A stage must then be
        // visible one step earlier, so one stage less may be in flight when a step ends.
        constexpr int INFLIGHT_P = NPIECE * (NSTAGE - 3);
        vmcnt_wait<INFLIGHT_P>();                       // stages 0 and 1 have landed
        __syncthreads();
        YMX_STAMP(2);
        uint4 af[2][NI], bfr[2][NJ];
        auto load_frags = [&](int stage, auto set_tag) {
            constexpr int SET = decltype(set_tag)::value;
            const char *base = smem + stage * STAGE;
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[SET][j] = *(const uint4 *)(base + b_off + j * 1024);
#pragma unroll
            for (int i = 0; i < NI; ++i) af[SET][i] = *(const uint4 *)(base + a_off + i * 1024);
        };
        int st = 0;                                      // stage of tile s
        auto step = [&](int s, auto set_tag) {
            constexpr int SET = decltype(set_tag)::value;
            int stn = st + NSTAGE - 1;
            if (stn >= NSTAGE) stn -= NSTAGE;
            dma(s + NSTAGE - 1, stn);
            const int st1 = st + 1 == NSTAGE ? 0 : st + 1;
            load_frags(st1, std::integral_constant<int, SET ^ 1>{});     // tile s+1 (zeros past the end)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, af[SET][i]), __builtin_bit_cast(bf16x8_t, bfr[SET][j]), acc[i][j], 0, 0, 0);
            vmcnt_wait<INFLIGHT_P>();                   // tile s+2 has landed
            __syncthreads();
            st = st1;
        };
        load_frags(0, std::integral_constant<int, 0>{});
        int s = 0;
        for (; s + 1 < nk; s += 2) {
            step(s, std::integral_constant<int, 0>{});
            step(s + 1, std::integral_constant<int, 1>{});
        }
        if (s < nk) step(s, std::integral_constant<int, 0>{});
    } else {
        vmcnt_wait<INFLIGHT>();
        __syncthreads();
        YMX_STAMP(2);
        int st = 0;                                      // stage of tile s
        auto step = [&](int s) {
            int stn = st + NSTAGE - 1;
            if (stn >= NSTAGE) stn -= NSTAGE;
            dma(s + NSTAGE - 1, stn);
            compute(st);
            vmcnt_wait<INFLIGHT>();
            __syncthreads();
            st = st + 1 == NSTAGE ? 0 : st + 1;
        };
        for (int s = 0; s < nk - 1; ++s) step(s);
        if constexpr (Epi::NH == 2) {
            if (res_pre) epi_load_res<NI, EHQ>(a, 0, cbase, pbase, lane, rpre[0]);   // older than the last step's pieces
        }
        step(nk - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    YMX_STAMP(3);

    // ---- epilogue: per-wave transpose through LDS (conv_epilogue.h) --------------------
    epi_run<OT, NI, NJ, EHQ>(a, acc, smem + wave * Epi::LDS_BYTES, cbase, pbase, lane, rpre);
    YMX_STAMP_END(5);
}

template <typename OT, int TQ, int NSTAGE, int EHQ = 64>
static int launch_dma2_cfg(const ConvArgs &a, hipStream_t s) {
    constexpr int TP = TQ >= 256 ? 128 : 256;
    constexpr int WROWS = TQ < 64 ? 64 : TQ;
    constexpr int WQ = TQ >= 256 ? 128 : TQ;
    constexpr int PIPE = NSTAGE * (TP + WROWS) * 64, EPI = 4 * EpiGeom<WQ / 16, EHQ>::LDS_BYTES;
    constexpr int LDS = PIPE > EPI ? PIPE : EPI;
    auto kern = conv_igemm_dma2_kernel<OT, TQ, NSTAGE, EHQ>;
    static LdsReserve reserved;        // one per kernel instantiation; remembers the devices already set up
    if (const int rc = reserve_lds(reserved, (const void *)kern, LDS, "conv_igemm_dma2")) return rc;
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
    // A short K (up to four steps) needs no deep ring: two stages = 40 KB of LDS, and the
    // workgroups per CU that buys (four of the 32-channel tile, three of the 64-channel one,
    // whose transpose images are then 32 channels wide) hide the prologue and epilogue
    // latencies that dominate such layers.  Measured: 1x1 64->32 @208^2 -25 %, 1x1 128->64
    // @104^2 -13 %; no gain for the nine-step stride-2 3x3 32->64, which keeps four stages.
    if (g_dma2_stages != 3) {
        if (a.Cout <= 32 && a.K <= 64) return launch_dma2_cfg<OT, 32, 2>(a, s);
        if (a.Cout > 32 && a.Cout <= 64 && a.K <= 128) return launch_dma2_cfg<OT, 64, 2, 32>(a, s);
    }
    if (g_dma2_stages == 4) {
        if (a.Cout <= 32) return launch_dma2_cfg<OT, 32, 4>(a, s);
        if (a.Cout <= 64) return launch_dma2_cfg<OT, 64, 4>(a, s);
    }
    if (a.Cout <= 32) return launch_dma2_cfg<OT, 32, 3>(a, s);
    if (a.Cout <= 64) return launch_dma2_cfg<OT, 64, 3>(a, s);
    if (a.Cout <= 128) return launch_dma2_cfg<OT, 128, 3>(a, s);
    return launch_dma2_cfg<OT, 256, 3>(a, s);
}

// bf16 operands only; out_dtype bf16 or f32.
int launch_conv_igemm_dma2(const ConvArgs &a, hipStream_t s) {
    if (a.dtype != YMX_BF16) return fail(YMX_EUNSUPPORTED, "conv_igemm_dma2: bf16 operands only");
    if (a.z) return launch_dma2_shape<bf16_zout_t>(a, s);
    if (a.d2s_c) return launch_dma2_shape<bf16_d2s_t>(a, s);
    if (a.out_dtype == YMX_BF16) return launch_dma2_shape<bf16_t>(a, s);
    return launch_dma2_shape<float>(a, s);
}

}  // namespace ymx

YMX_STAMP_EXPORT(ymx_debug_stamps_dma2)
