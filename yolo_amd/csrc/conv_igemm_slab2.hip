// bf16 3x3 stride-1 convolution of the narrow first-stage layers (Cin = 32 or 64), input
// pixels resident in LDS across the nine filter taps, several workgroups per CU (gfx950).
//
// conv_igemm_dma2.hip copies the (256 pixels x 32 K) input tile of every K step from L2 into
// LDS: nine times the same pixels for a 3x3 filter, and with K <= 576 those copies, not
// the MFMAs, are what such a layer spends its time on.  Here, as in conv_igemm_halo.hip,
// a tile of consecutive output pixels stages its input once: the contiguous pixel run
// [p0 - W - 1, p0 + TP - 1 + W + 1] with all Cin channels (one channel block), and tap
// (r, c) of output pixel p0 + n reads slab row n + r*W + c; taps outside the image are
// redirected to a zero row.  Only the weights stream per K step (32 deep), through a ring
// of four or five small stages with counted waits.
//
//   Cin = 32   tile 64 ch x 256 px, 1 x 4 waves, slab rows of 64 B,  9 K steps
//   Cin = 64   tile 128 ch x 128 px, 2 x 2 waves, slab rows of 128 B, 18 K steps
//   every wave owns 64 ch x 64 px; LDS = slab + weight ring + 1 KB zeros <= 80 KB, so
//   two workgroups share a CU and one's epilogue overlaps the other's loop
//
// Same K order and epilogue as every other MFMA kernel here: bit-identical results.
#include "conv_epilogue.h"

namespace ymx {

template <typename OT, int CIN>
__global__ void __launch_bounds__(256, 2)
conv_igemm_slab2_kernel(const ConvArgs a, const int npieces) {
    constexpr int TQ = CIN == 32 ? 64 : 128, TP = CIN == 32 ? 256 : 128;
    constexpr int WGQ = TQ / 64;                                 // wave grid: channels x pixels
    constexpr int NI = 4, NJ = 4;                                // 64 ch x 64 px per wave
    constexpr int ROWB = CIN * 2;                                // slab row bytes
    constexpr int RPP = 1024 / ROWB;                             // slab rows per DMA piece
    constexpr int KSUB = CIN / 32;                               // K steps per tap
    constexpr int NSTEP = 9 * KSUB;
    constexpr int WT = TQ * 64;                                  // weight stage: TQ rows x 32 K
    constexpr int NWI = TQ / 64;                                 // weight pieces (16 rows x 64 B) per wave per step
    constexpr int NSTAGE = CIN == 32 ? 5 : 4;                    // weight ring: a stage has NSTAGE-3 steps to land
    constexpr int INFLIGHT = NWI * (NSTAGE - 3);                 // pieces that may still fly when a step ends
    constexpr unsigned OOB = 0x80000000u;
    typedef EpiGeom<NI> Epi;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    YMX_STAMP(0);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wq = wave % WGQ, wp = wave / WGQ;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int p0 = id * TP;
    const i32x4_t xrsrc = make_rsrc(a.x, a.x_bytes);
    const i32x4_t wrsrc = make_rsrc(a.w, a.w_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const int slab_bytes = npieces * 1024;
    const int zero_off = slab_bytes + NSTAGE * WT;

    // ---- shortcut input, fetched first (oldest in the queue: every wait below covers it), in the store pattern of the epilogue ----------------
    const int cbase = wq * 64, pbase = p0 + wp * 64;
    epi_u32x4_t rpre[Epi::NH][Epi::NPASS];
    if (epi_res_pre(a)) {
        epi_load_res<NI>(a, 0, cbase, pbase, lane, rpre[0]);
    } else {
#pragma unroll
        for (int n = 0; n < Epi::NPASS; ++n) rpre[0][n] = epi_u32x4_t{0u, 0u, 0u, 0u};
    }

    // ---- slab DMA: piece i = slab rows RPP*i .. +RPP-1 = input pixels p0 - W - 1 + row ----
    // the lane filling 16-byte slot s of a row fetches K-chunk s ^ swizzle(row)
    const int srow = CIN == 32 ? lane >> 2 : lane >> 3;
    const int schunk = CIN == 32 ? (lane & 3) ^ ((-(srow >> 2)) & 3) : (lane & 7) ^ srow;
    const unsigned slab_off0 = ((unsigned)(p0 - a.W - 1 + srow) * (unsigned)a.x_ld + (unsigned)(schunk * 8)) * 2u;
    const unsigned slab_step = (unsigned)(RPP * 2) * (unsigned)a.x_ld;
    for (int i = wave; i < npieces; i += 4)
        lds_dma16(xrsrc, lds0 + (unsigned)(i * 1024), slab_off0 + (unsigned)i * slab_step);
    if (wave == 0) lds_dma16(xrsrc, lds0 + (unsigned)zero_off, OOB);         // the zero row(s)

    // ---- weight ring: stage = TQ rows x 64 B, 16 rows per piece, rows swizzled as 64-B rows ----
    const int wrow = lane >> 2;
    const int wchunk = (lane & 3) ^ ((-(wrow >> 2)) & 3);
    const unsigned woff0 = (unsigned)(((wave * NWI * 16 + wrow) * a.Kw + wchunk * 8) * 2);
    const unsigned wpiece = (unsigned)(16 * a.Kw * 2);
    auto dma_w = [&](int s, int stage) {
        const unsigned sbase = lds0 + (unsigned)(slab_bytes + stage * WT + wave * NWI * 1024);
        const unsigned wk = s < NSTEP ? woff0 + (unsigned)s * 64u : OOB;
#pragma unroll
        for (int i = 0; i < NWI; ++i) lds_dma16(wrsrc, sbase + (unsigned)(i * 1024), wk + (unsigned)i * wpiece);
    };
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i) dma_w(i, i);

    // ---- fragment geometry ---------------------------------------------------------------
    unsigned fmask[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int unused;
        pixel_geometry(a, pbase + j * 16 + (lane & 15), unused, fmask[j]);
    }
    const int hb0 = wp * 64 + (lane & 15);                       // slab row of block 0, tap (0, 0)
    const int q4 = lane >> 4;
    const int frow = lane & 15;
    const int a_off = slab_bytes + (wq * 64 + frow) * 64 + ((q4 ^ ((-(frow >> 2)) & 3)) << 4);   // + stage * WT + i * 1024

    f32x4_t acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    YMX_STAMP(1);
    // shortcut, slab, zero row and the first two weight stages have landed
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
    __syncthreads();
    YMX_STAMP(2);

    // The fragments of step s+1 are read from LDS while the MFMAs of step s run (two register
    // sets), so a weight stage must be visible one step before it is used: the wait at the end
    // of step s covers stage s+2.  All step indices are compile-time (the loop is unrolled).
    uint4 af[2][NI], bfr[2][NJ];
    auto load_frags = [&](int s, int set) {
        const int tap = s / KSUB, sub = s - tap * KSUB;
        const int tr = tap / 3;
        const int hb = hb0 + tr * a.W + (tap - tr * 3);
        // this lane's row and K-chunk in the slab; rows 16 j further share the swizzle
        const int base = CIN == 32 ? hb * 64 + ((q4 ^ ((-(hb >> 2)) & 3)) << 4) : hb * 128 + ((q4 ^ (hb & 7)) << 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int off = ((fmask[j] >> tap) & 1u) ? base + j * 16 * ROWB : zero_off;
            bfr[set][j] = *(const uint4 *)(smem + (sub ? off ^ 64 : off));
        }
        const char *wbase = smem + a_off + (s % NSTAGE) * WT;
#pragma unroll
        for (int i = 0; i < NI; ++i) af[set][i] = *(const uint4 *)(wbase + i * 1024);
    };
    load_frags(0, 0);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        dma_w(s + NSTAGE - 1, (s + NSTAGE - 1) % NSTAGE);
        if (s + 1 < NSTEP) load_frags(s + 1, (s + 1) & 1);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, af[s & 1][i]), __builtin_bit_cast(bf16x8_t, bfr[s & 1][j]), acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    YMX_STAMP(3);

    epi_run<OT, NI, NJ>(a, acc, smem + wave * Epi::LDS_BYTES, cbase, pbase, lane, rpre);
    YMX_STAMP_END(5);
}

// Shapes this kernel takes: 3x3 stride 1, 32 input channels into <= 64 output channels or
// 64 into <= 128, rows short enough for the slab of two workgroups to share a CU's LDS.
// LDS bytes of one workgroup: the input slab (tile pixels plus a row and a pixel on either
// side, in 1 KB DMA pieces), the weight ring, one zero row; never less than the epilogue's
// transpose images.  Two workgroups share a CU, so 80 KB is the limit.
static constexpr int SLAB2_LDS_MAX = 80 * 1024;
static int slab2_lds_bytes(int Cin, int W, int *npieces_out = nullptr) {
    const int TQ = Cin == 32 ? 64 : 128, TP = Cin == 32 ? 256 : 128, NSTAGE = Cin == 32 ? 5 : 4;
    const long long rows = TP + 2ll * W + 2;
    const long long npieces = (rows * Cin * 2 + 1023) / 1024;
    if (npieces_out) *npieces_out = (int)(npieces > 0x7fffffff ? 0x7fffffff : npieces);
    long long lds = npieces * 1024 + NSTAGE * TQ * 64 + 1024;
    if (lds < 4 * EpiGeom<4>::LDS_BYTES) lds = 4 * EpiGeom<4>::LDS_BYTES;
    return lds > 0x7fffffff ? 0x7fffffff : (int)lds;
}

bool conv_slab2_eligible(const ConvArgs &a) {
    if (a.dtype != YMX_BF16 || a.k != 3 || a.stride != 1) return false;
    if (!((a.Cin == 32 && a.Cout <= 64) || (a.Cin == 64 && a.Cout <= 128))) return false;
    return slab2_lds_bytes(a.Cin, a.W) <= SLAB2_LDS_MAX;      // rows up to 343 / 111 pixels
}

template <typename OT, int CIN>
static int launch_slab2_cfg(const ConvArgs &a, hipStream_t s) {
    constexpr int TP = CIN == 32 ? 256 : 128;
    int npieces = 0;
    const int lds = slab2_lds_bytes(CIN, a.W, &npieces);
    if (lds > SLAB2_LDS_MAX) return fail(YMX_EUNSUPPORTED, "conv_igemm_slab2: row of %d pixels too long", a.W);
    auto kern = conv_igemm_slab2_kernel<OT, CIN>;
    static LdsReserve reserved;        // one per kernel instantiation; remembers the devices already set up
    if (const int rc = reserve_lds(reserved, (const void *)kern, 80 * 1024, "conv_igemm_slab2")) return rc;
    const long long grid = ((long long)a.P + TP - 1) / TP;
    if (grid > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_igemm_slab2: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a, npieces);
    return check_launch("conv_igemm_slab2");
}

int launch_conv_igemm_slab2(const ConvArgs &a, hipStream_t s) {
    if (!conv_slab2_eligible(a)) return fail(YMX_EUNSUPPORTED, "conv_igemm_slab2: shape not supported");
    if (a.Cin == 32) {
        if (a.z) return launch_slab2_cfg<bf16_zout_t, 32>(a, s);
        if (a.out_dtype == YMX_BF16) return launch_slab2_cfg<bf16_t, 32>(a, s);
        return launch_slab2_cfg<float, 32>(a, s);
    }
    if (a.z) return launch_slab2_cfg<bf16_zout_t, 64>(a, s);
    if (a.out_dtype == YMX_BF16) return launch_slab2_cfg<bf16_t, 64>(a, s);
    return launch_slab2_cfg<float, 64>(a, s);
}

}  // namespace ymx

YMX_STAMP_EXPORT(ymx_debug_stamps_dma6)
