// bf16 implicit-GEMM convolution, 256 x 256 tile, phase-staggered wave groups (gfx950).
//
// Same mathematics, operand orientation and zero padding as conv_igemm_dma.hip (weights =
// MFMA A operand, pixels = B, D[row = channel][col = pixel]; out-of-range buffer offsets
// make the LDS-DMA write zeros), same 8-wave workgroup (2 x 4 waves of 128 channels x 64
// pixels), same K order per accumulator -- results are bit-identical.  The K loop is
// organised so that the matrix pipe of every SIMD always has a wave to serve:
//
//   groups    waves 0-3 and 4-7 (one of each per SIMD) run the same sequence one barrier
//             apart: while one group issues the 32 MFMAs of two quadrants, the other does
//             its LDS reads and DMA issue
//   intervals four per 64-deep K tile (load 1 | mfma (m0,n0) (m0,n1) | load 2 | mfma (m1,n1)
//             (m1,n0)), a barrier behind each
//   pieces    a K tile is staged as four 16 KB pieces W(m0) X(n0) X(n1) W(m1) -- the rows of
//             all waves' first/second channel halves and pixel halves.  Eight slots (two K
//             tiles) of LDS; counted vmcnt waits, see the loop
//   tiles     256 pixels (NJ = 4 blocks of 16 per wave) or 192 (NJ = 3) for layers whose
//             256-pixel tiles fill the last round of CUs badly: the second pixel half of a wave
//             is then one block, X(n1) carries 64 real rows (its other DMA instructions fetch
//             nothing and keep the counted waits exact) and the intervals hold 16 / 8 / 8 / 16
//             MFMAs -- the form conv_igemm_halo.hip has
#include "conv_epilogue.h"

namespace ymx {

// One output tile: 64 * NJ pixels from p0, 256 channels from q0.
template <typename OT, int NJ>
__device__ __forceinline__ void dma4_tile(const ConvArgs &a, const int p0, const int q0, const int nk) {
    constexpr int WQ = 128, WP = 16 * NJ;
    constexpr int NI = WQ / 16;
    constexpr int PIECE = 128 * 128;                        // 128 rows x 64 bf16
    constexpr unsigned OOB = 0x80000000u;
    typedef EpiGeom<NI, 64, WP> Epi;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    YMX_STAMP(0);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wq = wave & 1, wp = wave >> 1;                // wave tile position (as in the other kernels)
    const int group = wave >> 2;
    const i32x4_t xrsrc = make_rsrc(a.x, a.x_bytes);
    const i32x4_t wrsrc = make_rsrc(a.w, a.w_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

    // ---- DMA geometry ------------------------------------------------------------------
    // A piece is 16 instructions of 8 rows x 128 B; this wave issues instructions 2*wave and
    // 2*wave+1, i.e. piece rows 16*wave + 8*i + srow.  Row r of an X piece is pixel
    // (r>>5)*WP + half*32 + (r&31) of the tile (rows past the wave's WP pixels stay empty), row r of a W piece is channel
    // (r>>6)*128 + half*64 + (r&63).  The lane that fills 16-byte slot s of a row fetches
    // K-chunk s ^ (row & 7) (the swizzle the fragment reads undo).
    const int srow = lane >> 3;
    const int chunk = (lane & 7) ^ srow;
    int xoff[2][2];         // [pixel half][instruction]
    unsigned xmask[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 16 * wave + 8 * i + srow;
            pixel_geometry(a, p0 + (r >> 5) * WP + h * 32 + (r & 31), xoff[h][i], xmask[h][i]);
            if (h * 32 + (r & 31) >= WP) xmask[h][i] = 0u;     // no such pixel in a 48-pixel wave tile
        }
    unsigned woff[2];       // byte offset of this lane's chunk in K tile 0: [instruction], channel half 0
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * wave + 8 * i + srow;
        woff[i] = (unsigned)(((q0 + (r >> 6) * 128 + (r & 63)) * a.Kw + chunk * 8) * 2);
    }
    const unsigned whalf = (unsigned)(64 * a.Kw * 2);       // channel half 1 = 64 rows further

    // Issue piece `pos` (0: W(m0), 1: X(n0), 2: X(n1), 3: W(m1)) of K tile kt into its slot.
    // Tiles past the end of K and taps outside the image use out-of-range offsets.
    int tapoff = 0;
    unsigned tapbit = 0;
    auto issue = [&](int kt, auto pos_tag) {
        constexpr int POS = decltype(pos_tag)::value;
        const unsigned sbase = lds0 + (unsigned)(((kt & 1) * 4 + POS) * PIECE + wave * 2048);
        if constexpr (POS == 0 || POS == 3) {
            const unsigned base = kt < nk ? (unsigned)kt * 128u + (POS == 3 ? whalf : 0u) : OOB;
#pragma unroll
            for (int i = 0; i < 2; ++i) lds_dma16(wrsrc, sbase + (unsigned)(i * 1024), kt < nk ? base + woff[i] : OOB);
        } else {
            if constexpr (POS == 1) {       // K index -> (tap, channel) -> (row, col), once per tile
                if (a.CB == 64 && (a.k == 1 || a.k == 3)) {
                    // whole 64-channel blocks: a K tile is one (block, tap), the same for every lane,
                    // so the decode is scalar arithmetic
                    const int kk = a.k * a.k;
                    const int blk = kk == 1 ? kt : (int)(((unsigned)kt * 7282u) >> 16);      // kt / 9 for kt < 32768
                    const int tap = kt - blk * kk;
                    const int tr = kk == 1 ? 0 : (tap * 11) >> 5, tc = tap - tr * a.k;       // tap / 3 for tap < 9
                    tapoff = (tr * a.W + tc) * a.x_ld + blk * 64 + chunk * 8;
                    tapbit = kt < nk ? (1u << tap) : 0u;
                } else {
                    k_decode(a, kt * 64 + chunk * 8, tapoff, tapbit);
                }
            }
            constexpr int H = POS - 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned off = (xmask[H][i] & tapbit) ? (unsigned)(xoff[H][i] + tapoff) * 2u : OOB;
                lds_dma16(xrsrc, sbase + (unsigned)(i * 1024), off);
            }
        }
    };

    // Loop form of the weight pieces: wcur[1][i] / wcur[0][i] are this lane's offsets of W(m1) in
    // K tile kt + 1 and of W(m0) in K tile kt + 2, advanced by 128 bytes per tile.
    // Tiles past the end of K fetch whatever lies behind the row (or zeros past the buffer) into
    // slots that nothing reads any more.
    unsigned wcur[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { wcur[0][i] = woff[i] + 256u; wcur[1][i] = woff[i] + whalf + 128u; }
    auto issue_w1_next = [&](int kt) {                  // W(m1) of tile kt + 1
        const unsigned sbase = lds0 + (unsigned)((((kt + 1) & 1) * 4 + 3) * PIECE + wave * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16_steady(wrsrc, sbase + (unsigned)(i * 1024), wcur[1][i]);
    };
    auto issue_w0_next2 = [&](int kt) {                 // W(m0) of tile kt + 2
        const unsigned sbase = lds0 + (unsigned)(((kt & 1) * 4) * PIECE + wave * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16_steady(wrsrc, sbase + (unsigned)(i * 1024), wcur[0][i]);
    };

    f32x4_t acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment reads: piece row = wave part + block*16 + (lane & 15); chunk kk*4 + (lane >> 4)
    const int swz0 = (((lane >> 4)) ^ (lane & 7)) << 4;
    const int swz1 = ((4 + (lane >> 4)) ^ (lane & 7)) << 4;
    const int a_row = (wq * 64 + (lane & 15)) * 128;        // + ib * 2048 within W(m0) / W(m1)
    const int b_row = (wp * 32 + (lane & 15)) * 128;        // + jb * 2048 within X(n0) / X(n1)

    uint4 af[4][2];         // weight fragments of the current channel half: [block][kk]
    uint4 bfr[2][2][2];     // pixel fragments: [pixel half][block][kk]

    auto read_a = [&](int kt, int mh) {
        const char *base = smem + ((kt & 1) * 4 + (mh ? 3 : 0)) * PIECE + a_row;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            af[ib][0] = *(const uint4 *)(base + ib * 2048 + swz0);
            af[ib][1] = *(const uint4 *)(base + ib * 2048 + swz1);
        }
    };
    auto read_b = [&](int kt, int nh) {
        const char *base = smem + ((kt & 1) * 4 + 1 + nh) * PIECE + b_row;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            if (nh * 2 + jb >= NJ) continue;
            bfr[nh][jb][0] = *(const uint4 *)(base + jb * 2048 + swz0);
            bfr[nh][jb][1] = *(const uint4 *)(base + jb * 2048 + swz1);
        }
    };
    auto mfma_quadrant = [&](auto mh_tag, auto nh_tag) {
        constexpr int MH = decltype(mh_tag)::value, NHALF = decltype(nh_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    if constexpr (NHALF * 2 + 1 < NJ) {
                        acc[MH * 4 + ib][NHALF * 2 + jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, af[ib][kk]), __builtin_bit_cast(bf16x8_t, bfr[NHALF][jb][kk]),
                            acc[MH * 4 + ib][NHALF * 2 + jb], 0, 0, 0);
                    } else {
                        if (jb == 0)
                            acc[MH * 4 + ib][NHALF * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8_t, af[ib][kk]), __builtin_bit_cast(bf16x8_t, bfr[NHALF][0][kk]),
                                acc[MH * 4 + ib][NHALF * 2], 0, 0, 0);
                    }
        __builtin_amdgcn_s_setprio(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;

    // ---- prologue: pieces 0..5 in flight, pieces 0, 1 and 2 landed -----------------------
    issue(0, I0{});
    issue(0, I1{});
    issue(0, I2{});
    issue(0, I3{});
    issue(1, I0{});
    issue(1, I1{});
    YMX_STAMP(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    YMX_STAMP(2);
    if (group == 1) __builtin_amdgcn_s_barrier();          // second group runs one barrier behind

    // A raw barrier (no implied drain of the memory counters) that the compiler may not move
    // LDS reads across: a read hoisted over it would see a piece some wave has not waited for.
    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // Four intervals per K tile: a wave alternates between a LOAD interval (fragment reads, DMA
    // issue, waits) and an MFMA interval of two quadrants (32 MFMAs); the second group runs one
    // barrier behind, so on every SIMD one wave computes while the other loads.
    //   load 1:  A(m0), B(n0), B(n1) of tile kt; issue X(n1), W(m1) of tile kt+1
    //   mfma 1:  (m0,n0) (m0,n1)
    //   load 2:  A(m1) of tile kt; issue W(m0), X(n0) of tile kt+2
    //   mfma 2:  (m1,n1) (m1,n0)
    // Eight DMA instructions per wave and K tile in a fixed order.  vmcnt(8) at the end of load 1
    // = "W(m1) of THIS tile has landed" (read in load 2: one more barrier for this group, and
    // the other group's same wait lies before that barrier too); vmcnt(6) at the end of load 2 =
    // "W(m0), X(n0), X(n1) of the next tile have landed" (read in its load 1).  A load interval
    // ends with lgkmcnt(0), so its reads are complete before the barrier behind it and the slot
    // of a piece may be re-issued from the next interval on; every re-issue here is later.
    YMX_REGION_DECL;
    for (int kt = 0; kt < nk; ++kt) {
        YMX_REGION_BEGIN(1);
        read_a(kt, 0);
        read_b(kt, 0);
        read_b(kt, 1);
        YMX_REGION_END(1);
        YMX_REGION_BEGIN(12);
        issue(kt + 1, I2{});
        issue_w1_next(kt);
        YMX_REGION_END(12);
        YMX_REGION_BEGIN(2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        YMX_REGION_END(2);
        YMX_REGION_BEGIN(3);
        barrier();
        YMX_REGION_END(3);
        YMX_REGION_BEGIN(4);
        mfma_quadrant(I0{}, I0{});
        mfma_quadrant(I0{}, I1{});
        YMX_REGION_END(4);
        YMX_REGION_BEGIN(5);
        barrier();
        YMX_REGION_END(5);
        YMX_REGION_BEGIN(6);
        read_a(kt, 1);
        YMX_REGION_END(6);
        YMX_REGION_BEGIN(7);
        issue_w0_next2(kt);
        issue(kt + 2, I1{});
#pragma unroll
        for (int i = 0; i < 2; ++i) { wcur[0][i] += 128u; wcur[1][i] += 128u; }
        YMX_REGION_END(7);
        YMX_REGION_BEGIN(8);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        YMX_REGION_END(8);
        YMX_REGION_BEGIN(9);
        barrier();
        YMX_REGION_END(9);
        YMX_REGION_BEGIN(10);
        mfma_quadrant(I1{}, I1{});
        mfma_quadrant(I1{}, I0{});
        YMX_REGION_END(10);
        YMX_REGION_BEGIN(11);
        barrier();
        YMX_REGION_END(11);
    }
    YMX_REGION_STORE(wave, lane);
    if (group == 0) barrier();

    // first half of the shortcut input (the fragment registers are free now)
    const int cbase = q0 + wq * WQ, pbase = p0 + wp * WP;
    epi_u32x4_t rpre[Epi::NH][Epi::NPASS];
    if (epi_res_pre(a)) {
        epi_load_res<NI, 64, WP>(a, 0, cbase, pbase, lane, rpre[0]);
    } else {
#pragma unroll
        for (int n = 0; n < Epi::NPASS; ++n) rpre[0][n] = epi_u32x4_t{0u, 0u, 0u, 0u};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    YMX_STAMP(3);

    // ---- epilogue: per-wave transpose through LDS (conv_epilogue.h); LDS is free now ----
    static_assert(8 * Epi::LDS_BYTES <= 8 * PIECE, "per-wave transpose images must fit in LDS");
    epi_run<OT, NI, NJ>(a, acc, smem + wave * Epi::LDS_BYTES, cbase, pbase, lane, rpre);
    YMX_STAMP_END(5);
}

template <typename OT, int NJ>
__global__ void __launch_bounds__(512, 2)
conv_igemm_dma4_kernel(const ConvArgs a, const int ntile_q, const int nk) {
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    dma4_tile<OT, NJ>(a, (id / ntile_q) * (64 * NJ), (id % ntile_q) * 256, nk);
}

// Both tile forms in one launch (see conv_igemm_halo_mixed_kernel): the first `nbig` workgroups take
// 256-pixel tiles over the pixels below `pbig`, the rest 192-pixel tiles from there on.
template <typename OT>
__global__ void __launch_bounds__(512, 2)
conv_igemm_dma4_mixed_kernel(const ConvArgs a, const int ntile_q, const int nk, const int nbig, const int pbig) {
    const int b = blockIdx.x;
    if (b < nbig) {
        const int id = xcd_remap(b, nbig);
        dma4_tile<OT, 4>(a, (id / ntile_q) * 256, (id % ntile_q) * 256, nk);
    } else {
        const int id = xcd_remap(b - nbig, (int)gridDim.x - nbig);
        dma4_tile<OT, 3>(a, pbig + (id / ntile_q) * 192, (id % ntile_q) * 256, nk);
    }
}

thread_local int g_dma4_pixels = 0;      // 0: pick the tile per layer; 256 / 192: force it (tuning option "dma4_pixels")

// Pixels per tile for a layer, by the rule of conv_halo_tile_pixels: a launch takes as many rounds
// as its tiles need on the chip's 256 CUs (one workgroup per CU), and a round of 192-pixel tiles
// costs about 0.9 of a round of 256-pixel tiles (measured 0.87-0.97 on the detector's 1x1 and
// stride-2 layers, whose 8-36 K tiles leave the fixed costs of a tile more weight than the 3x3
// layers of the resident-slab kernel do): 192 wins where it does not add a round.
TilePlan conv_dma4_plan(const ConvArgs &a) {
    return plan_wide_tiles(a.P, (a.Cout + 255) / 256, 0.9, g_dma4_pixels, g_halo_big_rounds);
}
int conv_dma4_tile_pixels(const ConvArgs &a) { return conv_dma4_plan(a).pixels; }

template <typename OT, int NJ>
static int launch_dma4_cfg(const ConvArgs &a, hipStream_t s) {
    constexpr int LDS = 8 * 128 * 128;
    constexpr int TP = 64 * NJ;
    auto kern = conv_igemm_dma4_kernel<OT, NJ>;
    static LdsReserve reserved;        // one per kernel instantiation; remembers the devices already set up
    if (const int rc = reserve_lds(reserved, (const void *)kern, LDS, "conv_igemm_dma4")) return rc;
    const int ntile_p = (a.P + TP - 1) / TP;
    const int ntile_q = (a.Cout + 255) / 256;
    const int nk = (a.K + 63) / 64;
    const long long grid = (long long)ntile_p * ntile_q;
    if (grid > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_igemm_dma4: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, a, ntile_q, nk);
    return check_launch("conv_igemm_dma4");
}

template <typename OT>
static int launch_dma4_mixed(const ConvArgs &a, const TilePlan &pl, hipStream_t s) {
    constexpr int LDS = 8 * 128 * 128;
    auto kern = conv_igemm_dma4_mixed_kernel<OT>;
    static LdsReserve reserved;
    if (const int rc = reserve_lds(reserved, (const void *)kern, LDS, "conv_igemm_dma4 (mixed)")) return rc;
    const int ntile_q = (a.Cout + 255) / 256;
    const long long rest = (long long)a.P - pl.big_pixels;
    if (pl.big_tiles <= 0 || (pl.big_tiles & 7) || rest <= 0 || (long long)pl.big_tiles * 256 != (long long)pl.big_pixels * ntile_q)
        return fail(YMX_EARG, "conv_igemm_dma4: inconsistent mixed tiling");
    const long long grid = pl.big_tiles + (rest + 191) / 192 * ntile_q;
    if (grid > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_igemm_dma4: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, a, ntile_q, (a.K + 63) / 64, pl.big_tiles, pl.big_pixels);
    return check_launch("conv_igemm_dma4 (mixed)");
}

// bf16 operands only; out_dtype bf16 or f32.
int launch_conv_igemm_dma4(const ConvArgs &a, hipStream_t s) {
    if (a.dtype != YMX_BF16) return fail(YMX_EUNSUPPORTED, "conv_igemm_dma4: bf16 operands only");
    const TilePlan pl = conv_dma4_plan(a);
    if (pl.pixels == 448) {
        if (a.z) return launch_dma4_mixed<bf16_zout_t>(a, pl, s);
        if (a.d2s_c) return launch_dma4_mixed<bf16_d2s_t>(a, pl, s);
        if (a.out_dtype == YMX_BF16) return launch_dma4_mixed<bf16_t>(a, pl, s);
        return launch_dma4_mixed<float>(a, pl, s);
    }
    if (pl.pixels == 192) {
        if (a.z) return launch_dma4_cfg<bf16_zout_t, 3>(a, s);
        if (a.d2s_c) return launch_dma4_cfg<bf16_d2s_t, 3>(a, s);
        if (a.out_dtype == YMX_BF16) return launch_dma4_cfg<bf16_t, 3>(a, s);
        return launch_dma4_cfg<float, 3>(a, s);
    }
    if (a.z) return launch_dma4_cfg<bf16_zout_t, 4>(a, s);
    if (a.d2s_c) return launch_dma4_cfg<bf16_d2s_t, 4>(a, s);
    if (a.out_dtype == YMX_BF16) return launch_dma4_cfg<bf16_t, 4>(a, s);
    return launch_dma4_cfg<float, 4>(a, s);
}

}  // namespace ymx

YMX_STAMP_EXPORT(ymx_debug_stamps_dma4)
#ifdef YMX_REGION_DMA4
YMX_REGION_EXPORT
#endif
