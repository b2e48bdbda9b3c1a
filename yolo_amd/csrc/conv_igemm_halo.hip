// bf16 3x3 stride-1 convolution, 256 x 256 tile, input pixels resident in LDS across the
// nine filter taps (gfx950).
//
// The LDS-DMA kernels of conv_igemm_dma*.hip move (256 pixels + 256 channels) x 64 K per
// 64-deep K tile from L2 into LDS, about 32 B per clock per CU at full MFMA rate -- which is
// what one CU can pull from its L2, so the copy and the matrix pipe share the critical path.
// Half of those bytes are input pixels, and for a 3x3 convolution they are the same pixels
// nine times over, shifted by one row or column.  This kernel stages them once:
//
//   K order    channel blocks of 64 outermost, the nine taps inside (the order of the
//              packed weights, see common.h), so one block of input channels serves nine
//              consecutive K tiles
//   slab       for a tile of 256 consecutive output pixels (row-major over images, rows,
//              columns) and one channel block, the inputs of all nine taps are the
//              contiguous pixel run [p0 - W - 1, p0 + 255 + W + 1]: 256 + 2W + 2 rows of
//              128 B.  Tap (r, c) of output pixel p0 + n reads slab row n + r*W + c.  Two
//              slabs (current block, next block) stay in LDS; the next one is fetched in
//              the background, one DMA instruction per wave per K tile
//   padding    slab rows exist for every pixel of the run, so a tap that falls outside the
//              image would read a real neighbour (the previous row's last pixel, the next
//              image): for such (pixel, tap) pairs, known from the per-pixel validity masks
//              of pixel_geometry, the fragment read is pointed at a zero-filled KB instead
//   weights    streamed per K tile as in conv_igemm_dma4.hip: two 16 KB pieces W(m0), W(m1)
//              (the waves' first and second 64 channels), four ring slots
//   schedule   the phase-staggered wave groups of conv_igemm_dma4.hip: four phases per K
//              tile (quadrants (m0,n0) (m0,n1) (m1,n1) (m1,n0)), one group does its LDS
//              reads and DMA issue while the other issues its 16 MFMAs.  DMA per wave and
//              K tile: 5 instructions instead of 8
//
// Per accumulator the K terms are added in the same order as in every other MFMA kernel
// here, so the results are bit-identical to theirs.
//
// LDS: 2 slabs x ceil((258 + 2W) / 8) KB + 64 KB weight ring + 1 KB that absorbs the
// no-op DMA instructions which keep the per-tile instruction count constant (the counted
// vmcnt waits rely on it): W <= 59 with 256-pixel tiles; 192-pixel tiles (slabs of
// 194 + 2W rows, at most 47 KB each) reach W <= 91, e.g. the 76-wide maps of a 608 x 608 input.
#include "conv_epilogue.h"

namespace ymx {

// NJ = 16-pixel blocks per wave: 4 (256-pixel tiles) or 3 (192-pixel tiles, for layers whose
// 256-pixel tiles fill the last round of CUs badly).  With three blocks the second pixel half
// of a wave is one block, so the phases carry 16, 8, 8 and 16 MFMAs; loads, waits and barriers
// are the same in both forms.
// One output tile: TP = 64 * NJ pixels from p0, 256 channels from q0.
template <typename OT, int NJ>
__device__ __forceinline__ void halo_tile(const ConvArgs &a, const int p0, const int q0, const int ncb, const int npieces) {
    constexpr int WQ = 128, WP = 16 * NJ;
    constexpr int NI = WQ / 16;
    constexpr int WPIECE = 128 * 128;                       // 128 channel rows x 64 bf16
    constexpr unsigned OOB = 0x80000000u;
    typedef EpiGeom<NI, 64, WP> Epi;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    YMX_STAMP(0);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wq = wave & 1, wp = wave >> 1;                // wave tile position (as in the other kernels)
    const int group = wave >> 2;
    const int nk = ncb * 9;
    const i32x4_t xrsrc = make_rsrc(a.x, a.x_bytes);
    const i32x4_t wrsrc = make_rsrc(a.w, a.w_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const int slab_bytes = npieces * 1024;
    const unsigned ring0 = lds0 + 2u * (unsigned)slab_bytes;       // weight ring: slot (kt&1)*2 + half
    const unsigned dump = ring0 + 4u * WPIECE;
    const char *ring = smem + 2 * slab_bytes;

    // ---- DMA geometry ------------------------------------------------------------------
    // Every DMA instruction fills 8 rows x 128 B; the lane that fills 16-byte slot s of row
    // `row` fetches K-chunk s ^ (row & 7), the swizzle the fragment reads undo.
    const int srow = lane >> 3;
    const int chunk = (lane & 7) ^ srow;
    // slab piece i = slab rows 8i .. 8i+7 = input pixels p0 - W - 1 + 8i + srow (any image)
    // (modular unsigned arithmetic: a pixel before the tensor wraps to an offset beyond it and reads zeros)
    const unsigned slab_off0 = ((unsigned)(p0 - a.W - 1 + srow) * (unsigned)a.x_ld + (unsigned)(chunk * 8)) * 2u;
    const unsigned slab_step = 16u * (unsigned)a.x_ld;                          // bytes from one piece to the next
    // weight pieces as in conv_igemm_dma4.hip: this wave fills piece rows 16*wave + 8*i + srow,
    // row r = channel (r>>6)*128 + half*64 + (r&63)
    unsigned woff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * wave + 8 * i + srow;
        woff[i] = (unsigned)(((q0 + (r >> 6) * 128 + (r & 63)) * a.Kw + chunk * 8) * 2);
    }
    const unsigned whalf = (unsigned)(64 * a.Kw * 2);

    auto issue_w = [&](int kt, int half) {              // prologue form
        const unsigned sbase = ring0 + (unsigned)(((kt & 1) * 2 + half) * WPIECE + wave * 2048);
        const unsigned base = (unsigned)kt * 128u + (half ? whalf : 0u);
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(wrsrc, sbase + (unsigned)(i * 1024), kt < nk ? base + woff[i] : OOB);
    };
    // Loop form: wcur[1][i] / wcur[0][i] are this lane's offsets of W(m1) in K tile kt + 1 and of
    // W(m0) in K tile kt + 2, advanced by 128 bytes per tile.  Tiles past the end of K
    // (the last two issues) fetch whatever lies behind the row -- or zeros past the buffer -- into
    // ring slots that nothing reads any more.
    unsigned wcur[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { wcur[0][i] = woff[i] + 256u; wcur[1][i] = woff[i] + whalf + 128u; }
    auto issue_w_next = [&](int kt) {                   // W(m1) of tile kt + 1
        const unsigned sbase = ring0 + (unsigned)((((kt + 1) & 1) * 2 + 1) * WPIECE + wave * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16_steady(wrsrc, sbase + (unsigned)(i * 1024), wcur[1][i]);
    };
    auto issue_w_next2 = [&](int kt) {                  // W(m0) of tile kt + 2
        const unsigned sbase = ring0 + (unsigned)(((kt & 1) * 2) * WPIECE + wave * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16_steady(wrsrc, sbase + (unsigned)(i * 1024), wcur[0][i]);
    };
    // slab piece `idx` of channel block cb into slab cb&1; out-of-range idx / cb: a no-op
    // instruction (zeros into the spare KB) so that the instruction count stays constant
    auto issue_slab = [&](int cb, int idx) {
        const bool real = idx < npieces && cb < ncb;
        const unsigned dst = real ? lds0 + (unsigned)((cb & 1) * slab_bytes + idx * 1024) : dump;
        const unsigned src = real ? slab_off0 + (unsigned)idx * slab_step + (unsigned)cb * 128u : OOB;
        lds_dma16(xrsrc, dst, src);
    };
    auto issue_slab_steady = [&](int cb, int idx) {
        const bool real = idx < npieces && cb < ncb;
        const unsigned dst = real ? lds0 + (unsigned)((cb & 1) * slab_bytes + idx * 1024) : dump;
        const unsigned src = real ? slab_off0 + (unsigned)idx * slab_step + (unsigned)cb * 128u : OOB;
        lds_dma16_steady(xrsrc, dst, src);
    };

    // ---- prologue: slab of block 0, weights of tile 0 and W(m0) of tile 1 --------------------
    // (issued before the fragment geometry below is worked out, which then runs under the loads)
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) issue_slab(0, jj * 8 + wave);
    lds_dma16(xrsrc, dump, OOB);                           // the spare KB starts out (and stays) zero
    issue_w(0, 0);
    issue_w(0, 1);
    issue_w(1, 0);

    // ---- fragment geometry ---------------------------------------------------------------
    // this lane's four pixels (16-pixel blocks j of the wave's 64) and which taps they have
    unsigned fmask[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int unused;
        pixel_geometry(a, p0 + wp * WP + j * 16 + (lane & 15), unused, fmask[j]);
    }
    const int hb0 = wp * WP + (lane & 15);                  // slab row of block 0, tap (0, 0)
    const int q4 = lane >> 4;
    const int swz0 = ((q4) ^ (lane & 7)) << 4;              // weight rows: row & 7 == lane & 7
    const int swz1 = ((4 + q4) ^ (lane & 7)) << 4;
    const int a_row = (wq * 64 + (lane & 15)) * 128;

    f32x4_t acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    uint4 af[4][2];         // weight fragments of the current channel half: [block][kk]
    uint4 bfr[2][2][2];     // pixel fragments: [pixel half][block][kk]

    auto read_a = [&](int kt, int mh) {
        const char *base = ring + ((kt & 1) * 2 + mh) * WPIECE + a_row;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            af[ib][0] = *(const uint4 *)(base + ib * 2048 + swz0);
            af[ib][1] = *(const uint4 *)(base + ib * 2048 + swz1);
        }
    };
    // pixel half nh of the current tap: slab rows hb + (2*nh + jb) * 16.  Where the tap falls
    // outside the image the lane reads the always-zero spare KB instead (one select on the
    // address; lanes sharing that address are a broadcast, not a bank conflict)
    typedef unsigned int frag_u32x4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const frag_u32x4_t *lds_frag_t;
    const unsigned zero_abs = lds0 + (unsigned)(2 * slab_bytes + 4 * WPIECE);
    auto read_b = [&](int slab_off, int shift, int tap, int nh, int set, bool live) {
        const int hb = hb0 + shift;
        const unsigned base = (lds0 + (unsigned)slab_off) + (unsigned)(hb * 128 + ((q4 ^ (hb & 7)) << 4));     // (hb + 16 j) & 7 == hb & 7
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int j = nh * 2 + jb;
            if (j >= NJ) continue;
            // all ones where the tap lies inside the image for this lane's pixel, else zero: picks the slab row or the zero row
            const unsigned in = live ? (unsigned)__builtin_amdgcn_sbfe((int)fmask[j], (unsigned)tap, 1u) : 0u;
            const unsigned off = ((base + (unsigned)(j * 2048)) & in) | (zero_abs & ~in);
            bfr[set][jb][0] = __builtin_bit_cast(uint4, *(lds_frag_t)(uintptr_t)off);
            bfr[set][jb][1] = __builtin_bit_cast(uint4, *(lds_frag_t)(uintptr_t)(off ^ 64u));          // chunk 4 + q4
        }
    };
    auto mfma_quadrant = [&](auto mh_tag, auto nh_tag, auto set_tag) {
        constexpr int MH = decltype(mh_tag)::value, NHALF = decltype(nh_tag)::value, SET = decltype(set_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    if constexpr (NHALF * 2 + 1 < NJ) {
                        acc[MH * 4 + ib][NHALF * 2 + jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, af[ib][kk]), __builtin_bit_cast(bf16x8_t, bfr[SET][jb][kk]),
                            acc[MH * 4 + ib][NHALF * 2 + jb], 0, 0, 0);
                    } else {
                        if (jb == 0)
                            acc[MH * 4 + ib][NHALF * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8_t, af[ib][kk]), __builtin_bit_cast(bf16x8_t, bfr[SET][0][kk]),
                                acc[MH * 4 + ib][NHALF * 2], 0, 0, 0);
                    }
        __builtin_amdgcn_s_setprio(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    // A raw barrier (no implied drain of the memory counters) that the compiler may not move
    // LDS reads across: a read hoisted over it would see data some wave has not waited for.
    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    YMX_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
    YMX_STAMP(2);
    if (group == 1) barrier();                             // second group runs one barrier behind

    // Four intervals per K tile, separated by barriers: a wave alternates between a LOAD interval
    // (fragment reads into registers its last MFMAs are done with, DMA issue, waits) and an
    // MFMA interval of two quadrants (32 MFMAs); the second group runs one barrier behind, so
    // on every SIMD one wave is in its MFMA interval while the other loads.
    //   load 1:  A(m0), B(n0), B(n1) of this tile (16 reads and the tap / border arithmetic)
    //   mfma 1:  quadrants (m0,n0) (m0,n1)
    //   load 2:  A(m1) (8 reads); issue W(m1) of the next tile, W(m0) of the tile after next and
    //            one slab piece
    //   mfma 2:  quadrants (m1,n1) (m1,n0)
    // The loading wave gets only the issue slots the MFMA wave of its SIMD leaves free, so what it
    // has to issue is what an interval costs beyond its 32 MFMAs; measured per K tile with all
    // five DMA instructions in load 1: load 1 = 770 cycles, load 2 = 280, an MFMA interval 580
    // -- hence every DMA instruction sits in load 2 (520 / 480 / 580: -4 .. -6 % per layer).
    // Five DMA instructions per wave and K tile in a fixed order (W(m1) x 2, W(m0) x 2, slab), so
    // the counted waits mean: vmcnt(3) at the end of load 1 = "W(m1) of THIS tile (issued in the
    // last tile's load 2) has landed" -- read in load 2, after one more barrier for this group and
    // after the other group's same wait; vmcnt(6) at the end of load 2 = "W(m0) of the next tile
    // (issued one tile ago) has landed" -- read in its load 1.  A slab piece is covered by the
    // vmcnt(3) one tile later.  Every load interval ends with lgkmcnt(0): a fragment read is
    // complete before the barrier that follows it, so a ring slot or slab may be re-issued from
    // the next interval on (the other group's reads of it ended one barrier earlier still).
    int cb = 0, tap = 0;
    YMX_REGION_DECL;
    for (int kt = 0; kt < nk; ++kt) {
        const int slab = (cb & 1) * slab_bytes;
        const int tr = (tap * 11) >> 5;                     // tap / 3 for tap < 9
        const int shift = tr * a.W + (tap - tr * 3);
        // load 1
        YMX_REGION_BEGIN(1);
        read_a(kt, 0);
        read_b(slab, shift, tap, 0, 0, true);
        read_b(slab, shift, tap, 1, 1, true);
        YMX_REGION_END(1);
        YMX_REGION_BEGIN(2);
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        YMX_REGION_END(2);
        YMX_REGION_BEGIN(3);
        barrier();
        YMX_REGION_END(3);
        // mfma 1
        YMX_REGION_BEGIN(4);
        mfma_quadrant(I0{}, I0{}, I0{});
        mfma_quadrant(I0{}, I1{}, I1{});
        YMX_REGION_END(4);
        YMX_REGION_BEGIN(5);
        barrier();
        YMX_REGION_END(5);
        // load 2
        YMX_REGION_BEGIN(6);
        read_a(kt, 1);
        YMX_REGION_END(6);
        YMX_REGION_BEGIN(7);
        issue_w_next(kt);
        issue_w_next2(kt);
        issue_slab_steady(tap < 6 ? cb + 1 : ncb, tap * 8 + wave);
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
        // mfma 2
        YMX_REGION_BEGIN(10);
        mfma_quadrant(I1{}, I1{}, I1{});
        mfma_quadrant(I1{}, I0{}, I0{});
        YMX_REGION_END(10);
        YMX_REGION_BEGIN(11);
        barrier();
        YMX_REGION_END(11);
        if (++tap == 9) { tap = 0; ++cb; }
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
    epi_run<OT, NI, NJ>(a, acc, smem + wave * Epi::LDS_BYTES, cbase, pbase, lane, rpre);
    YMX_STAMP_END(5);
}

template <typename OT, int NJ>
__global__ void __launch_bounds__(512, 2)
conv_igemm_halo_kernel(const ConvArgs a, const int ntile_q, const int ncb, const int npieces) {
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tq = id % ntile_q, tp = id / ntile_q;
    halo_tile<OT, NJ>(a, tp * (64 * NJ), tq * 256, ncb, npieces);
}

// Both tile forms in one launch: the first `nbig` workgroups (a whole number of rounds of the
// chip) take 256-pixel tiles over the pixels below `pbig`, the rest 192-pixel tiles over the
// pixels from there on.  A launch of 2.6 rounds of 256-pixel tiles then costs two such rounds
// and a shorter one instead of three (workgroups are dispatched in id order, so the short tiles
// run last).  Each segment keeps the XCD-contiguous tile order of its own (nbig is a multiple of 8).
template <typename OT>
__global__ void __launch_bounds__(512, 2)
conv_igemm_halo_mixed_kernel(const ConvArgs a, const int ntile_q, const int ncb, const int npieces4, const int npieces3,
                             const int nbig, const int pbig) {
    const int b = blockIdx.x;
    if (b < nbig) {
        const int id = xcd_remap(b, nbig);
        halo_tile<OT, 4>(a, (id / ntile_q) * 256, (id % ntile_q) * 256, ncb, npieces4);
    } else {
        const int id = xcd_remap(b - nbig, (int)gridDim.x - nbig);
        halo_tile<OT, 3>(a, pbig + (id / ntile_q) * 192, (id % ntile_q) * 256, ncb, npieces3);
    }
}

// The shapes this kernel takes: 3x3, stride 1, whole 64-channel blocks, rows short enough
// for two slabs to fit in LDS.
bool conv_halo_eligible(const ConvArgs &a) {
    return a.dtype == YMX_BF16 && a.k == 3 && a.stride == 1 && a.Cin % 64 == 0 && a.W <= 91;
}

thread_local int g_tile_mix = 1;         // 0: the layer's own choice is between the two pure forms only (tuning option "tile_mix")
thread_local int g_halo_big_rounds = 0;  // mixed form: whole rounds of 256-pixel tiles (0: all but the last; tuning option "halo_big_rounds")
thread_local int g_halo_pixels = 0;      // 0: pick the tile form per layer; 256 / 192: force it; 448: force the mixed form where one exists (tuning option "halo_pixels")

// How a wide layer's launch is tiled (this kernel and conv_igemm_dma4.hip).  A launch takes as many
// rounds as its tiles need on the chip's 256 CUs (one workgroup per CU), and a round of 192-pixel
// tiles takes `f192` of a round of 256-pixel tiles (three quarters of the MFMAs behind the same
// barriers and fixed costs).  Layers with few tiles gain from the short form: 172 tiles of 256
// pixels leave a third of the CUs idle for a whole round where 228 tiles of 192 fill the chip for
// a shorter one.  The mixed form runs whole rounds of 256-pixel tiles and the remaining pixels as
// 192-pixel tiles: 676 tiles of 256 pixels are three rounds, 512 of them plus 219 short ones two
// rounds and a short one; 2 704 tiles (10.56 rounds) run as 2 560 + 192 short ones.  The few
// candidates nearest the end of the launch are priced (more short rounds can beat fewer: 5.28
// rounds measure faster as 4 + 2 short ones than as 5 + a sparse short one).  Measured gains of
// the mixed form exceed the priced ones (the short tiles at the end also run beside fewer busy
// CUs: 512 -> 1024 @ 52 x 52, batch 64: 1.20 ms as 11 rounds, 1.09 as 10 + 1 short), so a small
// margin suffices.
TilePlan plan_wide_tiles(long long P, long long ntq, double f192, int forced_pixels, int forced_rounds) {
    TilePlan pl = {192, 0, 0};
    const long long t256 = (P + 255) / 256, t192 = (P + 191) / 192;
    const long long r256 = (t256 * ntq + 255) / 256;
    const long long r192 = (t192 * ntq + 255) / 256;
    double cmix = 1e30;
    long long big_ptiles = 0;
    if (r256 >= 2 && ntq <= 256) {
        const long long first = forced_rounds > 0 ? (forced_rounds < r256 ? forced_rounds : r256 - 1) : (r256 > 4 ? r256 - 4 : 1);
        const long long last = forced_rounds > 0 ? first : r256 - 1;
        for (long long rounds = first; rounds <= last; ++rounds) {
            const long long bp = rounds * 256 / ntq;
            const long long rest = P - bp * 256;
            if (bp <= 0 || rest <= 0 || bp * ntq % 8 != 0 || bp * 256 >= 0x7fffffffLL) continue;
            // the short tiles: whole rounds after a few rounds of big tiles; after many, the workgroups have drifted
            // apart and a partly filled last round costs little more than its share (2 560 + 192 tiles measure as
            // 10.0 big rounds, 512 + 219 as 2.66, 256 + 110 as 1.88)
            const long long n3 = (rest + 191) / 192 * ntq;
            const double whole = (double)((n3 + 255) / 256), frac = (double)n3 / 256.0;
            const double drift = rounds >= 10 ? 1.0 : (double)rounds / 10.0;
            const double c = (double)(bp * ntq) / 256.0 + (whole - (whole - frac) * drift) * f192;
            if (c < cmix) { cmix = c; big_ptiles = bp; }
        }
    }
    const double c256 = (double)r256, c192 = (double)r192 * f192;
    if (forced_pixels == 256) { pl.pixels = 256; return pl; }
    if (forced_pixels == 192) return pl;
    if (cmix < 1e29 && (forced_pixels == 448 || (g_tile_mix && cmix < 0.985 * (c256 < c192 ? c256 : c192)))) {
        pl.pixels = 448;
        pl.big_tiles = (int)(big_ptiles * ntq);
        pl.big_pixels = (int)(big_ptiles * 256);
        return pl;
    }
    pl.pixels = c192 < c256 * 0.97 ? 192 : 256;
    return pl;
}

TilePlan conv_halo_plan(const ConvArgs &a) {
    if (a.W > 59) return TilePlan{192, 0, 0};                   // two slabs of a 256-pixel tile would not fit in LDS
    const long long ntq = (a.Cout + 255) / 256;
    const long long r256 = (((long long)a.P + 255) / 256 * ntq + 255) / 256;
    // (layers of 512 and more input channels, 72+ K tiles, in launches of a few rounds: the fixed costs of a
    // tile weigh less and a 192-pixel round measures 0.70-0.73 of a 256-pixel one)
    const double f192 = (a.Cin >= 512 && r256 <= 4) ? 0.72 : 0.82;
    return plan_wide_tiles(a.P, ntq, f192, g_halo_pixels, g_halo_big_rounds);
}

int conv_halo_tile_pixels(const ConvArgs &a) { return conv_halo_plan(a).pixels; }

static int halo_lds_bytes(int npieces, int nj) {
    int lds = 2 * npieces * 1024 + 4 * 128 * 128 + 1024;
    const int epi = nj == 4 ? 8 * EpiGeom<8, 64, 64>::LDS_BYTES : 8 * EpiGeom<8, 64, 48>::LDS_BYTES;      // the epilogue's transpose images
    return lds < epi ? epi : lds;
}

template <typename OT, int NJ>
static int launch_halo_cfg(const ConvArgs &a, hipStream_t s) {
    constexpr int TP = 64 * NJ;
    const int npieces = (TP + 2 * a.W + 2 + 7) / 8;
    if (npieces > 47) return fail(YMX_EUNSUPPORTED, "conv_igemm_halo: rows of %d pixels do not fit", a.W);
    const int lds = halo_lds_bytes(npieces, NJ);
    auto kern = conv_igemm_halo_kernel<OT, NJ>;
    static LdsReserve reserved;        // one per kernel instantiation; remembers the devices already set up
    if (const int rc = reserve_lds(reserved, (const void *)kern, 160 * 1024, "conv_igemm_halo")) return rc;
    const int ntile_p = (a.P + TP - 1) / TP;
    const int ntile_q = (a.Cout + 255) / 256;
    const long long grid = (long long)ntile_p * ntile_q;
    if (grid > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_igemm_halo: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, a, ntile_q, a.Cin / 64, npieces);
    return check_launch("conv_igemm_halo");
}

template <typename OT>
static int launch_halo_mixed(const ConvArgs &a, const TilePlan &pl, hipStream_t s) {
    const int np4 = (256 + 2 * a.W + 2 + 7) / 8, np3 = (192 + 2 * a.W + 2 + 7) / 8;
    if (np4 > 47) return fail(YMX_EUNSUPPORTED, "conv_igemm_halo: rows of %d pixels do not fit", a.W);
    const int l4 = halo_lds_bytes(np4, 4), l3 = halo_lds_bytes(np3, 3);
    auto kern = conv_igemm_halo_mixed_kernel<OT>;
    static LdsReserve reserved;
    if (const int rc = reserve_lds(reserved, (const void *)kern, 160 * 1024, "conv_igemm_halo (mixed)")) return rc;
    const int ntile_q = (a.Cout + 255) / 256;
    const long long rest = (long long)a.P - pl.big_pixels;
    if (pl.big_tiles <= 0 || (pl.big_tiles & 7) || rest <= 0 || (long long)pl.big_tiles * 256 != (long long)pl.big_pixels * ntile_q)
        return fail(YMX_EARG, "conv_igemm_halo: inconsistent mixed tiling");
    const long long grid = pl.big_tiles + (rest + 191) / 192 * ntile_q;
    if (grid > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_igemm_halo: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), l4 > l3 ? l4 : l3, s, a, ntile_q, a.Cin / 64, np4, np3,
                       pl.big_tiles, pl.big_pixels);
    return check_launch("conv_igemm_halo (mixed)");
}

int launch_conv_igemm_halo(const ConvArgs &a, hipStream_t s) {
    if (!conv_halo_eligible(a)) return fail(YMX_EUNSUPPORTED, "conv_igemm_halo: shape not supported");
    const TilePlan pl = conv_halo_plan(a);
    if (pl.pixels == 448) {
        if (a.z) return launch_halo_mixed<bf16_zout_t>(a, pl, s);
        if (a.out_dtype == YMX_BF16) return launch_halo_mixed<bf16_t>(a, pl, s);
        return launch_halo_mixed<float>(a, pl, s);
    }
    if (pl.pixels == 192) {
        if (a.z) return launch_halo_cfg<bf16_zout_t, 3>(a, s);
        if (a.out_dtype == YMX_BF16) return launch_halo_cfg<bf16_t, 3>(a, s);
        return launch_halo_cfg<float, 3>(a, s);
    }
    if (a.z) return launch_halo_cfg<bf16_zout_t, 4>(a, s);
    if (a.out_dtype == YMX_BF16) return launch_halo_cfg<bf16_t, 4>(a, s);
    return launch_halo_cfg<float, 4>(a, s);
}

}  // namespace ymx

YMX_STAMP_EXPORT(ymx_debug_stamps_dma5)
#ifdef YMX_REGION_HALO
YMX_REGION_EXPORT
#endif
