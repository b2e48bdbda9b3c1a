// ResidialBlock(64, 128) on a 128-channel map in one kernel (gfx950): x + conv3x3_128(conv1x1_64(x))
// (layers.py:92-106; the two blocks of backbones.py:15-16 at a quarter of the input resolution), each
// convolution with its folded batch norm / bias and activation, the shortcut added after the second
// activation.  The scheme of conv_resblock.hip -- an 8 x 16 output tile per workgroup from the
// 10 x 18 tile of the 1x1 convolution's output under it, which exists only in LDS, x read once for
// the 1x1 convolution and (from the cache it filled) once for the shortcut -- with one change that
// the wider 3x3 convolution forces: its weights are 144 KB, too many for LDS beside the tile, so
// they live in REGISTERS, split by output channel:
//
//   input     the tile's 180 input pixels x 128 channels (45 KB) come into LDS by LDS-DMA, 4 pixels per
//             instruction, issued a whole tile ahead: no register holds them, and both the 1x1
//             convolution and the shortcut read them there (x is read from memory exactly once; a first
//             version that loaded them per lane spent a fifth of its time issuing those loads)
//   phase A   the 180 pixels of the 64-channel tile, 16 at a time per wave: MFMA B operands from the
//             input tile, the 1x1 weights from LDS (16 KB); K = 128 as four MFMAs per 16-channel group,
//             in the K order of the packed weights; affine + activation on pairs, one rounding, 16
//             bytes per lane into LDS
//   phase B   wave w owns output channels 16w .. 16w+15 for ALL 128 pixels of the tile: its 18 weight
//             fragments (9 taps x 2 halves of the 64 input channels) stay in 72 registers for the
//             workgroup's life; it walks the 10 tile rows, and every pixel fragment it reads (row m,
//             column offset c, channel half) feeds the up to three output rows m, m-1, m-2 it is a
//             tap of: 60 LDS reads for 144 MFMAs.  Per accumulator the K terms arrive tap by tap,
//             the two channel halves inside a tap -- the order of the packed weights, so the same bits
//             as conv_igemm_slab2.hip
//   epilogue  affine, activation, shortcut (read from the input tile before it is overwritten) added in fp32, one
//             rounding, in the accumulator layout; the bf16 results of all eight waves meet in a
//             32 KB LDS image of the output tile and leave as 16-byte stores, 256 contiguous bytes
//             per pixel
//
// 218 registers, 116 KB of LDS, one workgroup (8 waves) per CU, persistent.
#include "conv_epilogue.h"

namespace ymx {

typedef __attribute__((ext_vector_type(2))) unsigned int rc_u32x2_t;

constexpr int RC_TH = 8, RC_TW = 16;                      // output tile of a workgroup (rows x columns)
constexpr int RC_MR = RC_TH + 2, RC_MC = RC_TW + 2;       // tile of the 1x1 convolution's output with its halo: 10 x 18
constexpr int RC_NBLK = RC_MR + 2;                        // 16-pixel blocks: columns 0-15 of each row, then columns 16 and 17 of all rows
constexpr int RC_ROWB = RC_MC * 128;                      // bytes of a tile row: 18 pixels x 64 bf16
constexpr int RC_MIDB = RC_MR * RC_ROWB;                  // 23 040
constexpr int RC_W1B = 64 * 256;                          // 1x1 weights: 64 output channels x 128 bf16
constexpr int RC_STAGEB = RC_TH * RC_TW * 256;            // output tile: 128 pixels x 128 bf16
constexpr int RC_AFFB = 2 * 64 * 4;                       // scale and shift of the 1x1 convolution
constexpr int RC_NPX = RC_MR * RC_MC;                     // 180 pixels of the tile with its halo
constexpr int RC_XB = RC_NPX * 256;                       // their 128 input channels: 45 KB, 45 LDS-DMA instructions of 4 pixels
constexpr int RC_LDS = RC_XB + RC_MIDB + RC_W1B + RC_STAGEB + RC_AFFB;

struct ResBlock128Args {
    const bf16_t *x;                // [B,H,W,128] bf16, pixel stride x_ld
    unsigned x_bytes;
    int x_ld;
    const bf16_t *w1;               // packed [rows >= 64][Kw1] bf16: the 1x1 convolution, K index = input channel
    int Kw1;
    const float *scale1, *shift1;   // may be null (1 / 0)
    const bf16_t *w2;               // packed [rows >= 128][Kw2] bf16: the 3x3 convolution, K index = tap * 64 + channel
    int Kw2;
    const float *scale2, *shift2;
    bf16_t *y;                      // [B,H,W,128] with pixel stride y_ld
    int y_ld;
    int B, H, W;
    int tiles_x, tiles_y, ntiles;
};

template <int ACT>
__global__ void __launch_bounds__(512, 2)
conv_resblock128_kernel(const ResBlock128Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const xs = smem;                                    // input pixel q = 18 row + col of the tile at q * 256, 16-byte slot s holds channels 8 (s ^ (q & 15)) .. + 7
    char *const mid = smem + RC_XB;                           // pixel (row, col) at row * RC_ROWB + col * 128, 16-byte slot s at (s ^ (col & 7)) * 16
    char *const w1s = mid + RC_MIDB;                          // output channel n at n * 256, 16-byte slot s (8 input channels) at (s ^ (n & 15)) * 16
    char *const stage = w1s + RC_W1B;                         // output pixel p = 16 row + col at p * 256, slot s at (s ^ (p & 15)) * 16
    float *const aff1 = (float *)(stage + RC_STAGEB);
    const unsigned xs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4, li = lane & 15;
    const i32x4_t xrsrc = make_rsrc(a.x, a.x_bytes);
    constexpr unsigned OOB = 0x80000000u;

    // ---- 1x1 weights and its affine parameters into LDS ----
    for (int c = t; c < 64 * 16; c += 512) {
        const int n = c >> 4, s = c & 15;
        *(uint4 *)(w1s + n * 256 + ((s ^ (n & 15)) << 4)) = *(const uint4 *)(a.w1 + (size_t)n * a.Kw1 + s * 8);
    }
    if (t < 64) {
        aff1[t] = a.scale1 ? a.scale1[t] : 1.0f;
        aff1[64 + t] = a.shift1 ? a.shift1[t] : 0.0f;
    }

    // ---- 3x3 weights of this wave's 16 output channels as MFMA A operands, for the workgroup's life ----
    bf16x8_t w2r[9][2];
    {
        const bf16_t *row = a.w2 + (size_t)(16 * wave + li) * a.Kw2 + 8 * g;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) w2r[tap][kk] = __builtin_bit_cast(bf16x8_t, *(const uint4 *)(row + tap * 64 + kk * 32));
    }
    float sc2[4], sh2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 16 * wave + 4 * g + r;
        sc2[r] = a.scale2 ? a.scale2[c] : 1.0f;
        sh2[r] = a.shift2 ? a.shift2[c] : 0.0f;
    }

    // ---- tiles and blocks (as in conv_resblock.hip) ----
    const int tpi = a.tiles_x * a.tiles_y;
    struct Tile {
        int b, r0, c0;
        int base;           // pixel index of image pixel (r0 - 1, c0 - 1): tile row 0, column 0
        bool live, border;  // border: some pixel of the 10 x 18 tile lies outside the image
    };
    auto tile_of = [&](int tile) {
        Tile tl;
        tl.live = tile < a.ntiles;
        const int tt = tl.live ? tile : 0;
        tl.b = tt / tpi;
        const int rem = tt - tl.b * tpi;
        const int ty = rem / a.tiles_x;
        tl.r0 = ty * RC_TH;
        tl.c0 = (rem - ty * a.tiles_x) * RC_TW;
        tl.base = (tl.b * a.H + tl.r0 - 1) * a.W + tl.c0 - 1;
        tl.border = tl.r0 == 0 || tl.c0 == 0 || tl.r0 + RC_TH + 1 > a.H || tl.c0 + RC_TW + 1 > a.W;
        return tl;
    };
    auto block_px = [&](int blk, int &mr, int &mc, bool &valid) {
        if (blk < RC_MR) { mr = blk; mc = li; valid = true; }
        else {
            const int q = (blk - RC_MR) * 16 + li;               // pixel number inside the two column blocks (valid below 20)
            mr = q >> 1; mc = RC_TW + (q & 1); valid = q < 2 * RC_MR;
        }
    };
    const unsigned xpix = (unsigned)a.x_ld * 2u;             // bytes per pixel of x

    // The tile's input pixels come in by LDS-DMA, 4 pixels (1 KB) per instruction and 45 instructions per tile:
    // instruction i = wave + 8 j, lane l fetches pixel q = 4 i + (l >> 4), channels 8 c .. 8 c + 7 with
    // c = (l & 15) ^ (q & 15) (the swizzle on the source side), and the hardware puts lane l's 16 bytes at 1024 i + 16 l.
    // No register holds them, so a whole tile is in flight while the previous one is being computed; pixels
    // outside the image are out-of-range offsets, for which zeros arrive.
    int dq_row[6], dq_col[6];
    unsigned dq_off[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int q = 4 * (wave + 8 * j) + (lane >> 4);
        dq_row[j] = q / RC_MC;
        dq_col[j] = q - dq_row[j] * RC_MC;
        dq_off[j] = (unsigned)(dq_row[j] * a.W + dq_col[j]) * xpix + (unsigned)((((lane & 15) ^ (q & 15))) << 4);
    }
    auto issue = [&](const Tile &tl) {
        if (!tl.live) return;
        const unsigned base = (unsigned)tl.base * xpix;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int i = wave + 8 * j;
            if (i * 4 >= RC_NPX) continue;
            unsigned off = base + dq_off[j];
            if (tl.border) {
                const int iy = tl.r0 - 1 + dq_row[j], ix = tl.c0 - 1 + dq_col[j];
                if (!((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)) off = OOB;
            }
            lds_dma16(xrsrc, xs_lds + (unsigned)(i * 1024), off);
        }
    };
    // Row li of MFMA (group q, half tt) is output channel 32 q + 8 (li >> 2) + 4 tt + (li & 3), so that a lane's
    // accumulators of a group (rows 4g .. 4g+3 of both halves) are the 8 consecutive channels 32 q + 8g .. + 7.
    const int w1row = 8 * (li >> 2) + (li & 3);               // + 32 q + 4 tt
    auto compute = [&](const Tile &tl, int blk) {
        int mr, mc;
        bool valid;
        block_px(blk, mr, mc, valid);
        uint4 p[4];
        {
            const int q = valid ? mr * RC_MC + mc : 0;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) p[kk] = *(const uint4 *)(xs + q * 256 + (((kk * 4 + g) ^ (q & 15)) << 4));
        }
        bool inside = true;
        if (tl.border) {
            const int iy = tl.r0 - 1 + mr, ix = tl.c0 - 1 + mc;
            inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4_t d[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int n = 32 * q + 4 * tt + w1row;
                    const uint4 af = *(const uint4 *)(w1s + n * 256 + (((kk * 4 + g) ^ (n & 15)) << 4));
                    d[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, p[kk]),
                                                                    d[tt], 0, 0, 0);
                }
            const float4 s0 = *(const float4 *)(aff1 + 32 * q + 8 * g), s1 = *(const float4 *)(aff1 + 32 * q + 8 * g + 4);
            const float4 h0 = *(const float4 *)(aff1 + 64 + 32 * q + 8 * g), h1 = *(const float4 *)(aff1 + 64 + 32 * q + 8 * g + 4);
            uint4 out;
            out.x = epi_pack2(affine_act2<ACT>(f32x2_t{d[0][0], d[0][1]}, f32x2_t{s0.x, s0.y}, f32x2_t{h0.x, h0.y}));
            out.y = epi_pack2(affine_act2<ACT>(f32x2_t{d[0][2], d[0][3]}, f32x2_t{s0.z, s0.w}, f32x2_t{h0.z, h0.w}));
            out.z = epi_pack2(affine_act2<ACT>(f32x2_t{d[1][0], d[1][1]}, f32x2_t{s1.x, s1.y}, f32x2_t{h1.x, h1.y}));
            out.w = epi_pack2(affine_act2<ACT>(f32x2_t{d[1][2], d[1][3]}, f32x2_t{s1.z, s1.w}, f32x2_t{h1.z, h1.w}));
            if (!inside) out = make_uint4(0u, 0u, 0u, 0u);        // the 3x3 convolution's zero padding
            if (valid) *(uint4 *)(mid + mr * RC_ROWB + mc * 128 + (((4 * q + g) ^ (mc & 7)) << 4)) = out;
        }
    };

    // phase B: this lane's pixel fragments: tile row m, column li + c, channels 32 kk + 8g .. + 7
    int b_off[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b_off[c][kk] = (li + c) * 128 + (((4 * kk + g) ^ ((li + c) & 7)) << 4);
    // staging image: this lane writes channels 16 wave + 4g .. + 3 of pixel p = 16 row + li; reads back row `wave`
    const int st_w = li * 256 + (g & 1) * 8;                  // + row * 4096 + ((2 wave + (g >> 1)) ^ (p & 15)) * 16, and p & 15 == li
    const int st_wslot = ((2 * wave + (g >> 1)) ^ li) << 4;

    const int nb = (RC_NBLK - wave + 7) / 8;                   // blocks of this wave per tile: wave, wave + 8
    // the shortcut of this lane's output pixels (row, column li): tile pixel (row + 1, li + 1), channels 16 wave + 4g .. + 3
    int sc_off[RC_TH];
#pragma unroll
    for (int row = 0; row < RC_TH; ++row) {
        const int q = (row + 1) * RC_MC + li + 1;
        sc_off[row] = q * 256 + (((2 * wave + (g >> 1)) ^ (q & 15)) << 4) + (g & 1) * 8;
    }
    Tile cur = tile_of(blockIdx.x);
    issue(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                           // the 1x1 weights and the first tile's pixels are in LDS

    YMX_REGION_DECL;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        // ---- phase A ----
        YMX_REGION_BEGIN(1);
        compute(cur, wave);
        if (nb > 1) compute(cur, wave + 8);
        YMX_REGION_END(1);
        YMX_REGION_BEGIN(2);
        const Tile nxt = tile_of(tile + gridDim.x);
        rc_u32x2_t res[RC_TH];
#pragma unroll
        for (int row = 0; row < RC_TH; ++row) res[row] = *(const rc_u32x2_t *)(xs + sc_off[row]);
        YMX_REGION_END(2);
        YMX_REGION_BEGIN(3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                       // the tile is complete, every wave is done with the input pixels
        issue(nxt);                                            // the next tile's pixels: they land during phase B and the epilogue
        YMX_REGION_END(3);
        YMX_REGION_BEGIN(4);

        // ---- phase B: tile rows m = 0 .. 9, each fragment used by the output rows m - r it is tap row r of ----
        f32x4_t acc[RC_TH];
#pragma unroll
        for (int row = 0; row < RC_TH; ++row) acc[row] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < RC_MR; ++m)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *(const uint4 *)(mid + m * RC_ROWB + b_off[c][kk]));
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int row = m - r;
                        if (row >= 0 && row < RC_TH)
                            acc[row] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2r[r * 3 + c][kk], bf, acc[row], 0, 0, 0);
                    }
                }

        YMX_REGION_END(4);
        YMX_REGION_BEGIN(5);
        // ---- epilogue ----
#pragma unroll
        for (int row = 0; row < RC_TH; ++row) {
            f32x2_t o0 = act2<ACT>(affine2(f32x2_t{acc[row][0], acc[row][1]}, f32x2_t{sc2[0], sc2[1]}, f32x2_t{sh2[0], sh2[1]}));
            f32x2_t o1 = act2<ACT>(affine2(f32x2_t{acc[row][2], acc[row][3]}, f32x2_t{sc2[2], sc2[3]}, f32x2_t{sh2[2], sh2[3]}));
            // a separately rounded add, as in conv_epilogue.h: the same bits as the two launches
            const unsigned w0 = res[row][0], w1 = res[row][1];
            o0 = epi_add2(o0, f32x2_t{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)});
            o1 = epi_add2(o1, f32x2_t{__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)});
            *(rc_u32x2_t *)(stage + row * 4096 + st_w + st_wslot) = rc_u32x2_t{epi_pack2(o0), epi_pack2(o1)};
        }
        YMX_REGION_END(5);
        YMX_REGION_BEGIN(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's part of the next tile's pixels has landed
        __syncthreads();                                       // the output image and the next input tile are complete
        YMX_REGION_END(6);
        YMX_REGION_BEGIN(7);
        {
            const int orow = cur.r0 + wave;
            if (orow < a.H) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int px = h * 4 + g, col = cur.c0 + px;     // lane = (pixel px of the row, 8-channel chunk li)
                    const epi_u32x4_t v = *(const epi_u32x4_t *)(stage + wave * 4096 + px * 256 + ((li ^ px) << 4));
                    if (col < a.W)
                        __builtin_nontemporal_store(v, (epi_u32x4_t *)(a.y + ((size_t)(cur.b * a.H + orow) * a.W + col) * a.y_ld + li * 8));
                }
            }
        }
        YMX_REGION_END(7);
        cur = nxt;
    }
    YMX_REGION_STORE(wave, lane);
}

int launch_conv_resblock128(const void *x, int x_ld, const void *w1, const float *scale1, const float *shift1, const void *w2,
                            const float *scale2, const float *shift2, void *y, int y_ld, int B, int H, int W, int act, hipStream_t s) {
    if (!conv_resblock_supported(act)) return fail(YMX_EUNSUPPORTED, "conv_resblock: activation %d (Mish or leaky)", act);
    ResBlock128Args a;
    a.x_ld = x_ld ? x_ld : 128;
    a.y_ld = y_ld ? y_ld : 128;
    const long long xb = (long long)B * H * W * a.x_ld * 2;
    if (xb >= 0x80000000LL) return fail(YMX_EUNSUPPORTED, "conv_resblock: input of %lld bytes (2 GB at most)", xb);
    a.x = (const bf16_t *)x; a.x_bytes = (unsigned)xb;
    a.w1 = (const bf16_t *)w1; a.Kw1 = packed_kw(1, 128); a.scale1 = scale1; a.shift1 = shift1;
    a.w2 = (const bf16_t *)w2; a.Kw2 = packed_kw(3, 64); a.scale2 = scale2; a.shift2 = shift2;
    a.y = (bf16_t *)y;
    a.B = B; a.H = H; a.W = W;
    a.tiles_x = (W + RC_TW - 1) / RC_TW; a.tiles_y = (H + RC_TH - 1) / RC_TH;
    const long long nt = (long long)B * a.tiles_x * a.tiles_y;
    if (nt > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_resblock: too many pixels");
    a.ntiles = (int)nt;
    const int grid = a.ntiles < 256 ? a.ntiles : 256;           // one persistent workgroup per CU
    static LdsReserve reserved[2];
    if (act == YMX_ACT_MISH) {
        auto kern = conv_resblock128_kernel<YMX_ACT_MISH>;
        if (const int rc = reserve_lds(reserved[0], (const void *)kern, RC_LDS, "conv_resblock128")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), RC_LDS, s, a);
    } else {
        auto kern = conv_resblock128_kernel<YMX_ACT_LEAKY>;
        if (const int rc = reserve_lds(reserved[1], (const void *)kern, RC_LDS, "conv_resblock128")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), RC_LDS, s, a);
    }
    return check_launch("conv_resblock128");
}

}  // namespace ymx

#ifdef YMX_REGION_RB128
YMX_REGION_EXPORT
#endif
