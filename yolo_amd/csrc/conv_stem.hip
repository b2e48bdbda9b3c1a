// The backbone's first two layers in one kernel (gfx950): 3x3 stride-1 convolution of the fp32
// image to 32 channels, then the 3x3 stride-2 convolution to 64 channels (backbones.py:11-12,
// the same two YOLOConv layers open CSPDarknet53), each with its folded batch norm / bias and
// activation.
//
// Run as two launches the pair moves the 32-channel full-resolution map to HBM and back:
// 64 B written and 64 B read per input pixel, 1.4 GB per batch of 64 images at 416 x 416,
// more than the image (12 B per pixel) and the result (32 B per input pixel) together.
// Here a workgroup produces an 8 x 16 tile of the second layer's output from a 17 x 33 tile
// of the first layer's output that only ever exists in LDS:
//
//   phase A   the 561 first-layer pixels of the tile, 16 at a time per wave, exactly as
//             conv_first_rows_kernel computes them (conv_direct.hip: operand gather with
//             three 12-byte buffer loads per lane, K = 27 as two MFMAs per 16-channel half,
//             affine + activation on pairs, one rounding to bf16) -- written to LDS instead
//             of memory; pixels outside the image are the second layer's zero padding
//   phase B   second layer as 9 K steps of 32 (one filter tap each, the K order of the packed
//             weights and of conv_igemm_dma2.hip: the same products are added in the same
//             order), weights resident in LDS for the workgroup's whole life; a wave owns
//             32 channels x 2 output rows of 16 pixels
//   epilogue  from the accumulator layout (a lane holds 4 consecutive channels of a pixel):
//             affine, activation, one rounding, 8-byte stores
//
// so the outputs carry the same bits as the two-launch path.  LDS: the first-layer tile is kept
// as two column-parity planes per row (a stride-2 tap then reads 16 consecutive entries of one
// plane), 16-byte chunks XOR-swizzled; 35.1 KB + 36 KB of weights = two workgroups per CU.
// Workgroups are persistent (a grid of two per CU walks the tiles); a wave fetches the operands
// of its first block of the next tile before it starts phase B of the current one.
#include "conv_epilogue.h"

namespace ymx {

typedef __attribute__((ext_vector_type(3))) unsigned int stem_u32x3_t;
typedef __attribute__((ext_vector_type(2))) unsigned int stem_u32x2_t;

constexpr int STEM_TH = 8, STEM_TW = 16;                               // output tile of a workgroup (rows x columns)
constexpr int STEM_LR = 2 * STEM_TH + 1, STEM_LC = 2 * STEM_TW + 1;    // first-layer tile with its halo: 17 x 33
constexpr int STEM_NPX = STEM_LR * STEM_LC;                            // 561 pixels
constexpr int STEM_NBLK = (STEM_NPX + 15) / 16;                        // 36 blocks of 16
constexpr int STEM_NE = STEM_TW + 1;                                   // entries of the plane of tile columns 0, 2, .. 32
constexpr int STEM_ROWB = STEM_LC * 64;                                // bytes of a tile row: 33 pixels x 32 bf16
constexpr int STEM_L0B = STEM_LR * STEM_ROWB;                          // 35 904
constexpr int STEM_W1ROWB = 9 * 64;                                    // one output channel's 9 x 32 weights
constexpr int STEM_W1B = 64 * STEM_W1ROWB;                             // 36 864
constexpr int STEM_LDS = STEM_L0B + STEM_W1B;

struct StemArgs {
    const float *x;                 // [B,H,W,3] fp32, dense
    unsigned x_bytes;
    const float *w0;                // [27][32] fp32 (HWIO)
    const float *scale0, *shift0;   // may be null (1 / 0)
    const bf16_t *w1;               // packed [rows >= 64][Kw1] bf16, K index = tap * 32 + channel
    int Kw1;
    const float *scale1, *shift1;
    bf16_t *y;                      // [B,OH,OW,64] with pixel stride y_ld
    int y_ld;
    int B, H, W, OH, OW;
    int tiles_x, tiles_y, ntiles;
};

// 16-byte slot of channel chunk c (8 channels) of plane entry i: conflict-free for the fragment
// reads of phase B (16 consecutive entries per 16-lane group) and for phase A's writes.
__device__ __forceinline__ int stem_swz(int i) { return ((-(i >> 2)) & 3) ^ (i & 3); }

template <int ACT>
__global__ void __launch_bounds__(512, 4)
conv_stem_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const l0 = smem;
    char *const w1s = smem + STEM_L0B;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4, li = lane & 15;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);

    // ---- second-layer weights into LDS: row n at n * 576, tap t at + t * 64, chunk q at slot q ^ f(n & 15)
    for (int c = t; c < 64 * 36; c += 512) {
        const int n = c / 36, kq = c - n * 36;
        const uint4 v = *(const uint4 *)(a.w1 + (size_t)n * a.Kw1 + kq * 8);
        const int q = kq & 3, tap = kq >> 2;
        *(uint4 *)(w1s + n * STEM_W1ROWB + tap * 64 + ((q ^ ((-((n & 15) >> 2)) & 3)) << 4)) = v;
    }

    // ---- first-layer weights as MFMA A operands (the layout of conv_first_rows_kernel) ----
    bf16x8_t a1[2], a2[2];
    {
        const int gg = g < 3 ? g : 0;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int ch = 8 * (li >> 2) + 4 * tt + (li & 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = a.w0[(gg * 9 + j) * 32 + ch];
                a1[tt][j] = (__bf16)(g < 3 ? w : 0.0f);
            }
            const float w8 = a.w0[(gg * 9 + 8) * 32 + ch];
            a2[tt][0] = (__bf16)(g < 3 ? w8 : 0.0f);
#pragma unroll
            for (int j = 1; j < 8; ++j) a2[tt][j] = (__bf16)0.0f;
        }
    }
    float sc0[8], sh0[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        sc0[r] = a.scale0 ? a.scale0[8 * g + r] : 1.0f;
        sh0[r] = a.shift0 ? a.shift0[8 * g + r] : 0.0f;
    }

    // ---- tiles and blocks ----
    // A tile's 561 first-layer pixels are taken 16 at a time: blocks 0 .. 33 are the two halves
    // (tile columns 0-15, 16-31) of the 17 rows, blocks 34 and 35 the last column (tile column 32)
    // of rows 0-15 and of row 16.  Row and half of a row block are scalars, so a lane's image
    // offset and LDS address are a per-tile, per-block scalar plus a constant of the lane.
    const int tpi = a.tiles_x * a.tiles_y;
    struct Tile {
        int b, r0, c0;
        int base;           // element offset (x 3 floats) of image pixel (2 r0 - 2, 2 c0 - 2): tile row 0's tap row 0, left neighbour of column 0
        bool live, border;  // border: some pixel of the tile or of its 3x3 neighbourhoods lies outside the image
    };
    auto tile_of = [&](int tile) {
        Tile tl;
        tl.live = tile < a.ntiles;
        const int tt = tl.live ? tile : 0;
        tl.b = tt / tpi;
        const int rem = tt - tl.b * tpi;
        const int ty = rem / a.tiles_x;
        tl.r0 = ty * STEM_TH;
        tl.c0 = (rem - ty * a.tiles_x) * STEM_TW;
        tl.base = (tl.b * a.H + 2 * tl.r0 - 2) * a.W + 2 * tl.c0 - 2;
        tl.border = tl.r0 == 0 || tl.c0 == 0 || 2 * tl.r0 + STEM_LR > a.H || 2 * tl.c0 + STEM_LC > a.W;
        return tl;
    };
    constexpr unsigned OOB = 0x80000000u;
    const unsigned gl_row = g < 3 ? (unsigned)((g * a.W + li) * 12) : OOB;     // lane part of a row block's byte offset (k-group 3: zeros)
    const unsigned gl_col = g < 3 ? (unsigned)(((li + g) * a.W + 32) * 12) : OOB;
    const int ll_row = ((li & 1) ? STEM_NE * 64 : 0) + (li >> 1) * 64 + ((g ^ stem_swz(li >> 1)) << 4);   // LDS: half 0 of a row
    const int ll_col = li * STEM_ROWB + 16 * 64 + ((g ^ stem_swz(16)) << 4);

    // operand gather of one block: lane (pixel li, tap row g) loads the pixels ix-1, ix, ix+1 of image row iy+g-1
    auto issue = [&](const Tile &tl, int blk, stem_u32x3_t (&p)[3]) {
        if (!tl.live || blk >= STEM_NBLK) return;
        unsigned off;
        if (blk < 2 * STEM_LR) off = (unsigned)((tl.base + (blk >> 1) * a.W + (blk & 1) * 16) * 12) + gl_row;
        else off = (unsigned)((tl.base + (blk - 2 * STEM_LR) * 16 * a.W) * 12) + gl_col;
        unsigned o0 = off, o1 = off + 12u, o2 = off + 24u;
        if (tl.border) {
            int iy, ix;
            if (blk < 2 * STEM_LR) { iy = 2 * tl.r0 - 1 + (blk >> 1); ix = 2 * tl.c0 - 1 + (blk & 1) * 16 + li; }
            else { iy = 2 * tl.r0 - 1 + (blk - 2 * STEM_LR) * 16 + li; ix = 2 * tl.c0 + 31; }
            const bool rok = g < 3 && (unsigned)(iy + g - 1) < (unsigned)a.H;
            o0 = (rok && (unsigned)(ix - 1) < (unsigned)a.W) ? o0 : OOB;
            o1 = (rok && (unsigned)ix < (unsigned)a.W) ? o1 : OOB;
            o2 = (rok && (unsigned)(ix + 1) < (unsigned)a.W) ? o2 : OOB;
        }
        p[0] = __builtin_amdgcn_raw_buffer_load_b96(xr, o0, 0, 0);
        p[1] = __builtin_amdgcn_raw_buffer_load_b96(xr, o1, 0, 0);
        p[2] = __builtin_amdgcn_raw_buffer_load_b96(xr, o2, 0, 0);
    };
    auto compute = [&](const Tile &tl, int blk, const stem_u32x3_t (&p)[3]) {
        // K order of conv_first_rows_kernel: (left, centre, right pixel) x 3 channels, the ninth value in a second MFMA
        const uint4 b1 = make_uint4(epi_pack2(f32x2_t{__uint_as_float(p[0][0]), __uint_as_float(p[0][1])}),
                                    epi_pack2(f32x2_t{__uint_as_float(p[0][2]), __uint_as_float(p[1][0])}),
                                    epi_pack2(f32x2_t{__uint_as_float(p[1][1]), __uint_as_float(p[1][2])}),
                                    epi_pack2(f32x2_t{__uint_as_float(p[2][0]), __uint_as_float(p[2][1])}));
        const uint4 b2 = make_uint4(epi_pack2(f32x2_t{__uint_as_float(p[2][2]), 0.0f}), 0u, 0u, 0u);
        f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0], __builtin_bit_cast(bf16x8_t, b1), d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[1], __builtin_bit_cast(bf16x8_t, b1), d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0], __builtin_bit_cast(bf16x8_t, b2), d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[1], __builtin_bit_cast(bf16x8_t, b2), d1, 0, 0, 0);
        // channels 8g .. 8g+3 from d0, 8g+4 .. 8g+7 from d1; neighbours stay a pair from the affine step to the rounding
        uint32_t oq[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            oq[r] = epi_pack2(affine_act2<ACT>(f32x2_t{d0[2 * r], d0[2 * r + 1]}, f32x2_t{sc0[2 * r], sc0[2 * r + 1]},
                                               f32x2_t{sh0[2 * r], sh0[2 * r + 1]}));
            oq[2 + r] = epi_pack2(affine_act2<ACT>(f32x2_t{d1[2 * r], d1[2 * r + 1]}, f32x2_t{sc0[4 + 2 * r], sc0[5 + 2 * r]},
                                                   f32x2_t{sh0[4 + 2 * r], sh0[5 + 2 * r]}));
        }
        uint4 out = make_uint4(oq[0], oq[1], oq[2], oq[3]);
        int addr;
        bool wr = true;
        if (blk < 2 * STEM_LR) {
            const int h = blk & 1;
            addr = (blk >> 1) * STEM_ROWB + h * 512 + (ll_row ^ (h * 32));
            if (tl.border) {
                const int iy = 2 * tl.r0 - 1 + (blk >> 1), ix = 2 * tl.c0 - 1 + h * 16 + li;
                if (!((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)) out = make_uint4(0u, 0u, 0u, 0u);
            }
        } else {
            const int rowv = (blk - 2 * STEM_LR) * 16 + li;
            addr = (blk - 2 * STEM_LR) * 16 * STEM_ROWB + ll_col;
            wr = rowv < STEM_LR;
            if (tl.border) {
                const int iy = 2 * tl.r0 - 1 + rowv, ix = 2 * tl.c0 + 31;
                if (!((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)) out = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (wr) *(uint4 *)(l0 + addr) = out;                   // outside the image: the second layer's zero padding
    };

    // phase B geometry: wave = (channel half wq, row pair wp); lane = (channel or pixel li, K chunk g)
    const int wq = wave & 1, wp = wave >> 1;
    const int a_off = ((wq * 32 + li) * STEM_W1ROWB) + ((g ^ ((-(li >> 2)) & 3)) << 4);      // + i * 16 rows + tap * 64
    const int b_e0 = li * 64 + ((g ^ stem_swz(li)) << 4);                                     // plane of even tile columns, entry li
    const int b_e1 = (li + 1) * 64 + ((g ^ stem_swz(li + 1)) << 4);                           // ... entry li + 1
    const int b_o0 = (STEM_NE + li) * 64 + ((g ^ stem_swz(li)) << 4);                         // plane of odd tile columns, entry li
    float sc1[2][4], sh1[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = wq * 32 + i * 16 + 4 * g + r;
            sc1[i][r] = a.scale1 ? a.scale1[c] : 1.0f;
            sh1[i][r] = a.shift1 ? a.shift1[c] : 0.0f;
        }

    const int nb = (STEM_NBLK - wave + 7) / 8;                 // blocks of this wave per tile: wave, wave + 8, ...
    stem_u32x3_t pa[3], pb[3];
    Tile cur = tile_of(blockIdx.x);
    issue(cur, wave, pa);
    __syncthreads();                                           // the weights are in LDS

    YMX_REGION_DECL;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        // ---- phase A ----
        YMX_REGION_BEGIN(1);
        for (int k = 0; k < nb; k += 2) {
            issue(cur, k + 1 < nb ? wave + 8 * (k + 1) : STEM_NBLK, pb);
            compute(cur, wave + 8 * k, pa);
            if (k + 1 >= nb) break;
            if (k + 2 < nb) issue(cur, wave + 8 * (k + 2), pa);
            compute(cur, wave + 8 * (k + 1), pb);
        }
        const Tile nxt = tile_of(tile + gridDim.x);
        issue(nxt, wave, pa);                                  // lands during phase B and the epilogue
        YMX_REGION_END(1);
        YMX_REGION_BEGIN(2);
        __syncthreads();
        YMX_REGION_END(2);
        YMX_REGION_BEGIN(3);

        // ---- phase B: 9 taps x (2 channel blocks x 2 output rows) ----
        f32x4_t acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, c = tap - 3 * r;
            const int boff = c == 0 ? b_e0 : c == 1 ? b_o0 : b_e1;
            uint4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const uint4 *)(w1s + a_off + i * 16 * STEM_W1ROWB + tap * 64);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const uint4 *)(l0 + (2 * (2 * wp + j) + r) * STEM_ROWB + boff);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                        __builtin_bit_cast(bf16x8_t, bf[j]), acc[i][j], 0, 0, 0);
        }
        YMX_REGION_END(3);
        YMX_REGION_BEGIN(4);
        __syncthreads();                                       // every wave has read the tile: the next phase A may overwrite it
        YMX_REGION_END(4);
        YMX_REGION_BEGIN(5);

        // ---- epilogue: lane = (pixel column li, channels 4g .. 4g+3 of each 16-channel block) ----
        // The bf16 results go through a 2 KB image private to the wave (in the tile's LDS, which every
        // wave is done with) and leave as 16-byte stores, 64 contiguous bytes per pixel: 8-byte stores
        // straight from the accumulator layout touch every 128-byte line of y eight times.
        // image: pixel p = 16 j + li at p * 64, 16-byte slot s (8 channels) at (s ^ ((p >> 1) & 3)) * 16
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = 16 * j + li;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x2_t o0 = act2<ACT>(affine2(f32x2_t{acc[i][j][0], acc[i][j][1]}, f32x2_t{sc1[i][0], sc1[i][1]},
                                                     f32x2_t{sh1[i][0], sh1[i][1]}));
                const f32x2_t o1 = act2<ACT>(affine2(f32x2_t{acc[i][j][2], acc[i][j][3]}, f32x2_t{sc1[i][2], sc1[i][3]},
                                                     f32x2_t{sh1[i][2], sh1[i][3]}));
                *(stem_u32x2_t *)(l0 + wave * 2048 + p * 64 + (((2 * i + (g >> 1)) ^ ((p >> 1) & 3)) << 4) + (g & 1) * 8) =
                    stem_u32x2_t{epi_pack2(o0), epi_pack2(o1)};
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = 16 * h + (lane >> 2), s = lane & 3;         // h = the wave's output row, lane >> 2 = column, s = 8-channel chunk
            const int orow = cur.r0 + 2 * wp + h, oc = cur.c0 + (lane >> 2);
            const epi_u32x4_t v = *(const epi_u32x4_t *)(l0 + wave * 2048 + p * 64 + ((s ^ ((p >> 1) & 3)) << 4));
            if (orow < a.OH && oc < a.OW)
                __builtin_nontemporal_store(v, (epi_u32x4_t *)(a.y + ((size_t)(cur.b * a.OH + orow) * a.OW + oc) * a.y_ld + wq * 32 + s * 8));
        }
        YMX_REGION_END(5);
        __syncthreads();                                       // the images are read: the next phase A may overwrite the tile
        cur = nxt;
    }
    YMX_REGION_STORE(wave, lane);
}

bool conv_stem_supported(int act) { return act == YMX_ACT_MISH || act == YMX_ACT_LEAKY; }

int launch_conv_stem(const float *x, const float *w0, const float *scale0, const float *shift0, const void *w1, const float *scale1,
                     const float *shift1, void *y, int B, int H, int W, int act, int y_ld, hipStream_t s) {
    if (!conv_stem_supported(act)) return fail(YMX_EUNSUPPORTED, "conv_stem: activation %d (Mish or leaky)", act);
    const long long xb = (long long)B * H * W * 12;
    if (xb >= 0x80000000LL) return fail(YMX_EUNSUPPORTED, "conv_stem: image batch of %lld bytes (2 GB at most)", xb);
    StemArgs a;
    a.x = x; a.x_bytes = (unsigned)xb; a.w0 = w0; a.scale0 = scale0; a.shift0 = shift0;
    a.w1 = (const bf16_t *)w1; a.Kw1 = packed_kw(3, 32); a.scale1 = scale1; a.shift1 = shift1;
    a.y = (bf16_t *)y; a.y_ld = y_ld ? y_ld : 64;
    a.B = B; a.H = H; a.W = W;
    a.OH = (H + 1 - 3) / 2 + 1; a.OW = (W + 1 - 3) / 2 + 1;
    a.tiles_x = (a.OW + STEM_TW - 1) / STEM_TW; a.tiles_y = (a.OH + STEM_TH - 1) / STEM_TH;
    const long long nt = (long long)B * a.tiles_x * a.tiles_y;
    if (nt > 0x7fffffffLL || (long long)B * H * W >= 0x20000000LL) return fail(YMX_EUNSUPPORTED, "conv_stem: too many pixels");
    a.ntiles = (int)nt;
    const int grid = a.ntiles < 512 ? a.ntiles : 512;           // two persistent workgroups per CU
    static LdsReserve reserved[2];
    if (act == YMX_ACT_MISH) {
        auto kern = conv_stem_kernel<YMX_ACT_MISH>;
        if (const int rc = reserve_lds(reserved[0], (const void *)kern, STEM_LDS, "conv_stem")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), STEM_LDS, s, a);
    } else {
        auto kern = conv_stem_kernel<YMX_ACT_LEAKY>;
        if (const int rc = reserve_lds(reserved[1], (const void *)kern, STEM_LDS, "conv_stem")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), STEM_LDS, s, a);
    }
    return check_launch("conv_stem");
}

}  // namespace ymx

#ifdef YMX_REGION_STEM
YMX_REGION_EXPORT
#endif
