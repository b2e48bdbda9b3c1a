// The first ResidialBlock of the backbone in one kernel (gfx950): x + conv3x3_64(conv1x1_32(x)) on
// a 64-channel map (layers.py:92-106 with filter_num1 = 32, filter_num2 = 64; backbones.py:13),
// each convolution with its folded batch norm / bias and activation, the shortcut added after the
// second activation.
//
// Run as two launches the block reads x twice (as the 1x1 convolution's input and as the shortcut)
// and moves the 32-channel map to HBM and back: 128 + 64 B read and 64 B written by the first
// launch, 64 + 128 B read and 128 B written by the second, per pixel -- 1.4 GB per batch of 64
// images at 208 x 208, against 128 B in and 128 B out (0.7 GB) for the block as a whole.  Here a
// workgroup produces an 8 x 16 tile of the block's output from the 10 x 18 tile of the 1x1
// convolution's output under it, which only ever exists in LDS (the structure of conv_stem.hip):
//
//   phase A   the 180 pixels of the 32-channel tile, 16 at a time per wave: a lane loads 2 x 16
//             bytes of its pixel's 64 input channels straight into the MFMA B operand (K = 64 as two
//             MFMAs per 16-channel half, the K order of the packed weights and of
//             conv_igemm_dma2.hip), affine + activation on pairs, one rounding to bf16, 16 bytes
//             per lane into LDS; pixels outside the image are the 3x3 convolution's zero padding
//   phase B   the 3x3 convolution as 9 K steps of 32 (one filter tap each, again the K order of the
//             packed weights), its 36 KB of weights resident in LDS for the workgroup's life; a wave
//             owns one output row of 16 pixels x all 64 channels
//   epilogue  in the accumulator layout (a lane holds 4 consecutive channels of a pixel): affine,
//             activation, the shortcut (8-byte loads of x issued before phase B) added in fp32, one
//             rounding; the bf16 results go through a 2 KB LDS image private to the wave and leave
//             as 16-byte stores, 128 contiguous bytes per pixel (8-byte stores from the accumulator
//             layout touch every 128-byte line four times: measured, they alone held the kernel at
//             0.34 ms where it takes 0.21 without any store)
//
// so the outputs carry the same bits as the two launches.  LDS 11.25 + 36 + 16 KB: two workgroups
// per CU; workgroups are persistent, and a wave fetches the operands of its blocks of the next tile
// before phase B of the current one.
#include "conv_epilogue.h"

namespace ymx {

typedef __attribute__((ext_vector_type(2))) unsigned int rb_u32x2_t;

constexpr int RB_TH = 8, RB_TW = 16;                      // output tile of a workgroup (rows x columns)
constexpr int RB_MR = RB_TH + 2, RB_MC = RB_TW + 2;       // tile of the 1x1 convolution's output with its halo: 10 x 18
constexpr int RB_NBLK = RB_MR + 2;                        // 16-pixel blocks: columns 0-15 of each row, then columns 16 and 17 of all rows (20 pixels)
constexpr int RB_ROWB = RB_MC * 64;                       // bytes of a tile row: 18 pixels x 32 bf16
constexpr int RB_MIDB = RB_MR * RB_ROWB;                  // 11 520
constexpr int RB_W2ROWB = 9 * 64;                         // one output channel's 9 x 32 weights
constexpr int RB_W2B = 64 * RB_W2ROWB;                    // 36 864
constexpr int RB_STAGEB = 16 * 128;                       // a wave's output row: 16 pixels x 64 bf16
constexpr int RB_AFFB = 2 * 64 * 4;                       // scale and shift of the 3x3 convolution
constexpr int RB_LDS = RB_MIDB + RB_W2B + 8 * RB_STAGEB + RB_AFFB;

struct ResBlockArgs {
    const bf16_t *x;                // [B,H,W,64] bf16, pixel stride x_ld
    unsigned x_bytes;
    int x_ld;
    const bf16_t *w1;               // packed [rows >= 32][Kw1] bf16: the 1x1 convolution, K index = input channel
    int Kw1;
    const float *scale1, *shift1;   // may be null (1 / 0)
    const bf16_t *w2;               // packed [rows >= 64][Kw2] bf16: the 3x3 convolution, K index = tap * 32 + channel
    int Kw2;
    const float *scale2, *shift2;
    bf16_t *y;                      // [B,H,W,64] with pixel stride y_ld
    int y_ld;
    int B, H, W;
    int tiles_x, tiles_y, ntiles;
};

// 16-byte slot of channel chunk c (8 channels) of the tile pixel in column i: slot c ^ rb_swz(i).
// Any 8 consecutive columns then fall on 8 different 16-byte slots of a 128-byte LDS row for every
// chunk (4 even columns in one half, 4 odd ones in the other, each with 4 different keys), which is
// what phase B's fragment reads at column offsets 0, 1, 2 and phase A's writes touch.
__device__ __forceinline__ int rb_swz(int i) { return (i >> 1) & 3; }

template <int ACT>
__global__ void __launch_bounds__(512, 4)
conv_resblock_kernel(const ResBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const mid = smem;
    char *const w2s = smem + RB_MIDB;
    char *const aff = smem + RB_MIDB + RB_W2B + 8 * RB_STAGEB;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4, li = lane & 15;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- 3x3 weights into LDS: row n at n * 576, tap t at + t * 64, chunk q at slot q ^ f(n & 15)
    for (int c = t; c < 64 * 36; c += 512) {
        const int n = c / 36, kq = c - n * 36;
        const uint4 v = *(const uint4 *)(a.w2 + (size_t)n * a.Kw2 + kq * 8);
        const int q = kq & 3, tap = kq >> 2;
        *(uint4 *)(w2s + n * RB_W2ROWB + tap * 64 + ((q ^ ((-((n & 15) >> 2)) & 3)) << 4)) = v;
    }

    if (t < 64) {
        ((float *)aff)[t] = a.scale2 ? a.scale2[t] : 1.0f;
        ((float *)aff)[64 + t] = a.shift2 ? a.shift2[t] : 0.0f;
    }

    // ---- 1x1 weights as MFMA A operands.  Row li of MFMA tt is output channel 8 (li >> 2) + 4 tt +
    // (li & 3), so that a lane's accumulators (rows 4g .. 4g+3 of each) are the 8 consecutive
    // channels 8g .. 8g+7 of its pixel: one 16-byte LDS store per lane.
    bf16x8_t a1[2][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int ch = 8 * (li >> 2) + 4 * tt + (li & 3);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            a1[tt][kk] = __builtin_bit_cast(bf16x8_t, *(const uint4 *)(a.w1 + (size_t)ch * a.Kw1 + kk * 32 + 8 * g));
    }
    float sc0[8], sh0[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        sc0[r] = a.scale1 ? a.scale1[8 * g + r] : 1.0f;
        sh0[r] = a.shift1 ? a.shift1[8 * g + r] : 0.0f;
    }

    // ---- tiles and blocks ----
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
        tl.r0 = ty * RB_TH;
        tl.c0 = (rem - ty * a.tiles_x) * RB_TW;
        tl.base = (tl.b * a.H + tl.r0 - 1) * a.W + tl.c0 - 1;
        tl.border = tl.r0 == 0 || tl.c0 == 0 || tl.r0 + RB_TH + 1 > a.H || tl.c0 + RB_TW + 1 > a.W;
        return tl;
    };
    // tile row / column of this lane's pixel in block blk (a scalar pair for the row blocks)
    const int q1 = li, q2 = 16 + li;                         // pixel number inside the two column blocks (valid below 20)
    auto block_px = [&](int blk, int &mr, int &mc, bool &valid) {
        if (blk < RB_MR) { mr = blk; mc = li; valid = true; }
        else {
            const int q = blk == RB_MR ? q1 : q2;
            mr = q >> 1; mc = RB_TW + (q & 1); valid = q < 2 * RB_MR;
        }
    };
    const unsigned xpix = (unsigned)a.x_ld * 2u;             // bytes per pixel of x

    // operand gather of one block: lane (pixel li, K group g) loads channels 8g .. 8g+7 and 32+8g .. 32+8g+7
    auto issue = [&](const Tile &tl, int blk, uint4 (&p)[2]) {
        if (!tl.live || blk >= RB_NBLK) return;
        int mr, mc;
        bool valid;
        block_px(blk, mr, mc, valid);
        if (tl.border) {
            const int iy = tl.r0 - 1 + mr, ix = tl.c0 - 1 + mc;
            valid = valid && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        }
        const unsigned off = valid ? (unsigned)(tl.base + mr * a.W + mc) * xpix + (unsigned)(g * 16) : OOB;
        p[0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        p[1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, valid ? off + 64u : OOB, 0, 0));
    };
    auto compute = [&](const Tile &tl, int blk, const uint4 (&p)[2]) {
        f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0][kk], __builtin_bit_cast(bf16x8_t, p[kk]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[1][kk], __builtin_bit_cast(bf16x8_t, p[kk]), d1, 0, 0, 0);
        }
        uint32_t oq[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            oq[r] = epi_pack2(affine_act2<ACT>(f32x2_t{d0[2 * r], d0[2 * r + 1]}, f32x2_t{sc0[2 * r], sc0[2 * r + 1]},
                                               f32x2_t{sh0[2 * r], sh0[2 * r + 1]}));
            oq[2 + r] = epi_pack2(affine_act2<ACT>(f32x2_t{d1[2 * r], d1[2 * r + 1]}, f32x2_t{sc0[4 + 2 * r], sc0[5 + 2 * r]},
                                                   f32x2_t{sh0[4 + 2 * r], sh0[5 + 2 * r]}));
        }
        uint4 out = make_uint4(oq[0], oq[1], oq[2], oq[3]);
        int mr, mc;
        bool valid;
        block_px(blk, mr, mc, valid);
        if (tl.border) {
            const int iy = tl.r0 - 1 + mr, ix = tl.c0 - 1 + mc;
            if (!((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)) out = make_uint4(0u, 0u, 0u, 0u);   // the 3x3 convolution's zero padding
        }
        if (valid) *(uint4 *)(mid + mr * RB_ROWB + mc * 64 + ((g ^ rb_swz(mc)) << 4)) = out;
    };

    // phase B geometry: wave = output row of the tile; lane = (channel of a 16-channel block or pixel li, K chunk g)
    const int a_off = (li * RB_W2ROWB) + ((g ^ ((-(li >> 2)) & 3)) << 4);                   // + i * 16 rows + tap * 64
    int b_off[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) b_off[c] = wave * RB_ROWB + (li + c) * 64 + ((g ^ rb_swz(li + c)) << 4);
    char *const stage = smem + RB_MIDB + RB_W2B + wave * RB_STAGEB;
    // staging image: pixel li at li * 128, 16-byte slot s (8 channels) at (s ^ (li & 7)) * 16
    const int st_w = li * 128 + (g & 1) * 8;                  // + ((2 i + (g >> 1)) ^ (li & 7)) * 16
    const int st_px = lane >> 3, st_slot = lane & 7;           // read-back: pixels st_px and st_px + 8, channels 8 st_slot .. + 7

    const int nb = (RB_NBLK - wave + 7) / 8;                   // blocks of this wave per tile: wave, wave + 8
    uint4 pa[2], pb[2];
    Tile cur = tile_of(blockIdx.x);
    issue(cur, wave, pa);
    if (nb > 1) issue(cur, wave + 8, pb);
    __syncthreads();                                           // the weights are in LDS

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        // ---- phase A ----
        compute(cur, wave, pa);
        if (nb > 1) compute(cur, wave + 8, pb);
        const Tile nxt = tile_of(tile + gridDim.x);
        issue(nxt, wave, pa);                                  // land during phase B and the epilogue
        if (nb > 1) issue(nxt, wave + 8, pb);
        // the shortcut: this lane's output pixel (row `wave`, column li), channels 16 i + 4g .. + 3
        rb_u32x2_t res[4];
        const int orow = cur.r0 + wave, oc = cur.c0 + li;
        {
            const bool in = orow < a.H && oc < a.W;
            const unsigned off = in ? (unsigned)((cur.b * a.H + orow) * a.W + oc) * xpix + (unsigned)(g * 8) : OOB;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                res[i] = __builtin_bit_cast(rb_u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(xr, in ? off + (unsigned)(i * 32) : OOB, 0, 0));
        }
        __syncthreads();

        // ---- phase B: 9 taps x 4 channel blocks ----
        f32x4_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, c = tap - 3 * r;
            uint4 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const uint4 *)(w2s + a_off + i * 16 * RB_W2ROWB + tap * 64);
            const uint4 bf = *(const uint4 *)(mid + r * RB_ROWB + b_off[c]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                 __builtin_bit_cast(bf16x8_t, bf), acc[i], 0, 0, 0);
        }
        __syncthreads();                                       // every wave has read the tile: the next phase A may overwrite it

        // ---- epilogue: lane = (pixel column li, channels 4g .. 4g+3 of each 16-channel block) ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 sc = *(const float4 *)(aff + (i * 16 + 4 * g) * 4);
            const float4 sh = *(const float4 *)(aff + 256 + (i * 16 + 4 * g) * 4);
            f32x2_t o0 = act2<ACT>(affine2(f32x2_t{acc[i][0], acc[i][1]}, f32x2_t{sc.x, sc.y}, f32x2_t{sh.x, sh.y}));
            f32x2_t o1 = act2<ACT>(affine2(f32x2_t{acc[i][2], acc[i][3]}, f32x2_t{sc.z, sc.w}, f32x2_t{sh.z, sh.w}));
            // a separately rounded add, as in conv_epilogue.h: the same bits as the two launches
            const unsigned w0 = res[i][0], w1 = res[i][1];
            o0 = epi_add2(o0, f32x2_t{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)});
            o1 = epi_add2(o1, f32x2_t{__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)});
            *(rb_u32x2_t *)(stage + st_w + (((2 * i + (g >> 1)) ^ (li & 7)) << 4)) = rb_u32x2_t{epi_pack2(o0), epi_pack2(o1)};
        }
        // (the image is private to the wave: its own LDS accesses complete in order)
        if (orow < a.H) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int px = h * 8 + st_px, col = cur.c0 + px;
                const epi_u32x4_t v = *(const epi_u32x4_t *)(stage + px * 128 + ((st_slot ^ (px & 7)) << 4));
                if (col < a.W)
                    __builtin_nontemporal_store(v, (epi_u32x4_t *)(a.y + ((size_t)(cur.b * a.H + orow) * a.W + col) * a.y_ld + st_slot * 8));
            }
        }
        cur = nxt;
    }
}

bool conv_resblock_supported(int act) { return act == YMX_ACT_MISH || act == YMX_ACT_LEAKY; }

int launch_conv_resblock(const void *x, int x_ld, const void *w1, const float *scale1, const float *shift1, const void *w2,
                         const float *scale2, const float *shift2, void *y, int y_ld, int B, int H, int W, int act, hipStream_t s) {
    if (!conv_resblock_supported(act)) return fail(YMX_EUNSUPPORTED, "conv_resblock: activation %d (Mish or leaky)", act);
    ResBlockArgs a;
    a.x_ld = x_ld ? x_ld : 64;
    a.y_ld = y_ld ? y_ld : 64;
    const long long xb = (long long)B * H * W * a.x_ld * 2;
    if (xb >= 0x80000000LL) return fail(YMX_EUNSUPPORTED, "conv_resblock: input of %lld bytes (2 GB at most)", xb);
    a.x = (const bf16_t *)x; a.x_bytes = (unsigned)xb;
    a.w1 = (const bf16_t *)w1; a.Kw1 = packed_kw(1, 64); a.scale1 = scale1; a.shift1 = shift1;
    a.w2 = (const bf16_t *)w2; a.Kw2 = packed_kw(3, 32); a.scale2 = scale2; a.shift2 = shift2;
    a.y = (bf16_t *)y;
    a.B = B; a.H = H; a.W = W;
    a.tiles_x = (W + RB_TW - 1) / RB_TW; a.tiles_y = (H + RB_TH - 1) / RB_TH;
    const long long nt = (long long)B * a.tiles_x * a.tiles_y;
    if (nt > 0x7fffffffLL) return fail(YMX_EUNSUPPORTED, "conv_resblock: too many pixels");
    a.ntiles = (int)nt;
    const int grid = a.ntiles < 512 ? a.ntiles : 512;           // two persistent workgroups per CU
    static LdsReserve reserved[2];
    if (act == YMX_ACT_MISH) {
        auto kern = conv_resblock_kernel<YMX_ACT_MISH>;
        if (const int rc = reserve_lds(reserved[0], (const void *)kern, RB_LDS, "conv_resblock")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), RB_LDS, s, a);
    } else {
        auto kern = conv_resblock_kernel<YMX_ACT_LEAKY>;
        if (const int rc = reserve_lds(reserved[1], (const void *)kern, RB_LDS, "conv_resblock")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), RB_LDS, s, a);
    }
    return check_launch("conv_resblock");
}

}  // namespace ymx
