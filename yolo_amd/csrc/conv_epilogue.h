// Epilogue shared by the LDS-DMA convolution kernels: one wave turns its accumulator tile
// (16*NI channels x 64 pixels, MFMA C/D layout: a lane owns 4 channels of one pixel per
// 16x16 block) into finished output rows.
//
// A direct store from that layout would touch 16 pixel rows with 8 bytes each per
// instruction.  Instead, per half of up to 64 channels, the wave dumps the raw fp32
// accumulators into a private LDS image [pixel][HQ channels] (16-byte slots XOR-swizzled
// by the pixel index, conflict-free both ways), then every lane reads back 8 consecutive
// channels of one pixel, applies scale/shift (folded batch norm or bias), the activation
// and the shortcut, rounds once and stores 16 bytes: one store instruction writes complete
// 128-byte (64-byte for 32-channel tiles) runs of pixels.  No workgroup barrier is needed.
//
// The shortcut input is fetched with bounds-checked buffer loads in exactly the store
// pattern, ahead of time (the callers issue the first half's loads before their K loop
// ends), so its latency never sits between the accumulators and the stores.
//
// The common case (channel count and pixel strides multiples of 8) is compiled once per
// activation, so the per-value code is straight-line; anything else takes the generic
// scalar path below it.
#pragma once
#include <type_traits>
#include "lds_dma.h"

namespace ymx {

typedef __attribute__((ext_vector_type(4))) unsigned int epi_u32x4_t;
typedef __bf16 epi_bf16x2_t __attribute__((ext_vector_type(2)));

// Two IEEE additions (no contraction with whatever produced the operands) and one packed
// round-to-nearest-even conversion of a pair: v_pk_add_f32 and v_cvt_pk_bf16_f32.
__device__ __forceinline__ f32x2_t epi_add2(f32x2_t a, f32x2_t b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ uint32_t epi_pack2(f32x2_t v) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, epi_bf16x2_t));
}

// Output element types of the kernels: float, bf16_t, or this tag = bf16 outputs plus the
// value before the activation as a second bf16 output (ConvArgs::z), the form a training
// step's forward pass takes.  A kernel instantiated for the tag contains only that epilogue.
struct bf16_zout_t { bf16_t v; };
// ... and this one = bf16 outputs scattered depth-to-space (ConvArgs::d2s_c), the form the input
// gradient of a stride-2 layer takes.
struct bf16_d2s_t { bf16_t v; };
template <typename T> struct epi_out { typedef T type; static constexpr bool zout = false, d2s = false; };
template <> struct epi_out<bf16_zout_t> { typedef bf16_t type; static constexpr bool zout = true, d2s = false; };
template <> struct epi_out<bf16_d2s_t> { typedef bf16_t type; static constexpr bool zout = false, d2s = true; };

// HQMAX = channels transposed at a time: 64 (128-byte output runs, 16 KB image per wave) or 32
// (64-byte runs, 8 KB per wave: for kernels that trade run length for workgroups per CU).
// NPX = pixels of the wave tile: 64, or 48 for the three-block tiles of conv_igemm_halo.hip.
template <int NI, int HQMAX = 64, int NPX = 64>
struct EpiGeom {
    static constexpr int WQ = NI * 16;                       // channels of the wave tile
    static constexpr int HQ = WQ < HQMAX ? WQ : HQMAX, NH = WQ / HQ;
    static constexpr int SLOTS = HQ / 4, LPP = SLOTS / 2;    // 16-byte slots per fp32 row; lanes per pixel
    static constexpr int PPI = 64 / LPP, NPASS = NPX / PPI;  // pixels per store instruction; instructions per half
    static constexpr int LDS_BYTES = NPX * HQ * 4;           // private image of one wave
    static_assert(NPX % PPI == 0, "whole store instructions");
};

// True when the 16-byte fast path applies.
__device__ __forceinline__ bool epi_vec_ok(const ConvArgs &a) {
    return ((a.Cout & 7) == 0) && ((a.y_ld & 7) == 0) && (a.res == nullptr || (a.res_ld & 7) == 0);
}
// True when the shortcut is fetched ahead of time through epi_load_res.
__device__ __forceinline__ bool epi_res_pre(const ConvArgs &a) {
    return a.res != nullptr && epi_vec_ok(a) && a.res_bytes != 0;
}

// Issues the shortcut loads of one half: cbase/pbase are the wave tile's first channel and
// pixel.  Lanes outside the tensor use an out-of-range offset (zeros, no traffic).
template <int NI, int HQMAX = 64, int NPX = 64>
__device__ __forceinline__ void epi_load_res(const ConvArgs &a, int half, int cbase, int pbase, int lane,
                                             epi_u32x4_t (&r)[EpiGeom<NI, HQMAX, NPX>::NPASS]) {
    typedef EpiGeom<NI, HQMAX, NPX> G;
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)a.res, 0, (int)a.res_bytes, 0x00020000);
    const int co = cbase + half * G::HQ + (lane % G::LPP) * 8;
#pragma unroll
    for (int n = 0; n < G::NPASS; ++n) {
        const int p = pbase + n * G::PPI + lane / G::LPP;
        const unsigned off =
            (p < a.P && co < a.Cout) ? ((unsigned)p * (unsigned)a.res_ld + (unsigned)co) * 2u : 0x80000000u;
        r[n] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0);
    }
}

// rpre[0] must hold the first half's shortcut values if epi_res_pre(a); the second half
// (128-channel wave tiles) is fetched here, while the first half goes out.
template <typename OTAG, int NI, int NJ, int HQMAX = 64>
__device__ __forceinline__ void epi_run(const ConvArgs &a, f32x4_t (&acc)[NI][NJ], char *tbase, int cbase, int pbase,
                                        int lane, epi_u32x4_t (&rpre)[EpiGeom<NI, HQMAX, 16 * NJ>::NH][EpiGeom<NI, HQMAX, 16 * NJ>::NPASS]) {
    typedef EpiGeom<NI, HQMAX, 16 * NJ> G;
    typedef typename epi_out<OTAG>::type OT;
    constexpr bool ZKERNEL = epi_out<OTAG>::zout, D2S = epi_out<OTAG>::d2s;
    static_assert(NJ == 4 || NJ == 3, "wave tiles are 64 or 48 pixels wide");
    constexpr int HQ = G::HQ, NH = G::NH, SLOTS = G::SLOTS, LPP = G::LPP, PPI = G::PPI, NPASS = G::NPASS;
    OT *__restrict__ y = (OT *)a.y;
    const bf16_t *__restrict__ res = (const bf16_t *)a.res;
    const int sl = lane % LPP;

    auto to_lds = [&](int half) {
#pragma unroll
        for (int il = 0; il < HQ / 16; ++il) {
            const int slot = il * 4 + (lane >> 4);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int px = j * 16 + (lane & 15);
                *(f32x4_t *)(tbase + px * (HQ * 4) + ((slot ^ (px & (SLOTS - 1))) << 4)) = acc[half * (HQ / 16) + il][j];
            }
        }
    };
    auto read_row = [&](int px, float (&v)[8]) {
        const char *row = tbase + px * (HQ * 4);
        const int sw = px & (SLOTS - 1);
        const float4 v0 = *(const float4 *)(row + (((2 * sl) ^ sw) << 4));
        const float4 v1 = *(const float4 *)(row + (((2 * sl + 1) ^ sw) << 4));
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
        v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    };
    auto load_affine = [&](int half, float (&sc)[8], float (&sh)[8]) {
        const int co = cbase + half * HQ + sl * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int c = co + r;
            const bool in = c < a.Cout;
            sc[r] = in ? (a.scale ? a.scale[c] : 1.0f) : 0.0f;
            sh[r] = (in && a.shift) ? a.shift[c] : 0.0f;
        }
    };

    auto fast = [&](auto act_tag, auto res_tag, auto z_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool RES = decltype(res_tag)::value;
        constexpr bool ZOUT = decltype(z_tag)::value;          // also store the value before the activation (bf16)
#pragma unroll
        for (int half = 0; half < NH; ++half) {
            float sc[8], sh[8];
            if constexpr (RES && NH >= 2) {
                if (half >= 1) epi_load_res<NI, HQMAX, 16 * NJ>(a, half, cbase, pbase, lane, rpre[half]);   // lands during to_lds
            }
            load_affine(half, sc, sh);
            to_lds(half);
            if (half == 0) YMX_STAMP(4);
            const int co = cbase + half * HQ + sl * 8;
            float nxt[8];
            read_row(lane / LPP, nxt);
#pragma unroll
            for (int n = 0; n < NPASS; ++n) {
                const int px = n * PPI + lane / LPP;
                const int p = pbase + px;
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = nxt[r];
                if (n + 1 < NPASS) read_row(px + PPI, nxt);     // next pass's row is in flight during this one's math
                // pairs of neighbouring channels stay pairs from the affine step to the packed
                // conversion, so that no instruction is spent on re-pairing them
                f32x2_t o[4];
                uint32_t zq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x2_t t = affine2(f32x2_t{v[2 * r], v[2 * r + 1]}, f32x2_t{sc[2 * r], sc[2 * r + 1]},
                                              f32x2_t{sh[2 * r], sh[2 * r + 1]});
                    if constexpr (ZOUT) zq[r] = epi_pack2(t);
                    o[r] = act2<ACT>(t);
                }
                if constexpr (RES) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // a separately rounded add (never fused into the activation's last
                        // multiply): every conv kernel must give the same bits
                        const unsigned w = rpre[half][n][r];
                        o[r] = epi_add2(o[r], f32x2_t{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)});
                    }
                }
                if (p < a.P && co < a.Cout) {
                    OT *yp;
                    if constexpr (D2S) yp = y + conv_out_offset(a, p, co);
                    else yp = y + (size_t)p * a.y_ld + co;
                    if constexpr (ZOUT) store_stream16((bf16_t *)a.z + (size_t)p * a.Cout + co, zq[0], zq[1], zq[2], zq[3]);
                    if constexpr (sizeof(OT) == 2) {
                        store_stream16(yp, epi_pack2(o[0]), epi_pack2(o[1]), epi_pack2(o[2]), epi_pack2(o[3]));
                    } else {
                        store_stream16(yp, o[0].x, o[0].y, o[1].x, o[1].y);
                        store_stream16(yp + 4, o[2].x, o[2].y, o[3].x, o[3].y);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto fast_act = [&](auto res_tag) {
        if (a.act == YMX_ACT_MISH) fast(std::integral_constant<int, YMX_ACT_MISH>{}, res_tag, std::false_type{});
        else if (a.act == YMX_ACT_LEAKY) fast(std::integral_constant<int, YMX_ACT_LEAKY>{}, res_tag, std::false_type{});
        else fast(std::integral_constant<int, YMX_ACT_NONE>{}, res_tag, std::false_type{});
    };
    // with the second output (training forward): the two activations a derivative exists for
    auto fast_z = [&](auto res_tag) {
        if (a.act == YMX_ACT_MISH) fast(std::integral_constant<int, YMX_ACT_MISH>{}, res_tag, std::true_type{});
        else fast(std::integral_constant<int, YMX_ACT_LEAKY>{}, res_tag, std::true_type{});
    };

    if constexpr (D2S) {                         // the launcher guarantees epi_vec_ok, no shortcut, no activation
        fast(std::integral_constant<int, YMX_ACT_NONE>{}, std::false_type{}, std::false_type{});
    } else if constexpr (ZKERNEL) {                     // the launcher guarantees epi_vec_ok, a.z and Mish / leaky
        if (res == nullptr) fast_z(std::false_type{});
        else fast_z(std::true_type{});
    } else if (epi_vec_ok(a) && res == nullptr) {
        fast_act(std::false_type{});
    } else if (epi_res_pre(a)) {
        fast_act(std::true_type{});
    } else {
        // odd channel counts or pixel strides (e.g. the 255-channel output layers)
        const int act = a.act;
        for (int half = 0; half < NH; ++half) {
            float sc[8], sh[8];
            load_affine(half, sc, sh);
            to_lds(half);
            const int co = cbase + half * HQ + sl * 8;
            for (int n = 0; n < NPASS; ++n) {
                const int px = n * PPI + lane / LPP;
                const int p = pbase + px;
                float v[8];
                read_row(px, v);
                if (p >= a.P) continue;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (co + r < a.Cout) {
                        float o = act_f(affine_f(v[r], sc[r], sh[r]), act);
                        if (res) o = __fadd_rn(o, bf16_to_f32(res[(size_t)p * a.res_ld + co + r]));
                        elem<OT>::store(y + (size_t)p * a.y_ld + co + r, o);
                    }
                }
            }
        }
    }
}

}  // namespace ymx
